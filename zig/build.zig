//! build.zig for B-R-P/NMSLIB-ZIG on top of the MI355X engine (SURVEY.md 8f N3).
//!
//! UNVERIFIED: no Zig toolchain exists in the image this repository is built in.  Written against Zig 0.14's
//! std.Build API, the one the reference's own build.zig uses (build.zig:1-160 of the reference).
//!
//! What changes against the reference's build.zig:101-135: the 49 vendored NMSLIB translation units and
//! nmslib_c.cpp are NOT compiled; the `nmslib` module (the reference's unmodified lib.zig) links the prebuilt
//! libnmslib_c.so, which exports the same 37 C symbols (include/nmslib_c.h).  A second module, `nmslib_gpu_batch`
//! (nmslib_gpu_batch.zig in this directory), adds the batched query entry points that reach the GPU batch kernels.
//!
//!   zig build -Dreference=/path/to/NMSLIB-ZIG -Dengine=/path/to/this/repo
//!   zig build test -Dreference=... -Dengine=...        # the reference's ten tests + the batch test below
const std = @import("std");

pub fn build(b: *std.Build) void {
    const target = b.standardTargetOptions(.{});
    const optimize = b.standardOptimizeOption(.{ .preferred_optimize_mode = .ReleaseFast });
    const reference = b.option([]const u8, "reference", "checkout of B-R-P/NMSLIB-ZIG (for lib.zig)") orelse "../reference";
    const engine = b.option([]const u8, "engine", "root of the MI355X engine repository") orelse "..";

    const include_dir = b.pathJoin(&.{ engine, "include" });
    const lib_dir = b.pathJoin(&.{ engine, "nmslib_zig_amd" });

    // the reference's Zig API, untouched
    const nmslib = b.createModule(.{
        .root_source_file = .{ .cwd_relative = b.pathJoin(&.{ reference, "lib.zig" }) },
        .target = target,
        .optimize = optimize,
        .link_libc = true,
    });
    nmslib.addIncludePath(.{ .cwd_relative = include_dir }); // nmslib_c.h: same declarations as the reference's
    nmslib.addLibraryPath(.{ .cwd_relative = lib_dir });
    nmslib.addRPath(.{ .cwd_relative = lib_dir });
    nmslib.linkSystemLibrary("nmslib_c", .{}); // libnmslib_c.so (HIP kernels inside; needs libamdhip64.so.7 at run time)
    b.modules.put("nmslib", nmslib) catch unreachable;

    // batched entry points on top of it
    const batch = b.createModule(.{
        .root_source_file = b.path("nmslib_gpu_batch.zig"),
        .target = target,
        .optimize = optimize,
        .link_libc = true,
    });
    batch.addImport("nmslib", nmslib);
    batch.addIncludePath(.{ .cwd_relative = include_dir });
    b.modules.put("nmslib_gpu_batch", batch) catch unreachable;

    const test_step = b.step("test", "reference tests + batched-query tests against the GPU engine");
    const ref_tests = b.addTest(.{ .root_module = nmslib });
    test_step.dependOn(&b.addRunArtifact(ref_tests).step);
    const batch_tests = b.addTest(.{ .root_module = batch });
    test_step.dependOn(&b.addRunArtifact(batch_tests).step);
}
