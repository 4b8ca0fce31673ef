//! Batched k-NN queries for B-R-P/NMSLIB-ZIG over the MI355X engine (SURVEY.md 8f N3).
//!
//! UNVERIFIED: no Zig toolchain in the build image.  The exact C call sequence of this file is replayed, compiled
//! and checked by tests/zig_batch_sequence_check.c (run by tests/test_gpu_hnsw.py).
//!
//! Why: the reference's Index.knnQueryBatch (lib.zig:889-931) loops over nmslib_knn_query_get_size +
//! nmslib_knn_query_fill per query.  Against this engine that is correct but costs one GPU launch chain per query
//! (~0.5 ms each); nmslib_knn_query_batch (nmslib_c.h:319-324, never called by lib.zig) hands the whole batch to the
//! batched kernels.  These functions take the reference's own `Index` (its `handle` field is all they need) and return
//! the reference's own result types, so callers only swap `index.knnQueryBatch(qs, k, null)` for
//! `gpu.knnQueryBatch(&index, qs, k)`.
const std = @import("std");
const nmslib = @import("nmslib");
const c = @cImport({
    @cInclude("nmslib_c.h");
});

pub const Error = nmslib.Error;

fn check(rc: c.nmslib_error_t) Error!void {
    // same mapping as lib.zig:29-73 for the codes this path can return
    return switch (rc) {
        c.NMSLIB_SUCCESS => {},
        c.NMSLIB_ERROR_NULL_POINTER => error.NullPointer,
        c.NMSLIB_ERROR_INVALID_ARGUMENT => error.InvalidArgument,
        c.NMSLIB_ERROR_OUT_OF_MEMORY => error.OutOfMemory,
        c.NMSLIB_ERROR_SPACE_INCOMPATIBLE => error.SpaceIncompatible,
        c.NMSLIB_ERROR_QUERY_TOO_LARGE => error.QueryTooLarge,
        c.NMSLIB_ERROR_INDEX_BUILD_FAILED => error.IndexBuildFailed,
        c.NMSLIB_ERROR_QUERY_EXECUTION_FAILED => error.QueryExecutionFailed,
        else => error.Runtime,
    };
}

fn run(index: *nmslib.Index, flat: *const anyopaque, count: usize, elem_count: usize, k: usize) Error!nmslib.BatchResult {
    const alloc = index.allocator;
    if (!index.built) try index.buildIndex(null, false);
    c.nmslib_initialize_pool(@ptrCast(index.handle)); // lib.zig:802: finalises a dirty index once, then a no-op

    const results = try alloc.alloc(nmslib.QueryResult, count);
    var made: usize = 0;
    errdefer {
        for (results[0..made]) |r| r.deinit();
        alloc.free(results);
    }
    const c_results = try alloc.alloc(c.nmslib_result_t, count);
    defer alloc.free(c_results);
    for (c_results, 0..) |*r, i| { // caller-owned buffers of capacity k (nmslib_c.h:55-60)
        const ids = try alloc.alloc(i32, k);
        errdefer alloc.free(ids);
        const dists = try alloc.alloc(f32, k);
        results[i] = .{ .ids = ids[0..0], .distances = dists[0..0], .full_ids_ptr = ids, .full_dist_ptr = dists, .used = 0, .allocator = alloc };
        made = i + 1;
        r.* = .{ .ids = ids.ptr, .distances = dists.ptr, .size = 0, .capacity = k };
    }
    // ONE call: the engine pads the batch, runs the selection / graph-search kernels once, copies k ids + distances
    // per query back (nmslib_c.cpp:1003-1031 is a serial loop over the same arguments)
    try check(c.nmslib_knn_query_batch(@ptrCast(index.handle), flat, count, elem_count, k, c_results.ptr, null, 0));
    for (c_results, 0..) |r, i| {
        if (r.size > r.capacity) return error.Runtime;
        results[i].ids = results[i].full_ids_ptr.?[0..r.size];
        results[i].distances = results[i].full_dist_ptr.?[0..r.size];
        results[i].used = r.size;
    }
    return .{ .results = results, .allocator = alloc };
}

/// Dense float queries: `queries[i].len` must equal the index dimension.  One GPU batch.
pub fn knnQueryBatch(index: *nmslib.Index, queries: []const []const f32, k: usize) Error!nmslib.BatchResult {
    if (queries.len == 0 or k == 0) return error.InvalidArgument;
    if (index.data_type != .DenseVector) return error.SpaceIncompatible;
    const dim = queries[0].len;
    const flat = try index.allocator.alloc(f32, queries.len * dim); // [count][dim], row-major
    defer index.allocator.free(flat);
    for (queries, 0..) |q, i| {
        if (q.len != dim) return error.InvalidArgument;
        @memcpy(flat[i * dim ..][0..dim], q);
    }
    return run(index, @ptrCast(flat.ptr), queries.len, dim, k);
}

/// Already flat dense queries ([count][dim]): no copy at all on the Zig side.
pub fn knnQueryBatchFlat(index: *nmslib.Index, flat: []const f32, dim: usize, k: usize) Error!nmslib.BatchResult {
    if (dim == 0 or flat.len == 0 or flat.len % dim != 0 or k == 0) return error.InvalidArgument;
    if (index.data_type != .DenseVector) return error.SpaceIncompatible;
    return run(index, @ptrCast(flat.ptr), flat.len / dim, dim, k);
}

/// uint8 SIFT descriptors ([count][128] bytes, space l2sqr_sift).  This engine strides the batch by the index's
/// element size; the reference's nmslib_knn_query_batch strides every type by 4 * elem_count bytes
/// (nmslib_c.cpp:1018-1019) and therefore cannot take a packed uint8 batch.
pub fn knnQueryBatchUInt8(index: *nmslib.Index, queries: []const []const u8, k: usize) Error!nmslib.BatchResult {
    if (queries.len == 0 or k == 0) return error.InvalidArgument;
    if (index.data_type != .DenseUInt8Vector) return error.SpaceIncompatible;
    const dim = queries[0].len;
    const flat = try index.allocator.alloc(u8, queries.len * dim);
    defer index.allocator.free(flat);
    for (queries, 0..) |q, i| {
        if (q.len != dim) return error.InvalidArgument;
        @memcpy(flat[i * dim ..][0..dim], q);
    }
    return run(index, @ptrCast(flat.ptr), queries.len, dim, k);
}

test "batched queries equal the per-query loop" {
    const alloc = std.testing.allocator;
    var index = try nmslib.Index.init(alloc, "l2", null, "brute_force", .DenseVector, .Float);
    defer index.deinit();
    var rows: [64][8]f32 = undefined;
    var prng = std.Random.DefaultPrng.init(7);
    for (&rows) |*r| for (r) |*x| {
        x.* = prng.random().floatNorm(f32);
    };
    var slices: [64][]const f32 = undefined;
    for (&slices, 0..) |*s, i| s.* = rows[i][0..];
    try index.addDenseBatch(slices[0..], null);
    try index.buildIndex(null, false);
    const batch = try knnQueryBatch(&index, slices[0..16], 5);
    defer batch.deinit();
    for (batch.results, 0..) |r, i| {
        const one = try index.knnQuery(.{ .DenseVector = slices[i] }, 5);
        defer one.deinit();
        try std.testing.expectEqualSlices(i32, one.ids, r.ids);
        try std.testing.expectEqualSlices(f32, one.distances, r.distances);
        try std.testing.expectEqual(@as(i32, @intCast(i)), r.ids[0]); // a stored row finds itself first
    }
}
