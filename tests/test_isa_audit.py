"""ISA audit of the scan kernels (no GPU needed: hipcc cross-compiles).

bf_scan_bf16_kernel / bf_scan_f32_kernel issue their LDS fragment reads as asynchronous inline-asm `ds_read_b128`
and make them valid at hand-counted `s_waitcnt lgkmcnt(N)`.  The compiler does not know that a register targeted by
such a read holds nothing until the wait: in round 2 it copied fragments IN FRONT of their wait on the last block's
path (a race that lost neighbours in ~8 % of 600-query batches; tests/test_gpu_bruteforce.py::
test_fast_paths_are_deterministic).  This test replays every scan kernel's instruction stream with the queue of
outstanding LDS reads and fails if any instruction touches the destination of a read that no wait has retired yet."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _regs(tok):
    """v[a:b] / vN operands of one instruction line -> set of VGPR numbers (AGPRs and SGPRs ignored)."""
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(a), int(b) + 1))
    out.update(int(x) for x in re.findall(r"\bv(\d+)\b", tok))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_no_instruction_touches_an_lds_fragment_before_its_wait(tmp_path):
    src = os.path.join(ROOT, "nmslib_zig_amd", "csrc", "kernels", "bf_kernels.hip")
    asm = str(tmp_path / "bf.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", "-mllvm",
                           "-amdgpu-mfma-vgpr-form", "-S", "--cuda-device-only", src, "-o", asm],
                          stderr=subprocess.DEVNULL)
    kernels, cur, name = {}, None, None
    for line in open(asm):
        m = re.match(r"^(_ZN6gfxknn\d+bf_scan_(?:bf16|f32)_kernel\w+):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if ".end_amdhsa_kernel" in line:
                kernels[name], cur = cur, None
            else:
                cur.append(line)
    assert len(kernels) >= 12, sorted(kernels)
    for name, lines in kernels.items():
        pending = []      # (destination registers, line number) of LDS reads no wait has retired, oldest first
        at_label = {}     # reads in flight on the forward branches into a label
        reads = 0
        for no, line in enumerate(lines):
            ins = line.split(";")[0].strip()
            if not ins or ins.startswith("."):
                if ins.startswith(".LBB") and ins.endswith(":"):      # a label: paths join (forward branches only)
                    for item in at_label.pop(ins[:-1], []):
                        if item not in pending:
                            pending.append(item)
                    pending.sort(key=lambda it: it[1])
                continue
            if ins.endswith(":"):
                continue
            op = ins.split()[0]
            if op == "s_branch" or op.startswith("s_cbranch"):
                at_label.setdefault(ins.split()[1], []).extend(pending)
                if op == "s_branch":
                    pending = []          # the code behind an unconditional branch is reached from elsewhere
                continue
            if op.startswith("ds_read"):
                dst = _regs(ins.split(",")[0])
                addr = _regs(",".join(ins.split(",")[1:]))
                for regs, at in pending:
                    assert not (regs & (dst | addr)), f"{name}: line {no} `{ins}` reuses registers of the read at line {at}"
                pending.append((dst, no))
                reads += 1
                continue
            m = re.match(r"s_waitcnt\s+(.*)", ins)
            if m:
                c = re.search(r"lgkmcnt\((\d+)\)", m.group(1))
                if c:
                    keep = int(c.group(1))
                    pending = pending[len(pending) - keep:] if keep else []
                continue
            if op in ("s_barrier", "s_nop", "s_endpgm") or op.startswith("s_"):
                continue
            used = _regs(ins)
            for regs, at in pending:
                assert not (regs & used), f"{name}: line {no} `{ins}` touches the LDS read of line {at} before its wait"
        assert reads >= 16, (name, reads)
