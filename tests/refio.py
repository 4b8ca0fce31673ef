"""Helpers around the compiled reference (oracle/_ref) and its on-disk formats.

Test infrastructure only.  The formats parsed here are the reference's own:
  * optimized HNSW index: src/method/hnsw.cc:774-806 (writer) / :1025-1074 (reader)
  * object vectors (.dat): src/space.cc:88-105
"""
import json
import os
import struct
import subprocess
import tempfile

import numpy as np

from . import orc

HAVE_REF = os.path.exists(orc.REF_DRIVER) and os.path.exists(orc.REF_LIB)


def parse_optimized_index(path):
    """Parse the 68-byte header + level-0 block + per-node upper links."""
    with open(path, "rb") as f:
        raw = f.read()
    (flag, n, mem_per_obj, off_l0, off_data, maxlevel, enterpoint, maxM, maxM0, dist_func,
     search_method) = struct.unpack_from("<IIQQQiIQQiQ", raw, 0)
    assert flag == 1, "not an optimized index"
    hdr = 68
    blk = np.frombuffer(raw, np.uint8, n * mem_per_obj, hdr).reshape(n, mem_per_obj)
    obj_hdr = blk[:, off_data:off_data + 16].copy()
    ids = obj_hdr[:, 0:4].copy().view(np.int32).ravel()
    datalen = obj_hdr[:, 8:16].copy().view(np.uint64).ravel()
    payload = blk[:, off_data + 16:off_l0].copy()
    links0 = blk[:, off_l0:off_l0 + 4 * (maxM0 + 1)].copy().view(np.int32).reshape(n, maxM0 + 1)
    # unused slots are uninitialised memory in the file: zero them for comparisons
    cols = np.arange(maxM0 + 1)[None, :]
    links0 = np.where(cols <= links0[:, :1], links0, 0).astype(np.int32)
    pos = hdr + n * mem_per_obj
    levels = np.zeros(n, np.int32)
    up_off = np.full(n, -1, np.int64)
    chunks, cur = [], 0
    for i in range(n):
        (nbytes,) = struct.unpack_from("<I", raw, pos)
        pos += 4
        if nbytes:
            lv = nbytes // ((maxM + 1) * 4)
            levels[i] = lv
            arr = np.frombuffer(raw, np.int32, nbytes // 4, pos).reshape(lv, maxM + 1).copy()
            c = np.arange(maxM + 1)[None, :]
            arr = np.where(c <= arr[:, :1], arr, 0).astype(np.int32)
            up_off[i] = cur
            chunks.append(arr.ravel())
            cur += arr.size
            pos += nbytes
    # this repo's library appends a 32-byte trailer naming the space (ignored by the reference)
    assert pos == len(raw) or (pos + 32 == len(raw) and raw[pos:pos + 8] == b"GFXKNNv1")
    return dict(n=n, mem_per_obj=mem_per_obj, off_level0=off_l0, off_data=off_data,
                maxlevel=maxlevel, enterpoint=enterpoint, maxM=maxM, maxM0=maxM0,
                dist_func=dist_func, search_method=search_method, ids=ids, datalen=datalen,
                payload=payload, links0=links0, levels=levels, up_off=up_off,
                up_links=(np.concatenate(chunks) if chunks else np.zeros(0, np.int32)))


def run_ref_driver(space, method, base, queries, k, index_params="", query_params="", threads=1,
                   repeat=1, save=None, load=None, ids=None, workdir=None):
    """Run oracle/_ref/ref_driver; returns (ids [Q,k], dists [Q,k], cnt [Q], ndc [Q], info)."""
    u8 = space == "l2sqr_sift"
    dt = np.uint8 if u8 else np.float32
    base = np.ascontiguousarray(base, dt)
    queries = np.ascontiguousarray(queries, dt)
    own = workdir is None
    tmp = tempfile.mkdtemp(prefix="refdrv_") if own else workdir
    try:
        bp, qp, op = (os.path.join(tmp, x) for x in ("base.bin", "q.bin", "out"))
        base.tofile(bp)
        queries.tofile(qp)
        cmd = [orc.REF_DRIVER, "--space", space, "--method", method, "--data", bp, "--n",
               str(base.shape[0]), "--dim", str(base.shape[1]), "--queries", qp, "--nq",
               str(queries.shape[0]), "--k", str(k), "--out", op, "--threads", str(threads),
               "--repeat", str(repeat), "--index-params", index_params, "--query-params",
               query_params]
        if u8:
            cmd.append("--u8")
        if save:
            cmd += ["--save", save]
        if load:
            cmd += ["--load", load]
        if ids is not None:
            ip = os.path.join(tmp, "ids.i32")
            np.ascontiguousarray(ids, np.int32).tofile(ip)
            cmd += ["--ids", ip]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("ref_driver failed: " + r.stderr)
        info = json.loads(r.stdout.strip().splitlines()[-1])
        nq = queries.shape[0]
        out_ids = np.fromfile(op + ".ids.i32", np.int32).reshape(nq, k)
        out_d = np.fromfile(op + ".dists.f32", np.float32).reshape(nq, k)
        cnt = np.fromfile(op + ".cnt.i32", np.int32)
        ndc = np.fromfile(op + ".ndc.i64", np.int64)
        return out_ids, out_d, cnt, ndc, info
    finally:
        if own:
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)


# Synthetic inputs and the recall measure live in the package (bench.py uses them without the test package)
from nmslib_zig_amd.datasets import approx_equal_ulps, recall_nmslib, s_gauss, s_lowrank, s_sift_like  # noqa: E402,F401
