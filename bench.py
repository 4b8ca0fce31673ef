#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X.

metric   : queries/sec @ recall@10, 1M x 128-D L2, batch=1024; HBM GB/s vs peak  (BASELINE.json)
workloads: the default run measures three of BASELINE.json's configs in one go:
             C2 "bruteforce"  brute-force L2, 1M x 128 f32, k=10, batch 1024   -> the headline line
             C3 "hnsw"        HNSW l2 1M x 128, M=16 efS=128, k=10, batch 1024 -> "workloads": {"hnsw": ...}
             C4 "sift"        l2sqr_sift 1M x 128 u8, k=100, batch 4096        -> "workloads": {"sift": ...}
           every record carries its own roofline, cpu_baseline, host_entry and a recall against an
           INDEPENDENT exact ground truth (the reference's own sequential scan, oracle/_ref, on a query sample).
           --workload X runs one of them alone (cos768 = one shard of C5).
step     : one batch of queries, already resident in HBM, through the device-resident entry of the C ABI
           (nmslib_gpu_knn_query_batch_device) on torch's current stream.  The reference-ABI entry
           (nmslib_knn_query_batch: host pointers, PCIe inside the call) is timed beside it ("host_entry").
N > 1    : the corpus is sharded by rows over the ranks (one process per GPU); every rank searches its shard
           for the same batch, per-shard top-k lists are all-gathered over RCCL and merged on the GPU
           (nmslib_gpu_merge_topk).  Total corpus fixed -> "strong".  `--gpus N` without a torchrun
           environment launches the N ranks itself.

One JSON line on stdout (rank 0).  torch is plumbing only: device buffers, streams, torch.distributed.
The CPU baseline (rank 0, N == 1) times the real reference (oracle/_ref) on a bounded sample of the same
workload; the oracle is never part of the measured path.
"""
import argparse
import ctypes as C
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_I8_MFMA_TOPS = 5000.0     # i8 = 2x bf16 dense (~2.5 PF) per MI355X_MICROARCH.md
PEAK_BF16_MFMA_TFLOPS = 2500.0  # bf16 dense
PEAK_HBM_GBS = 8000.0          # HBM3E spec

WORKLOADS = {
    # name: space, method, dim, batch, k, description
    "bruteforce": dict(space="l2", method="seq_search", dim=128, batch=1024, k=10,
                       desc="brute-force L2 1Mx128 f32 k=10 batch=1024 (BASELINE configs[1])"),
    "hnsw": dict(space="l2", method="hnsw", dim=128, batch=1024, k=10,
                 desc="HNSW l2 1Mx128 f32 M=16 efS={ef} k=10 batch=1024 (BASELINE configs[2])"),
    "sift": dict(space="l2sqr_sift", method="seq_search", dim=128, batch=4096, k=100,
                 desc="l2sqr_sift 1Mx128 u8 k=100 batch=4096 (BASELINE configs[3])"),
    "cos768": dict(space="cosinesimil", method="hnsw", dim=768, batch=8192, k=10,
                   desc="HNSW cosinesimil {n}x{dim} f32 M=16 efS={ef} k=10 batch={batch} (one shard of BASELINE configs[4])"),
    # the sharded form of C5, scaled to what a bench run can build: rows_per_gpu fixed (weak), every rank generates
    # and indexes only its own rows, per-shard top-k all-gathered over RCCL and merged
    "cos768x": dict(space="cosinesimil", method="hnsw", dim=768, batch=8192, k=10,
                    desc="HNSW cosinesimil {n}x{dim} f32 sharded {world} ways, M=16 efS={ef} k=10 batch={batch} "
                         "(BASELINE configs[4] at {rpg} rows per GPU)"),
}


def note(msg):
    """progress line on stderr (long host-side phases must not look hung)"""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["all"] + list(WORKLOADS), default="all")
    ap.add_argument("--rows-per-gpu", type=int, default=1_000_000, help="cos768x: rows indexed by every rank")
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--ef", type=int, default=128)
    ap.add_argument("--ef-sweep", default="", help="hnsw / cos768: extra efSearch values measured on the same index, e.g. 256,512,1000")
    ap.add_argument("--rank", type=int, default=16, help="cos768 / cos768x: latent rank of the synthetic rows (16: the recall "
                                                         "set, recall@10 ~0.99 at efS=128; 64: the stress set, ~0.45 at 1M rows "
                                                         "for the reference's graph too -- tests/golden/c5_ref_1m768.json)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--index-cache", default="", help="hnsw, 1 GPU: save the built index here / load it if present")
    ap.add_argument("--gpu-build", type=int, default=-1, help="hnsw: 1 = batched GPU construction, 0 = host, -1 = library default")
    ap.add_argument("--space", default="", help="bruteforce workload: another dense space (l1, linf, cosinesimil, ...)")
    ap.add_argument("--index-extra", default="", help="hnsw: extra index parameters, k=v,k=v (experiments)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="queries in the CPU baseline sample (0 = auto)")
    ap.add_argument("--gt-sample", type=int, default=128, help="queries in the exact-ground-truth sample")
    ap.add_argument("--dump", default="", help="rank 0 saves the ids / distances of the last timed step here (.npz): "
                                                  "tests compare sharded and unsharded runs")
    return ap.parse_args()


def s_768(n, dim, seed, rank=64, chunk=1 << 18):
    """S-768: rank-`rank` latent + 0.1 noise, rows L2-normalised; generated in chunks."""
    A = np.random.default_rng(45).standard_normal((dim, rank)).astype(np.float32)
    out = np.empty((n, dim), np.float32)
    rng = np.random.default_rng(seed)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        x = rng.standard_normal((hi - lo, rank), dtype=np.float32) @ A.T
        x += 0.1 * rng.standard_normal((hi - lo, dim), dtype=np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        out[lo:hi] = x
    return out


def s_768_rows(lo, hi, dim, seed, rank=64, chunk=1 << 18):
    """Rows [lo, hi) of the S-768 family with one generator per 2^18-row chunk, so that every rank of a sharded run can
    produce exactly its own rows (the 100M x 768 corpus of C5 never exists in one place)."""
    A = np.random.default_rng(45).standard_normal((dim, rank)).astype(np.float32)
    out = np.empty((hi - lo, dim), np.float32)
    pos = lo
    while pos < hi:
        c = pos // chunk
        rng = np.random.default_rng([seed, c])
        m = chunk
        x = rng.standard_normal((m, rank), dtype=np.float32) @ A.T
        x += 0.1 * rng.standard_normal((m, dim), dtype=np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        a0, a1 = pos - c * chunk, min(hi, (c + 1) * chunk) - c * chunk
        out[pos - lo:pos - lo + (a1 - a0)] = x[a0:a1]
        pos += a1 - a0
    return out


def host_threads():
    """Hardware threads this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


class Ctx:
    pass


# ------------------------------------------------------------------------------------------------------------
# The reference on the host CPU (oracle/_ref): exact ground truth for recall + the timed CPU baseline.
# ------------------------------------------------------------------------------------------------------------
def reference_legs(w, X, Q, ef, want_baseline, cpu_sample, gt_sample):
    """-> (gt_ids, gt_dists, cpu_baseline dict or None, ref_ids of the baseline sample or None).
    gt_* = the reference's exact sequential scan at k+22 on the first gt_sample queries (tie-extended recall needs
    more than k entries).  The baseline = the reference's own method for this workload, all hardware threads,
    query q on thread q mod T (Experiments::Execute protocol), on a bounded sample."""
    from tests import orc, refio
    cores = host_threads()
    space, k = w["space"], w["k"]
    ngt = min(gt_sample, Q.shape[0])
    if not refio.HAVE_REF:
        # oracle/_ref was not shipped: this repo's scalar restatement (1 thread) on a handful of queries
        ngt = min(ngt, 8)
        t1 = time.time()
        gi, gd, _ = orc.seq_search(space, X, Q[:ngt], min(k + 22, X.shape[0]))
        base = None
        if want_baseline:
            base = {"value": round(ngt / (time.time() - t1), 2), "unit": "queries/s", "cores": 1, "kind": "port",
                    "sample": f"{ngt} of the {Q.shape[0]} queries against all {X.shape[0]} rows, exact scan (oracle port; "
                              "oracle/_ref not shipped)"}
        return gi, gd, base
    tmp = tempfile.mkdtemp(prefix="bench_ref_")
    try:
        note(f"reference exact scan for the ground truth ({ngt} queries, {cores} threads)")
        gi, gd, _, _, _ = refio.run_ref_driver(space, "seq_search", X, Q[:ngt], min(k + 22, X.shape[0]), "", "",
                                               threads=cores, workdir=tmp)
        base = None
        if want_baseline:
            t0 = time.time()
            if w["method"] == "hnsw":
                ip, qp = f"M=16,efConstruction=200,indexThreadQty={cores}", f"efSearch={ef}"
                ns = cpu_sample or min(Q.shape[0], 1024)
                rep = 3
            else:
                ip, qp = "", ""
                # ~10-30 s of CPU work: the reference scans ~25 (f32) / ~80 (u8) queries/s/thread at 1M rows
                per_thr = 80 if space == "l2sqr_sift" else 25
                ns = cpu_sample or min(Q.shape[0], max(64, int(per_thr * cores * 15 * 1e6 / max(X.shape[0], 1))))
                rep = 1
            note(f"timing the CPU baseline: reference {w['method']} on {ns} queries, {cores} threads")
            ids, d, cnt, ndc, info = refio.run_ref_driver(space, w["method"], X, Q[:ns], k, ip, qp, threads=cores,
                                                          repeat=rep, workdir=tmp)
            m = min(ns, ngt)
            gdd = gd[:m] ** 2 if (w["method"] == "hnsw" and space == "l2") else gd[:m]   # HNSW-l2 returns squared L2
            rec = refio.recall_nmslib(ids[:m], gi[:m], gdd, k, integer=(space == "l2sqr_sift"))
            base = {"value": round(float(info["qps"]), 2), "unit": "queries/s", "cores": cores, "kind": "reference",
                    "sample": f"{ns} of the {Q.shape[0]} queries against all {X.shape[0]} rows, method={w['method']}"
                              + (f", {qp}" if qp else "") + f", {cores} threads, best of {rep}",
                    "recall_at_k": round(float(rec), 4), "wall_s": round(time.time() - t0, 1)}
            if w["method"] == "hnsw":
                base["build_s"] = info["build_s"]
        return gi, gd, base
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def gpu_exact_ground_truth(space, Xs, lo, hi, Q, ngt, kk, cx):
    """Exact scan of the (possibly sharded) corpus for the first ngt queries: every rank scans its own rows, the
    per-shard lists are all-gathered and merged like the timed path.  Rank 0 gets (ids, dists); others None.
    The scan itself is checked against the reference's seq_search in the bruteforce / sift workloads and in tests/."""
    import torch
    import torch.distributed as dist
    import nmslib_zig_amd as nz
    bf = nz.Index(space, "seq_search")
    bf.addDenseBatch(Xs, np.arange(lo, hi, dtype=np.int32))
    bf.buildIndex()
    dq = torch.from_numpy(np.ascontiguousarray(Q[:ngt])).to(cx.dev)
    pack = torch.empty((2, ngt, kk), dtype=torch.int32, device=cx.dev)
    cnt = torch.empty((ngt,), dtype=torch.int32, device=cx.dev)
    st = torch.cuda.current_stream()
    bf.knn_device(dq.data_ptr(), ngt, Q.shape[1], kk, pack[0].data_ptr(), pack[1].data_ptr(), cnt.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    bf.close()
    if cx.world == 1:
        return pack[0].cpu().numpy(), pack[1].view(torch.float32).cpu().numpy()
    g = torch.empty((cx.world * 2, ngt, kk), dtype=torch.int32, device=cx.dev)
    if cx.backend == "nccl":
        dist.all_gather_into_tensor(g, pack)
    else:
        gh = torch.empty(g.shape, dtype=g.dtype)
        dist.all_gather_into_tensor(gh, pack.cpu())
        g.copy_(gh)
    m_ids = torch.empty((ngt, kk), dtype=torch.int32, device=cx.dev)
    m_ds = torch.empty((ngt, kk), dtype=torch.float32, device=cx.dev)
    nz._check(nz.lib().nmslib_gpu_merge_topk_strided(g.data_ptr() + ngt * kk * 4, g.data_ptr(), 2 * ngt * kk, cx.world, ngt, kk,
                                                     m_ds.data_ptr(), m_ids.data_ptr(), st.cuda_stream))
    torch.cuda.synchronize()
    return (m_ids.cpu().numpy(), m_ds.cpu().numpy()) if cx.rank == 0 else (None, None)


# ------------------------------------------------------------------------------------------------------------
def run_workload(a, name, cx):
    import torch
    import torch.distributed as dist
    import nmslib_zig_amd as nz
    from nmslib_zig_amd import datasets as refio   # synthetic sets + NMSLIB's recall (no test package in the timed path)

    w = dict(WORKLOADS[name])
    single = a.workload != "all"
    n = a.n
    if single:
        w["dim"] = a.dim or w["dim"]
        w["batch"] = a.batch or w["batch"]
        w["k"] = a.k or w["k"]
        if a.space and name == "bruteforce":
            w["space"] = a.space
    space, method, dim, nq, k = w["space"], w["method"], w["dim"], w["batch"], w["k"]
    u8 = space == "l2sqr_sift"
    rank, world, dev = cx.rank, cx.world, cx.dev

    sharded_gen = name == "cos768x"
    if sharded_gen:
        n = a.rows_per_gpu * world
    note(f"[{name}] generating data: n={n} dim={dim}")
    lo, hi = rank * n // world, (rank + 1) * n // world          # this rank's row shard
    if name == "sift":
        X, Q = refio.s_sift_like(n, 44), refio.s_sift_like(nq, 45)
    elif name == "cos768":
        X, Q = s_768(n, dim, 46, a.rank), s_768(nq, dim, 47, a.rank)
    elif sharded_gen:
        Xs, Q = s_768_rows(lo, hi, dim, 46, a.rank), s_768_rows(0, nq, dim, 47, a.rank)   # own rows only
    else:
        X, Q = refio.s_lowrank(n, dim, 42), refio.s_lowrank(nq, dim, 43)
    if not sharded_gen:
        Xs = X[lo:hi]
    cache = a.index_cache if (method == "hnsw" and world == 1 and single) else ""
    build_kw = {"gpu_build": a.gpu_build} if a.gpu_build >= 0 else {}
    build_kw.update(dict(kv.split("=") for kv in a.index_extra.split(",") if kv))
    t_build = time.time()
    graph_build_s = None
    if cache and os.path.exists(cache):
        note(f"loading cached index {cache}")
        idx = nz.Index.load(cache, load_data=False)              # the reference's optimized-index format
        idx.finalize()
    else:
        idx = nz.Index(space, method, data_type="DenseUInt8Vector" if u8 else "DenseVector",
                       dist_type="Int" if u8 else "Float")
        ids = np.arange(lo, hi, dtype=np.int32)                  # external id = global row
        (idx.addUInt8Batch if u8 else idx.addDenseBatch)(Xs, ids)
        if method == "hnsw":
            idx.buildIndex(M=16, efConstruction=200, **build_kw)
            graph_build_s = idx.stats()["build_seconds"]
            note(f"[{name}] graph built in {graph_build_s:.2f}s")
            if cache:
                idx.save(cache, False)
        else:
            idx.buildIndex()
    if method == "hnsw":
        idx.setQueryTimeParams(efSearch=a.ef)
    t_build = time.time() - t_build
    note(f"[{name}] index ready in {t_build:.1f}s; timing {a.steps} steps")

    dq = torch.from_numpy(Q).to(dev)
    pack = torch.empty((2, nq, k), dtype=torch.int32, device=dev)    # ids | distance bits: one collective moves both
    d_ids, d_ds = pack[0], pack[1].view(torch.float32)
    d_cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
    if world > 1:
        g_pack = torch.empty((world * 2, nq, k), dtype=torch.int32, device=dev)   # [world][2][nq][k]
        m_ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
        m_ds = torch.empty((nq, k), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream()

    def step():
        idx.knn_device(dq.data_ptr(), nq, Q.shape[1], k, d_ids.data_ptr(), d_ds.data_ptr(), d_cnt.data_ptr(),
                       stream.cuda_stream)
        if world > 1:
            if cx.backend == "nccl":
                dist.all_gather_into_tensor(g_pack, pack)
            else:
                g_host = torch.empty(g_pack.shape, dtype=g_pack.dtype)
                dist.all_gather_into_tensor(g_host, pack.cpu())
                g_pack.copy_(g_host)
            nz._check(nz.lib().nmslib_gpu_merge_topk_strided(g_pack.data_ptr() + nq * k * 4, g_pack.data_ptr(),
                                                             2 * nq * k, world, nq, k, m_ds.data_ptr(),
                                                             m_ids.data_ptr(), stream.cuda_stream))

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync()
    idx.kernel_timing(enable=True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    kern_ms, launches = idx.kernel_timing(enable=False, collect=True)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    res_ids = (m_ids if world > 1 else d_ids).cpu().numpy()
    res_ds = (m_ds if world > 1 else d_ds).cpu().numpy()
    if a.dump and rank == 0:
        np.savez(a.dump, ids=res_ids, dists=res_ds)
    counters = idx.read_counters(nq) if method == "hnsw" else None
    stats = idx.stats()
    last_path = int(stats.get("last_path", 0))
    gpu_gt = None
    if sharded_gen or (name == "cos768" and n > 1_000_000):
        # the reference's sequential scan of tens of GB is out of reach of a bench run: exact GPU scan instead
        note(f"[{name}] exact GPU scan for the ground truth")
        ngt = min(a.gt_sample, nq)
        gpu_gt = gpu_exact_ground_truth(space, Xs, lo, hi, Q, ngt, k + 22, cx)
    if rank != 0:
        idx.close()
        return None

    # ---- roofline of the dominant kernel (this rank's launches; HIP events on the launch stream) ---------
    # every interval recorded inside the timed loop belongs to one of its steps: a sliced batch (or a SearchOld batch
    # that was answered again with larger workspaces) has several per step, and all of them are the kernel's time
    kern_s = kern_ms / 1e3 / max(1, a.steps)
    rows_local = hi - lo
    if method == "hnsw":
        ndc, hops, hops_up = (c.astype(np.float64) for c in counters)
        # bytes/query = ndc*D*4 + hops0*(maxM0+1)*4 + hops_up*(maxM+1)*4 + ndc (visited)
        alg_bytes = float((ndc * dim * 4 + hops * 33 * 4 + hops_up * 17 * 4 + ndc).sum())
        roof = {"bound": "hbm", "achieved": round(alg_bytes / kern_s / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "kernel": ("hnsw_search_mw_kernel" if a.ef <= 256 and os.environ.get("NMSLIB_HNSW_MW", "1") != "0"
                           else ("hnsw_search_old_kernel" if a.ef >= 1000 else "hnsw_search_kernel")),
                "ndc_per_query": round(float(ndc.mean()), 1), "hops_per_query": round(float(hops.mean()), 1),
                "kernel_launches_per_step": round(launches / max(1, a.steps), 2)}
    elif u8:
        ops = 2.0 * nq * rows_local * 128
        roof = {"bound": "mfma", "achieved": round(ops / kern_s / 1e12, 2), "peak": PEAK_I8_MFMA_TOPS, "unit": "TOP/s",
                "kernel": "bf_scan_u8_kernel" if last_path == 3 else "bf_select_u8_kernel"}
    elif last_path == 1:
        # selection on the fp16 / bf16 matrix cores.  Per query tile the threshold kernel picks the scan: ONE fp16 product per
        # element (bf_scan_bf16_kernel; operands fp16(s x), s a power of two) where the sample shows room for its error, else the split product (f32 operands
        # as hi + lo bf16: qh.bh + qh.bl + ql.bh, bf_scan_f32_kernel).  `achieved` counts the algorithmic 2*Q*N*D.
        flops = 2.0 * nq * rows_local * dim             # 
        tiles, precise = int(stats.get("fast_tiles", 0)), int(stats.get("fast_tiles_precise", 0))
        products = 3.0 if tiles == 0 else (3.0 * precise + 1.0 * (tiles - precise)) / tiles
        one = tiles > 0 and precise == 0
        roof = {"bound": "mfma", "achieved": round(flops / kern_s / 1e12, 2), "peak": PEAK_BF16_MFMA_TFLOPS,
                "unit": "TFLOP/s", "kernel": "bf_scan_bf16_kernel" if one else "bf_scan_f32_kernel",
                "arithmetic": ("one fp16 product per element (fp16(s q) . fp16(s b), f32 accumulate) under a per-query error bound, "
                               if one else
                               "bf16 x 3 (f32 rows and queries split into hi + lo bf16; qh.bh + qh.bl + ql.bh, f32 accumulate), ")
                              + "proof + exact f32 re-rank",
                "query_tiles": tiles, "query_tiles_split_product": precise,
                "fallback_tiles": int(stats.get("fast_tiles_fallback", 0)),
                "mfma_products_per_element": round(products, 2),
                "issued_tflops": round(products * flops / kern_s / 1e12, 2),
                "frac_issued": round(products * flops / kern_s / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                "vs_f32_mfma_peak": round(flops / kern_s / 1e12 / PEAK_F32_MFMA_TFLOPS, 3),
                "clock_note": "peak is the 2.4 GHz figure; s_memtime inside the kernel shows ~1.75 GHz under this load "
                              "(DESIGN.md 6)"}
    else:
        flops = 2.0 * nq * rows_local * dim             # 2*Q*N*D
        roof = {"bound": "mfma", "achieved": round(flops / kern_s / 1e12, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "kernel": "bf_select_f32_kernel"}
    roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
    roof["kernel_ms"] = round(kern_s * 1e3, 4)
    roof["traffic"] = None
    tr = os.path.join(ROOT, "profiles", "traffic.json")   # HBM bytes/launch from a separate rocprofv3 --pmc pass
    std_shape = ((n == 1_000_000) if name != "cos768x" else (a.rows_per_gpu == 1_000_000 and a.rank == 16 and nq == 8192))
    if os.path.exists(tr) and std_shape and world == 1 and dim == WORKLOADS[name]["dim"] and not a.space and a.ef == 128:
        try:
            tj = json.load(open(tr))
            roof["traffic"] = tj.get(name)     # per workload: the same kernel moves other bytes on another shape
            if roof["traffic"] is not None:
                roof["traffic_source"] = tj.get("source")
        except Exception:
            pass
    if method == "hnsw":
        roof["algorithmic_bytes"] = int(alg_bytes)
        if roof["traffic"]:
            # rows that many queries of a batch visit are served by L2 / MALL: the bytes that really crossed the HBM
            # interface are the measured ones, and that is the fraction of the HBM peak the kernel holds
            roof["hbm_gbs_measured"] = round(roof["traffic"] / kern_s / 1e9, 1)
            roof["frac_hbm_measured"] = round(roof["traffic"] / kern_s / 1e9 / PEAK_HBM_GBS, 4)
            if roof["traffic"] < 0.9 * alg_bytes:
                roof["note"] = ("achieved / frac count the algorithmic bytes (SURVEY.md 8d); "
                                f"{100 * (1 - roof['traffic'] / alg_bytes):.0f} % of them were cache hits: frac_hbm_measured is "
                                "the share of the HBM peak")
    # the dominant kernel cannot take longer than the step that contains it
    roof["consistent"] = bool(kern_s * 1e3 <= elapsed / a.steps * 1e3 * 1.02)

    out = {
        "metric": {"sift": f"queries/sec @ recall@{k}, {n} x 128-D u8 l2sqr_sift, batch={nq}",
                   "hnsw": f"queries/sec @ recall@{k}, {n} x {dim}-D L2 HNSW, batch={nq}; HBM GB/s vs peak",
                   "cos768": f"queries/sec @ recall@{k}, {n} x {dim}-D cosinesimil HNSW, batch={nq}",
                   "cos768x": f"queries/sec @ recall@{k}, {n} x {dim}-D cosinesimil HNSW sharded {world} ways, batch={nq}"}.get(
                       name, "queries/sec @ recall@10, 1M x 128-D L2, batch=1024; HBM GB/s vs peak"),
        "value": round(a.steps * nq / elapsed, 1),
        "unit": "queries/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak" if sharded_gen else "strong",
        "vs_baseline": None,
        "dtype": "u8" if u8 else (("f16+f32" if stats.get("fast_tiles_precise", 0) == 0 and stats.get("fast_tiles", 0) else "bf16x3+f32")
                                  if last_path == 1 else "f32"),
        "data": "synthetic",
        "config": {
            "workload": w["desc"].format(ef=a.ef, n=n, dim=dim, batch=nq, world=world, rpg=hi - lo),
            "rows": n, "dim": dim, "batch": nq, "k": k, "rows_per_gpu": rows_local,
            "dataset": "S-sift-like seeds 44/45" if u8 else (f"S-768 rank-{a.rank} + 0.1 noise, unit rows, seeds 46/47"
                                                             if name in ("cos768", "cos768x") else
                                                             "S-lowrank rank-16 + 0.1 noise, seeds 42/43"),
            "sharding": (f"rows/{world} + {'RCCL' if cx.backend == 'nccl' else cx.backend} all-gather of per-shard top-k"
                         if world > 1 else "single GPU"),
            "entry": "nmslib_gpu_knn_query_batch_device (queries and results resident in HBM)",
            "build_s": round(t_build, 2),
        },
        "roofline": roof,
    }
    if graph_build_s is not None:
        out["config"]["graph_build_s"] = round(graph_build_s, 2)

    # ---- other efSearch values on the same index (recall filled in below) -----------------------------------------
    sweep = []
    if method == "hnsw" and a.ef_sweep and world == 1:
        for ef2 in [int(x) for x in a.ef_sweep.split(",") if x]:
            idx.setQueryTimeParams(efSearch=ef2)
            step()
            sync()
            idx.kernel_timing(enable=True)
            t1 = time.perf_counter()
            for _ in range(a.steps):
                step()
            sync()
            el2 = time.perf_counter() - t1
            kms, kl = idx.kernel_timing(enable=False, collect=True)
            c2 = [c.astype(np.float64) for c in idx.read_counters(nq)]
            ab = float((c2[0] * dim * 4 + c2[1] * 33 * 4 + c2[2] * 17 * 4 + c2[0]).sum())
            ks = kms / 1e3 / max(1, a.steps)
            sweep.append({"ef": ef2, "value": round(a.steps * nq / el2, 1), "ms_per_step": round(el2 / a.steps * 1e3, 4),
                          "kernel_ms": round(ks * 1e3, 4), "kernel_launches_per_step": round(kl / max(1, a.steps), 2),
                          "consistent": bool(ks * 1e3 <= el2 / a.steps * 1e3 * 1.02), "hbm_gbs": round(ab / ks / 1e9, 1),
                          "frac": round(ab / ks / 1e9 / PEAK_HBM_GBS, 4), "ndc_per_query": round(float(c2[0].mean()), 1),
                          "_ids": d_ids.cpu().numpy().copy()})
        idx.setQueryTimeParams(efSearch=a.ef)

    # ---- the reference-ABI entry: nmslib_knn_query_batch, host pointers, PCIe inside the call --------------
    if world == 1:
        h_ids = np.empty((nq, k), np.int32)
        h_ds = np.empty((nq, k), np.float32)
        res = (nz.Result * nq)()
        for i in range(nq):
            res[i] = nz.Result(h_ids[i].ctypes.data_as(C.POINTER(C.c_int32)), h_ds[i].ctypes.data_as(C.POINTER(C.c_float)), 0, k)
        hsteps = max(3, min(a.steps, 10))
        L = nz.lib()
        nz._check(L.nmslib_knn_query_batch(idx.h, Q.ctypes.data, nq, Q.shape[1], k, res, None, 0))
        th = time.perf_counter()
        for _ in range(hsteps):
            nz._check(L.nmslib_knn_query_batch(idx.h, Q.ctypes.data, nq, Q.shape[1], k, res, None, 0))
        th = (time.perf_counter() - th) / hsteps
        out["host_entry"] = {"entry": "nmslib_knn_query_batch (host pointers, PCIe inside the call)",
                             "ms_per_step": round(th * 1e3, 4), "queries_per_s": round(nq / th, 1),
                             "matches_device_entry": bool(np.array_equal(h_ids, res_ids))}
        # what a caller of the reference's own ABI (lib.zig, C) gets, beside `value` (queries resident in HBM)
        out["value_host_entry"] = round(nq / th, 1)
    idx.close()

    # ---- recall against an independent exact ground truth + the CPU baseline ---------------------------------
    # (cos768 beyond 200k rows: the reference's 768-D build takes too long for a bench run; exact scan only)
    want_base = world == 1 and not a.no_cpu_baseline and not (name in ("cos768", "cos768x") and n > 200_000)
    try:
        if gpu_gt is not None:
            gt_ids, gt_d = gpu_gt
            base = None
            gt_label = f"exact GPU scan (seq_search over the same rows), first {len(gt_ids)} queries, tie-extended at k+22"
        else:
            gt_ids, gt_d, base = reference_legs(w, Xs if sharded_gen else X, Q, a.ef, want_base, a.cpu_sample, a.gt_sample)
            gt_label = None
        m = len(gt_ids)
        gdd = gt_d ** 2 if (method == "hnsw" and space == "l2") else gt_d      # HNSW-l2 returns squared L2
        out["recall_at_k"] = round(float(refio.recall_nmslib(res_ids[:m], gt_ids, gdd, k, integer=u8)), 4)
        out["recall_ground_truth"] = gt_label or f"reference seq_search (oracle/_ref), exact, first {m} queries, tie-extended at k+22"
        for rec_ in sweep:
            rec_["recall_at_k"] = round(float(refio.recall_nmslib(rec_.pop("_ids")[:m], gt_ids, gdd, k)), 4)
        if sweep:
            out["ef_sweep"] = sweep
        if method != "hnsw":
            # the exact method must also reproduce the reference's distances on the sample (1e-5 rel; integers exact)
            ok = np.array_equal(res_ds[:m], gt_d[:, :k]) if u8 else np.allclose(res_ds[:m], gt_d[:, :k], rtol=1e-5, atol=1e-6)
            out["distances_match_reference"] = bool(ok)
        if base is not None:
            out["cpu_baseline"] = base
    except Exception as e:  # the baseline must never take the GPU number down with it
        for rec_ in sweep:
            rec_.pop("_ids", None)
        out["recall_at_k"] = None
        out["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": 0, "kind": "reference", "sample": f"failed: {e}"}
    return out


def heartbeat():
    """a line on stderr every minute: long host-side phases (data generation, 768-D builds) must not look hung"""
    import threading

    def run():
        t0 = time.time()
        while True:
            time.sleep(60)
            note(f"... still working ({time.time() - t0:.0f}s)")
    threading.Thread(target=run, daemon=True).start()


def main():
    a = parse()
    heartbeat()
    # --gpus N without a torchrun environment: launch the N ranks ourselves (before anything touches the GPU)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        port = 29500 + (os.getpid() % 2000)
        # ("--" ends torchrun's own options: it abbreviates, and "--n" would match its --nnodes)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), "--", os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: refusing to report a mismatched n_gpus", file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist
    cx = Ctx()
    cx.rank, cx.world = rank, world
    torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
    os.environ.setdefault("NMSLIB_GPU_DEVICE", str(torch.cuda.current_device()))
    cx.dev = torch.device("cuda", torch.cuda.current_device())
    # BENCH_BACKEND=gloo: rehearsal of the multi-rank protocol on ONE GPU (RCCL refuses two ranks on a device): the
    # collective then goes through host memory; everything else (shards, packed layout, strided merge) is the real path
    cx.backend = os.environ.get("BENCH_BACKEND", "nccl")
    if world > 1:
        if cx.backend == "nccl":
            dist.init_process_group("nccl", device_id=cx.dev)   # RCCL over xGMI
        else:
            dist.init_process_group(cx.backend)

    if a.workload == "all":
        out = run_workload(a, "bruteforce", cx)
        subs = {}
        for name in ("hnsw", "sift", "cos768x"):
            r = run_workload(a, name, cx)
            if r is not None:
                subs[name] = r
        if out is not None:
            out["workloads"] = subs
    else:
        out = run_workload(a, a.workload, cx)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
