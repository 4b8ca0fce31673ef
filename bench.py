#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X.

metric   : queries/sec @ recall@10, 1M x 128-D L2, batch=1024  (BASELINE.json)
workload : default = BASELINE.json configs[1]: brute-force L2, 1M x 128 f32, k=10, batch 1024
           (MFMA Q x B^T selection + exact re-rank).  --workload hnsw selects configs[2]
           (HNSW M=16 efS=128), --workload sift configs[3] (uint8, k=100, batch 4096).
step     : one batch of queries, already resident in HBM, through the device-resident entry of
           the C ABI (nmslib_gpu_knn_query_batch_device) on torch's current stream.
N > 1    : the corpus is sharded by rows over the ranks (one process per GPU); every rank
           searches its shard for the same batch, per-shard top-k lists are all-gathered over
           RCCL and merged on the GPU (nmslib_gpu_merge_topk).  Total corpus fixed -> "strong".

One JSON line on stdout (rank 0).  torch is plumbing only: device buffers, streams, events,
torch.distributed.  The CPU baseline (rank 0, N == 1) times the real reference (oracle/_ref)
on a bounded sample of the same workload; the oracle is never part of the measured path.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import nmslib_zig_amd as nz  # noqa: E402
from tests import refio  # noqa: E402  (synthetic data generators + recall definition only)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_I8_MFMA_TOPS = 5000.0     # i8 = 2x bf16 dense (~2.5 PF) per MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0          # HBM3E spec


def note(msg):
    """progress line on stderr (long host-side phases must not look hung)"""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["bruteforce", "hnsw", "sift", "cos768"], default="bruteforce")
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--ef", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--index-cache", default="", help="hnsw, 1 GPU: save the built index here / load it if present")
    ap.add_argument("--gpu-build", type=int, default=-1, help="hnsw: 1 = batched GPU construction, 0 = host, -1 = library default")
    ap.add_argument("--space", default="", help="bruteforce workload: another dense space (l1, linf, cosinesimil, ...)")
    ap.add_argument("--index-extra", default="", help="hnsw: extra index parameters, k=v,k=v (experiments)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="queries in the CPU baseline sample (0 = auto)")
    return ap.parse_args()


def s_768(n, dim, seed, rank=64, chunk=1 << 18):
    """S-768: rank-64 latent + 0.1 noise, rows L2-normalised; generated in chunks."""
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


def make_data(a):
    if a.workload == "sift":
        return refio.s_sift_like(a.n, 44), refio.s_sift_like(a.batch, 45)
    if a.workload == "cos768":
        return s_768(a.n, a.dim, 46), s_768(a.batch, a.dim, 47)
    return refio.s_lowrank(a.n, a.dim, 42), refio.s_lowrank(a.batch, a.dim, 43)


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


def cpu_baseline(a, X, Q, gt_ids, gt_d):
    """The reference's own code (oracle/_ref/ref_driver) on this host, all hardware threads,
    query q on thread q mod T (Experiments::Execute protocol).  Bounded sample."""
    from tests import orc
    cores = host_threads()
    space = {"sift": "l2sqr_sift", "cos768": "cosinesimil"}.get(a.workload, "l2")
    if a.workload in ("hnsw", "cos768"):
        method, ip, qp = "hnsw", f"M=16,efConstruction=200,indexThreadQty={cores}", f"efSearch={a.ef}"
        ns = a.cpu_sample or min(Q.shape[0], 1024)
    else:
        method, ip, qp = "seq_search", "", ""
        ns = a.cpu_sample or min(Q.shape[0], max(64, 32 * cores))  # ~20-30 s of CPU work at 1M rows
    Qs = Q[:ns]
    t0 = time.time()
    if refio.HAVE_REF:
        ids, d, cnt, ndc, info = refio.run_ref_driver(space, method, X, Qs, a.k, ip, qp, threads=cores, repeat=1)
        kind, qps, used = "reference", info["qps"], cores
        extra = {"build_s": info["build_s"]} if method == "hnsw" else {}
    else:
        # oracle/_ref was not shipped: time this repo's scalar restatement instead (1 thread)
        ns = min(ns, 8)
        Qs = Q[:ns]
        t1 = time.time()
        ids, d, _ = orc.seq_search(space, X, Qs, a.k)
        kind, qps, used, extra = "port", ns / (time.time() - t1), 1, {}
    rec = None
    if gt_ids is not None:
        m = min(ns, len(gt_ids))
        gd = gt_d[:m] ** 2 if (a.workload == "hnsw") else gt_d[:m]   # HNSW-l2 returns squared L2
        rec = refio.recall_nmslib(ids[:m], gt_ids[:m], gd, a.k, integer=(a.workload == "sift"))
    out = {"value": round(float(qps), 2), "unit": "queries/s", "cores": used, "kind": kind,
           "sample": f"{ns} of the {Q.shape[0]} queries against all {X.shape[0]} rows, method={method}"
                     + (f", {qp}" if qp else "") + f", {used} threads",
           "wall_s": round(time.time() - t0, 1)}
    if rec is not None:
        out["recall_at_k"] = round(float(rec), 4)
    out.update(extra)
    return out


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if a.dim is None:
        a.dim = 768 if a.workload == "cos768" else 128
    if a.batch is None:
        a.batch = {"sift": 4096, "cos768": 8192}.get(a.workload, 1024)
    if a.workload == "cos768":
        a.no_cpu_baseline = a.no_cpu_baseline or a.n > 200_000   # the reference's 768-D build takes too long beyond that
    if a.k is None:
        a.k = 100 if a.workload == "sift" else 10
    torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
    os.environ.setdefault("NMSLIB_GPU_DEVICE", str(torch.cuda.current_device()))
    dev = torch.device("cuda", torch.cuda.current_device())
    # BENCH_BACKEND=gloo: rehearsal of the multi-rank protocol on ONE GPU (RCCL refuses two ranks on a device): the
    # collective then goes through host memory; everything else (shards, packed layout, strided merge) is the real path
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    note(f"generating data: workload={a.workload} n={a.n}")
    X, Q = make_data(a)
    n, nq, k = X.shape[0], Q.shape[0], a.k
    lo, hi = rank * n // world, (rank + 1) * n // world          # this rank's row shard
    u8 = a.workload == "sift"
    space = "l2sqr_sift" if u8 else ("cosinesimil" if a.workload == "cos768" else "l2")
    if a.space and a.workload == "bruteforce":
        space = a.space
    method = "hnsw" if a.workload in ("hnsw", "cos768") else "seq_search"
    cache = a.index_cache if (method == "hnsw" and world == 1) else ""
    t_build = time.time()
    if cache and os.path.exists(cache):
        note(f"loading cached index {cache}")
        idx = nz.Index.load(cache, load_data=False)              # the reference's optimized-index format
        idx.finalize()
        idx.setQueryTimeParams(efSearch=a.ef)
    else:
        idx = nz.Index(space, method, data_type="DenseUInt8Vector" if u8 else "DenseVector",
                       dist_type="Int" if u8 else "Float")
        ids = np.arange(lo, hi, dtype=np.int32)                      # external id = global row
        (idx.addUInt8Batch if u8 else idx.addDenseBatch)(X[lo:hi], ids)
        note("building index (rows -> HBM" + (", HNSW construction" if method == "hnsw" else "") + ")")
        if method == "hnsw":
            extra = dict(kv.split("=") for kv in a.index_extra.split(",") if kv)
            idx.buildIndex(M=16, efConstruction=200, **({"gpu_build": a.gpu_build} if a.gpu_build >= 0 else {}), **extra)
            note(f"graph built in {idx.stats()['build_seconds']:.2f}s")
            idx.setQueryTimeParams(efSearch=a.ef)
            if cache:
                idx.save(cache, False)
        else:
            idx.buildIndex()
    t_build = time.time() - t_build
    note(f"index ready in {t_build:.1f}s; timing {a.steps} steps")

    dq = torch.from_numpy(Q).to(dev)
    pack = torch.empty((2, nq, k), dtype=torch.int32, device=dev)    # ids | distance bits: one collective moves both
    d_ids, d_ds = pack[0], pack[1].view(torch.float32)
    d_cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
    if world > 1:
        g_pack = torch.empty((world * 2, nq, k), dtype=torch.int32, device=dev)   # [world][2][nq][k], concatenated form
        m_ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
        m_ds = torch.empty((nq, k), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream()

    def step():
        idx.knn_device(dq.data_ptr(), nq, Q.shape[1], k, d_ids.data_ptr(), d_ds.data_ptr(), d_cnt.data_ptr(),
                       stream.cuda_stream)
        if world > 1:
            if backend == "nccl":
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
    counters = idx.read_counters(nq) if method == "hnsw" else None

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- ground truth for recall (exact k-NN; tie-extended per NMSLIB's definition) ----------
    gt_ids = gt_d = None
    recall = None
    if world > 1 and backend != "nccl" and method != "hnsw":
        # rehearsal only: the merged result of the shards against one exact scan of the whole corpus
        full = nz.Index(space, "seq_search", data_type="DenseUInt8Vector" if u8 else "DenseVector",
                        dist_type="Int" if u8 else "Float")
        (full.addUInt8Batch if u8 else full.addDenseBatch)(X)
        full.buildIndex()
        f_ids, f_ds, _ = full.knnQueryBatch(Q, k)
        full.close()
        recall = float((f_ids == res_ids).mean())
        assert np.array_equal(f_ds, res_ds), "sharded + merged distances differ from the unsharded scan"
    if world == 1:
        if method == "hnsw":
            ngt = nq if a.workload == "hnsw" else min(nq, 256)
            idx.close()                                                        # free the HBM copy first
            bf = nz.Index(space, "seq_search")
            bf.addDenseBatch(X)
            bf.buildIndex()
            gt_ids, gt_d, _ = bf.knnQueryBatch(Q[:ngt], k + 22)
            bf.close()
            # HNSW-l2 returns squared L2; cosine distances are the same on both paths
            recall = refio.recall_nmslib(res_ids[:ngt], gt_ids, gt_d ** 2 if space == "l2" else gt_d, k)
        else:
            # brute force IS the exact method; its recall against itself at k+22 checks the tie rule
            gt_ids, gt_d, _ = idx.knnQueryBatch(Q[:64], min(k + 22, 512))
            recall = refio.recall_nmslib(res_ids[:64], gt_ids, gt_d, k, integer=u8)

    # ---- roofline of the dominant kernel ---------------------------------------------------------
    kern_s = kern_ms / 1e3 / max(1, launches)
    rows_local = hi - lo
    if method == "hnsw":
        ndc, hops, hops_up = (c.astype(np.float64) for c in counters)
        D = Q.shape[1]
        # bytes/query = ndc*D*4 + hops0*(maxM0+1)*4 + hops_up*(maxM+1)*4 + ndc (visited)
        alg_bytes = float((ndc * D * 4 + hops * 33 * 4 + hops_up * 17 * 4 + ndc).sum())
        roof = {"bound": "hbm", "achieved": round(alg_bytes / kern_s / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "kernel": "hnsw_search_kernel", "ndc_per_query": round(float(ndc.mean()), 1),
                "hops_per_query": round(float(hops.mean()), 1)}
    elif u8:
        ops = 2.0 * nq * rows_local * 128
        roof = {"bound": "mfma", "achieved": round(ops / kern_s / 1e12, 2), "peak": PEAK_I8_MFMA_TOPS, "unit": "TOP/s",
                "kernel": "bf_select_u8_kernel"}
    else:
        flops = 2.0 * nq * rows_local * Q.shape[1]             # 2*Q*N*D
        roof = {"bound": "mfma", "achieved": round(flops / kern_s / 1e12, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "kernel": "bf_select_f32_kernel"}
    roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
    roof["kernel_ms"] = round(kern_s * 1e3, 4)
    roof["traffic"] = None
    tr = os.path.join(ROOT, "profiles", "traffic.json")       # PMC-measured HBM bytes/launch, if collected
    if os.path.exists(tr):
        try:
            roof["traffic"] = json.load(open(tr)).get(a.workload) if (a.n == 1_000_000 and world == 1 and a.dim in (128, 768) and not (a.workload == "bruteforce" and a.dim != 128)) else None
        except Exception:
            pass

    out = {
        "metric": {"sift": f"queries/sec @ recall@{k}, {n} x 128-D u8 l2sqr_sift, batch={nq}",
                   "cos768": f"queries/sec @ recall@{k}, {n} x {Q.shape[1]}-D cosinesimil HNSW, batch={nq}"}.get(
                       a.workload, "queries/sec @ recall@10, 1M x 128-D L2, batch=1024"),
        "value": round(a.steps * nq / elapsed, 1),
        "unit": "queries/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u8" if u8 else "f32",
        "data": "synthetic",
        "config": {
            "workload": {"bruteforce": "brute-force L2 1Mx128 f32 k=10 batch=1024 (BASELINE configs[1])",
                         "hnsw": f"HNSW l2 1Mx128 f32 M=16 efS={a.ef} k=10 batch=1024 (BASELINE configs[2])",
                         "sift": "l2sqr_sift 1Mx128 u8 k=100 batch=4096 (BASELINE configs[3])",
                         "cos768": f"HNSW cosinesimil {n}x{Q.shape[1]} f32 M=16 efS={a.ef} k=10 batch={nq} "
                                   "(one shard of BASELINE configs[4])"}[a.workload],
            "rows": n, "dim": int(Q.shape[1]), "batch": nq, "k": k, "rows_per_gpu": rows_local,
            "dataset": "S-sift-like seeds 44/45" if u8 else ("S-768 rank-64 + 0.1 noise, unit rows, seeds 46/47"
                                                             if a.workload == "cos768" else
                                                             "S-lowrank rank-16 + 0.1 noise, seeds 42/43"),
            "sharding": f"rows/{world} + RCCL all-gather of per-shard top-k" if world > 1 else "single GPU",
            "build_s": round(t_build, 2),
        },
        "recall_at_k": None if recall is None else round(float(recall), 4),
        "roofline": roof,
    }
    if world == 1 and a.workload != "cos768":
        # the reference's own host-pointer entry (nmslib_knn_query_batch): H2D of the batch + D2H of the results inside
        # the call.  Reported beside the device-resident number, never as `value`.
        import ctypes as C
        h_ids = np.empty((nq, k), np.int32)
        h_ds = np.empty((nq, k), np.float32)
        res = (nz.Result * nq)()
        for i in range(nq):
            res[i] = nz.Result(h_ids[i].ctypes.data_as(C.POINTER(C.c_int32)), h_ds[i].ctypes.data_as(C.POINTER(C.c_float)), 0, k)
        hsteps = max(3, min(a.steps, 10))
        L = nz.lib()
        if method == "hnsw":                      # (the HNSW index was closed for the ground-truth pass: rebuild it)
            idx = nz.Index(space, method)
            idx.addDenseBatch(X[lo:hi], np.arange(lo, hi, dtype=np.int32))
            idx.buildIndex(M=16, efConstruction=200, **({"gpu_build": a.gpu_build} if a.gpu_build >= 0 else {}))
            idx.setQueryTimeParams(efSearch=a.ef)
        nz._check(L.nmslib_knn_query_batch(idx.h, Q.ctypes.data, nq, Q.shape[1], k, res, None, 0))
        th = time.perf_counter()
        for _ in range(hsteps):
            nz._check(L.nmslib_knn_query_batch(idx.h, Q.ctypes.data, nq, Q.shape[1], k, res, None, 0))
        th = (time.perf_counter() - th) / hsteps
        out["host_entry"] = {"entry": "nmslib_knn_query_batch (host pointers, PCIe inside the call)",
                             "ms_per_step": round(th * 1e3, 4), "queries_per_s": round(nq / th, 1),
                             "matches_device_entry": bool(np.array_equal(h_ids, res_ids))}
    if world == 1 and not a.no_cpu_baseline:
        note("timing the CPU baseline (oracle/_ref) on a bounded sample")
        try:
            out["cpu_baseline"] = cpu_baseline(a, X, Q, gt_ids, gt_d)
        except Exception as e:  # the baseline must never take the GPU number down with it
            out["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": 0, "kind": "reference",
                                   "sample": f"failed: {e}"}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
