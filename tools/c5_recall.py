#!/usr/bin/env python3
"""C5 graph-quality experiment: recall@10 vs efSearch on S-768 (or any rank) for graphs built by
  (a) the reference (oracle/_ref/ref_driver, all host threads)   -- searched on the GPU through its saved index file,
  (b) this library's host builder (same threads),
  (c) this library's batched GPU builder, for several batch schedules.
All graphs are searched by the same GPU kernel (same-graph parity with the reference's search is a tested
property), so differences are graph quality only.  Ground truth = exact GPU scan (cosinesimil).

  python tools/c5_recall.py --n 1000000 --ref --host --gpu div=16 div=64 ...
Prints one JSON line per graph.  Test infrastructure / experiment only.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nmslib_zig_amd as nz  # noqa: E402
from tests import refio  # noqa: E402


def s_768(n, dim, seed, rank, noise, chunk=1 << 18):
    A = np.random.default_rng(45).standard_normal((dim, rank)).astype(np.float32)
    out = np.empty((n, dim), np.float32)
    rng = np.random.default_rng(seed)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        x = rng.standard_normal((hi - lo, rank), dtype=np.float32) @ A.T
        x += noise * rng.standard_normal((hi - lo, dim), dtype=np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        out[lo:hi] = x
    return out


def note(m):
    print(f"[c5 {time.strftime('%H:%M:%S')}] {m}", file=sys.stderr, flush=True)


def curve(idx, Q, k, efs, gt_ids, gt_d):
    out = {}
    for ef in efs:
        idx.setQueryTimeParams(efSearch=ef)
        ids, _, _ = idx.knnQueryBatch(Q, k)
        t0 = time.perf_counter()
        ids, _, _ = idx.knnQueryBatch(Q, k)
        dt = time.perf_counter() - t0
        ndc = idx.read_counters(len(Q))[0].mean()
        out[str(ef)] = {"recall": round(refio.recall_nmslib(ids, gt_ids, gt_d, k), 4), "ndc": round(float(ndc), 1),
                        "host_qps": round(len(Q) / dt, 1)}
    return out


def heartbeat():
    """a line on stderr every minute: long host-side builds must not look hung"""
    import threading

    def run():
        t0 = time.time()
        while True:
            time.sleep(60)
            note(f"... still working ({time.time() - t0:.0f}s)")
    threading.Thread(target=run, daemon=True).start()


def main():
    heartbeat()
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--rank", type=int, default=64)
    ap.add_argument("--noise", type=float, default=0.1)
    ap.add_argument("--nq", type=int, default=512)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--M", type=int, default=16)
    ap.add_argument("--efc", type=int, default=200)
    ap.add_argument("--efs", default="64,128,256,512,1000")
    ap.add_argument("--space", default="cosinesimil")
    ap.add_argument("--ref", action="store_true")
    ap.add_argument("--host", action="store_true")
    ap.add_argument("--gpu", nargs="*", default=None, help="GPU builder variants: comma lists of k=v index parameters")
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    efs = [int(x) for x in a.efs.split(",")]
    threads = a.threads or len(os.sched_getaffinity(0))
    note(f"data n={a.n} dim={a.dim} rank={a.rank} noise={a.noise}")
    X, Q = s_768(a.n, a.dim, 46, a.rank, a.noise), s_768(a.nq, a.dim, 47, a.rank, a.noise)
    bf = nz.Index(a.space, "seq_search")
    bf.addDenseBatch(X)
    bf.buildIndex()
    gt_ids, gt_d, _ = bf.knnQueryBatch(Q, a.k + 22)
    bf.close()
    base = {"n": a.n, "dim": a.dim, "rank": a.rank, "noise": a.noise, "space": a.space, "M": a.M, "efC": a.efc}
    for spec in (a.gpu or []):
        extra = dict(kv.split("=") for kv in spec.split(",") if kv and kv != "default")
        idx = nz.Index(a.space, "hnsw")
        idx.addDenseBatch(X)
        idx.buildIndex(M=a.M, efConstruction=a.efc, gpu_build=1, **extra)
        bs = idx.stats()["build_seconds"]
        note(f"gpu[{spec}] built in {bs:.1f}s")
        print(json.dumps({**base, "graph": f"gpu:{spec}", "build_s": round(bs, 2),
                          "curve": curve(idx, Q, a.k, efs, gt_ids, gt_d)}), flush=True)
        idx.close()
    if a.host:
        idx = nz.Index(a.space, "hnsw")
        idx.addDenseBatch(X)
        idx.buildIndex(M=a.M, efConstruction=a.efc, gpu_build=0, indexThreadQty=threads)
        bs = idx.stats()["build_seconds"]
        note(f"host builder: {bs:.1f}s")
        print(json.dumps({**base, "graph": f"host:{threads}thr", "build_s": round(bs, 2),
                          "curve": curve(idx, Q, a.k, efs, gt_ids, gt_d)}), flush=True)
        idx.close()
    if a.ref:
        assert refio.HAVE_REF, "oracle/_ref missing"
        with tempfile.TemporaryDirectory(prefix="c5ref_") as tmp:
            path = os.path.join(tmp, "ref.idx")
            note("reference build (ref_driver)")
            ids, d, cnt, ndc, info = refio.run_ref_driver(a.space, "hnsw", X, Q, a.k,
                                                          f"M={a.M},efConstruction={a.efc},indexThreadQty={threads}",
                                                          "efSearch=128", threads=threads, save=path, workdir=tmp)
            note(f"reference built in {info['build_s']:.1f}s")
            rec_cpu = refio.recall_nmslib(ids, gt_ids, gt_d, a.k)
            idx = nz.Index.load(path, load_data=False)
            idx.finalize()
            print(json.dumps({**base, "graph": f"reference:{threads}thr", "build_s": round(info["build_s"], 2),
                              "cpu_recall_ef128": round(rec_cpu, 4), "cpu_qps_ef128": info["qps"],
                              "curve": curve(idx, Q, a.k, efs, gt_ids, gt_d)}), flush=True)
            idx.close()


if __name__ == "__main__":
    main()
