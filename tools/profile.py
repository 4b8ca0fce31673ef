#!/usr/bin/env python3
"""Profile bench.py on the GPU box with rocprofv3 and write compact summaries.

    python3 tools/profile.py [--workload bruteforce|hnsw|sift] [--tag r01] [bench args...]

Passes (each its own rocprofv3 run, as the MI355X guide prescribes: counters never combined
with traces): kernel-trace --stats, two SQ counter sets, FETCH_SIZE, WRITE_SIZE.  Outputs
gpurun_out/prof/<tag>_<workload>_{stats,pmc}.json ; copy the ones to keep into profiles/.
This process never touches the GPU itself: it only spawns rocprofv3 with `python3 bench.py`
directly after `--`.
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PMC_SETS = {
    "sq1": "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA",
    "sq2": "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM",
    "sq3": "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR",
    "sq4": "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES",
    "grbm": "GRBM_GUI_ACTIVE",
    "ldslat": "LdsLatency",
    "vmemlat": "VmemLatency",
    "sq5": "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT",
    "fetch": "FETCH_SIZE",
    "write": "WRITE_SIZE",
}


def run(cmd, env):
    print("+", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return r.returncode, r.stdout


def main():
    args = sys.argv[1:]
    tag, workload = "r01", "bruteforce"
    if "--tag" in args:
        i = args.index("--tag")
        tag = args[i + 1]
        del args[i:i + 2]
    if "--workload" in args:
        workload = args[args.index("--workload") + 1]
    only_sets = None          # --sets a,b: only these counter passes (and no kernel trace) -> <tag>_<workload>_pmc.json
    if "--sets" in args:
        i = args.index("--sets")
        only_sets = args[i + 1].split(",")
        del args[i:i + 2]
    bench = ["python3", os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--no-cpu-baseline"] + args
    out = os.path.join(ROOT, "gpurun_out", "prof")
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")

    if only_sets is None:
        trace_pass(tag, workload, bench, out, env)
    counter_passes(tag, workload, bench, out, env, only_sets)


def trace_pass(tag, workload, bench, out, env):
    # ---- pass 0: kernel trace + stats ----
    d = os.path.join(out, f"{tag}_{workload}_trace")
    rc, log = run(["rocprofv3", "--kernel-trace", "--stats", "-d", d, "-o", "t", "--output-format", "csv", "--"] + bench, env)
    stats = {"rc": rc, "bench_json": None, "kernels": []}
    for line in log.splitlines():
        if line.startswith("{") and '"metric"' in line:
            stats["bench_json"] = json.loads(line)
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            stats["kernels"].append({k: row[k] for k in row})
    per_kernel = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            # key by (kernel, grid): the ground-truth call of bench.py uses another batch shape
            name = row.get("Kernel_Name", "") + " grid=" + row.get("Grid_Size_X", row.get("Grid_Size", "?"))
            dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
            e = per_kernel.setdefault(name, {"calls": 0, "total_us": 0.0, "min_us": 1e30, "max_us": 0.0, "durs": [],
                                             "vgpr": row.get("VGPR_Count"), "accum_vgpr": row.get("Accum_VGPR_Count"),
                                             "sgpr": row.get("SGPR_Count"), "lds": row.get("LDS_Block_Size"),
                                             "grid": row.get("Grid_Size_X"), "wg": row.get("Workgroup_Size_X")})
            e["calls"] += 1
            e["durs"].append(dur)
            e["total_us"] += dur
            e["min_us"] = min(e["min_us"], dur)
            e["max_us"] = max(e["max_us"], dur)
    for e in per_kernel.values():
        e["avg_us"] = round(e["total_us"] / e["calls"], 2)
        e["total_us"] = round(e["total_us"], 1)
        d = e.pop("durs")
        timed = d[3:] if len(d) > 6 else d                      # bench.py: 3 warm-up launches, then the timed ones
        e["timed_avg_us"] = round(sum(timed) / len(timed), 2)   # what bench.py's HIP events bracket
        e["median_us"] = round(sorted(d)[len(d) // 2], 2)
    stats["per_kernel"] = dict(sorted(per_kernel.items(), key=lambda kv: -kv[1]["total_us"]))
    json.dump(stats, open(os.path.join(out, f"{tag}_{workload}_stats.json"), "w"), indent=1)
    # keep rocprofv3's own summary, drop the raw traces (gpurun copies back at most 64 MiB)
    trace_dir = os.path.join(out, f"{tag}_{workload}_trace")
    for f in glob.glob(os.path.join(trace_dir, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(out, f"{tag}_{workload}_rocprofv3_kernel_stats.csv"))
    shutil.rmtree(trace_dir, ignore_errors=True)
    for k, e in list(stats["per_kernel"].items())[:6]:
        print(f"{e['timed_avg_us']:>10.1f} us (timed avg) x{e['calls']:<4d} vgpr={e['vgpr']}  {k[:100]}")


def counter_passes(tag, workload, bench, out, env, only_sets):
    # ---- counter passes ----
    pmc = {}
    for name, counters in PMC_SETS.items():
        if only_sets is not None and name not in only_sets:
            continue
        d = os.path.join(out, f"{tag}_{workload}_pmc_{name}")
        rc, log = run(["rocprofv3", "--pmc"] + counters.split() + ["-d", d, "-o", "c", "--output-format", "csv", "--"] + bench, env)
        agg = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                # key by (kernel, grid): bench.py's ground-truth call runs the same kernel on another batch shape
                k = row.get("Kernel_Name", "") + " grid=" + str(row.get("Grid_Size", row.get("Grid_Size_X", "?")))
                c = row.get("Counter_Name", "")
                v = float(row.get("Counter_Value", 0) or 0)
                e = agg.setdefault(k, {}).setdefault(c, [0.0, 0])
                e[0] += v
                e[1] += 1
        for k, cs in agg.items():
            for c, (tot, n) in cs.items():
                pmc.setdefault(k, {})[c] = {"per_dispatch": tot / max(1, n), "dispatches": n}
        shutil.rmtree(d, ignore_errors=True)
        if rc != 0:
            pmc.setdefault("_errors", {})[name] = log[-2000:]
    json.dump(pmc, open(os.path.join(out, f"{tag}_{workload}_pmc.json"), "w"), indent=1)
    # brief console summary
    for k, cs in pmc.items():
        if ("hnsw_search" in k or "bf_scan" in k) and cs.get("SQ_WAVES", {}).get("dispatches", 0) >= 15:
            print(k[:90], {c: round(v["per_dispatch"], 1) for c, v in cs.items() if c in
                           ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY",
                            "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_MFMA", "FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE")})
    if only_sets is not None:
        for k, cs in pmc.items():
            if "bf_scan" in k or "hnsw_search" in k or "rerank" in k:
                print(k[:100], {c: round(v["per_dispatch"], 1) for c, v in cs.items()})


if __name__ == "__main__":
    main()
