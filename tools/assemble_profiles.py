#!/usr/bin/env python3
"""Copy the summaries of tools/profile.py runs (gpurun_out/prof/<tag>_<workload>_*) into profiles/ under the round's
name and rebuild profiles/traffic.json (HBM bytes per launch of each workload's dominant kernel).

    python3 tools/assemble_profiles.py --tag x5 --round r02 bruteforce sift hnsw
"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOMINANT = ("bf_scan_bf16_kernel", "bf_scan_f32_kernel<0, false", "bf_scan_f32_kernel<1, false", "bf_scan_f32_kernel<2, false",
            "bf_scan_u8_kernel<4, false", "bf_scan_u8_kernel<2, false", "hnsw_search_mw_kernel", "hnsw_search_kernel")


def main():
    args = sys.argv[1:]
    tag, rnd = "x", "r02"
    if "--tag" in args:
        i = args.index("--tag"); tag = args[i + 1]; del args[i:i + 2]
    if "--round" in args:
        i = args.index("--round"); rnd = args[i + 1]; del args[i:i + 2]
    src = os.path.join(ROOT, "gpurun_out", "prof")
    dst = os.path.join(ROOT, "profiles")
    tpath = os.path.join(dst, "traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    for w in args:
        shutil.copy(os.path.join(src, f"{tag}_{w}_stats.json"), os.path.join(dst, f"{rnd}_{w}_kernel_stats.json"))
        shutil.copy(os.path.join(src, f"{tag}_{w}_pmc.json"), os.path.join(dst, f"{rnd}_{w}_pmc.json"))
        f = os.path.join(src, f"{tag}_{w}_rocprofv3_kernel_stats.csv")
        if os.path.exists(f):
            shutil.copy(f, os.path.join(dst, f"{rnd}_{w}_rocprofv3_kernel_stats.csv"))
        pmc = json.load(open(os.path.join(src, f"{tag}_{w}_pmc.json")))
        best = None   # the dominant kernel's most frequent launch shape = the timed batches of bench.py
        for name, c in pmc.items():
            if not any(d in name for d in DOMINANT) or "FETCH_SIZE" not in c:
                continue
            b = int((2 * c["FETCH_SIZE"]["per_dispatch"] + c.get("WRITE_SIZE", {"per_dispatch": 0})["per_dispatch"]) * 1024)
            nd = c["FETCH_SIZE"].get("dispatches", 0)
            timed = 20 <= nd <= 40      # (3 warm-up + 20 timed launches + the recall / ground-truth calls; the HNSW builder's
            #                              own searches show up with hundreds of launches of another shape)
            if best is None or (timed, b) > (best[3], best[1]):
                best = (name, b, nd, timed)
        if best:
            short = best[0].split("gfxknn::", 1)[1].split("<")[0].split("(")[0]
            traffic[short] = best[1]
            traffic[w] = best[1]
            print(w, short, best[1], "dispatches", best[2])
    traffic["source"] = (f"profiles/{rnd}_<workload>_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; bytes = "
                         "(2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch, the gfx950 correction of MI355X_MICROARCH.md)")
    json.dump(traffic, open(tpath, "w"), indent=1)


if __name__ == "__main__":
    main()
