#!/bin/bash
# counters of the select kernel for two debug variants (timing experiments)
cd /tmp; export TMPDIR=/tmp
for d in "$@"; do
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
    rm -rf /tmp/pmc_$d
    NMSLIB_GPU_DEBUG=$d rocprofv3 --pmc $set -d /tmp/pmc_$d -o c --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
    python3 - "$d" <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(lambda:[0.0,0])
for f in glob.glob(f"/tmp/pmc_{sys.argv[1]}/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "bf_select" in r["Kernel_Name"]:
            e=agg[r["Counter_Name"]]; e[0]+=float(r["Counter_Value"]); e[1]+=1
print("dbg",sys.argv[1],{k:round(v[0]/v[1]/1e6,2) for k,v in sorted(agg.items())})
PY
  done
done
