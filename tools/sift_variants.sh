#!/bin/bash
# timing experiments for the u8 select kernel (results are wrong for dbg&1)
for d in ${@:-0 1 256}; do
  echo -n "dbg=$d "; NMSLIB_GPU_DEBUG=$d python bench.py --workload sift --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['roofline']['kernel_ms'], j['roofline']['achieved'], j['ms_per_step'], j['recall_at_k'])"
done
