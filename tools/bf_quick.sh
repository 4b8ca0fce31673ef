#!/bin/bash
# quick timing of the brute-force workloads (no CPU legs): ms/step, dominant kernel ms
for w in ${WL:-bruteforce sift}; do
  python bench.py --workload $w --no-cpu-baseline --steps 20 2>/dev/null \
    | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$w', '${TAG}', r['ms_per_step'], r['roofline']['kernel_ms'], r['recall_at_k'], r.get('distances_match_reference'), r['host_entry']['ms_per_step'])"
done
