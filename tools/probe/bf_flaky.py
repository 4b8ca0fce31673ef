"""Stress probe: is the large-batch float fast path deterministic?  Repeats the same batch on an unsharded and a
2-shard index and reports every deviation from the first unsharded answer (and what the exact oracle says there)."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
import nmslib_zig_amd as nz
from tests import orc, refio
from tests.gpuutil import make_index

n, nq, k = 140003, 600, 10
X, Q = refio.s_lowrank(n, 128, 21), refio.s_lowrank(nq, 128, 22)
X[69990:70020] = X[3]
Q[0] = X[3]
one = make_index("l2", "seq_search", X, gpu_shards=1)
two = make_index("l2", "seq_search", X, gpu_shards=2)
ref = one.knnQueryBatch(Q, k)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = {"one": 0, "two": 0}
seen = set()
for it in range(reps):
    for name, idx in (("one", one), ("two", two)):
        r = idx.knnQueryBatch(Q, k)
        d = np.nonzero((r[0] != ref[0]).any(1) | (r[1] != ref[1]).any(1))[0]
        if len(d):
            bad[name] += 1
            print(it, name, "deviating queries", d.tolist(), "path", idx.stats()["last_path"], idx.stats()["fast_tiles_fallback"], flush=True)
            seen.update(d.tolist())
print("deviations", bad, "of", reps)
if seen:
    qs = sorted(seen)[:8]
    opos, odist, _ = orc.seq_search("l2", X, Q[qs], k)
    for j, qi in enumerate(qs):
        print("q", qi, "ref==oracle", np.array_equal(ref[0][qi], opos[j]), "ref ids", ref[0][qi].tolist(), "oracle", opos[j].tolist())
