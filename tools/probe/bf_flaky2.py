"""Stress probe 2: unsharded indexes of several sizes, repeated batches, deviations from the exact oracle."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
import nmslib_zig_amd as nz
from tests import orc, refio
from tests.gpuutil import make_index

nq, k = 600, 10
X, Q = refio.s_lowrank(140003, 128, 21), refio.s_lowrank(nq, 128, 22)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for n in [int(x) for x in sys.argv[2].split(",")]:
    Xn = np.ascontiguousarray(X[:n])
    idx = make_index("l2", "seq_search", Xn, gpu_shards=1)
    opos, odist, _ = orc.seq_search("l2", Xn, Q[:64], k)
    first = idx.knnQueryBatch(Q, k)
    okfirst = np.array_equal(first[0][:64], opos)
    bad = 0
    devq = {}
    for it in range(reps):
        r = idx.knnQueryBatch(Q, k)
        d = np.nonzero((r[0] != first[0]).any(1))[0]
        if len(d):
            bad += 1
            for q in d.tolist():
                devq[q] = devq.get(q, 0) + 1
    st = idx.stats()
    print(f"n={n} path={st['last_path']} first==oracle(64q)={okfirst} deviating batches {bad}/{reps} queries {devq}", flush=True)
    idx.close()
