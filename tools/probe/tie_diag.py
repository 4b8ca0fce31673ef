import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import nmslib_zig_amd as nz
from tests import orc
from tests.gpuutil import make_index
from tests.test_gpu_hnsw_mw import _search
rng = np.random.default_rng(5)
base = rng.integers(0, 3, size=(3000, 24)).astype(np.float32)
X = np.concatenate([base, base, base])[rng.permutation(9000)]
Q = rng.integers(0, 3, size=(128, 24)).astype(np.float32)
idx = make_index("l2", "hnsw", X, M=8, efConstruction=40, indexThreadQty=1)
g = orc.HnswGraph.build("l2", X, 8, 40)
for ef, k in ((10, 10), (7, 5), (16, 16), (64, 10)):
    idx.setQueryTimeParams(efSearch=ef)
    opos, odist, ocnt, ondc, ohops = g.search(Q, k, ef)
    for name, mode, pipe in (("onewave", "0", "1"), ("onewave-nopipe", "0", "0"), ("multiwave", "2", "1")):
        os.environ["NMSLIB_HNSW_PIPE"] = pipe
        r = _search(idx, Q, k, mode)
        print(ef, k, name, "ids!=oracle", int((r[0] != opos).sum()), "dist!=", int((r[1] != odist).sum()),
              "ndc!=", int((r[3][0] != ondc).sum()), "hops!=", int((r[3][1] != ohops).sum()), flush=True)
