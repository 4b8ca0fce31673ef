import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import nmslib_zig_amd as nz
from tests import orc, refio
from tests.gpuutil import make_index
n, nq, k = 80000, 1024, 10
X, Q = refio.s_gauss(n, 128, 171), refio.s_gauss(nq, 128, 172)
opos, odist, _ = orc.seq_search("l2", X, Q[:64], k)
idx = make_index("l2", "seq_search", X)
for terms in ("1", "3"):
    if terms == "3": os.environ["NMSLIB_GPU_F32_TERMS"] = "3"
    ids, ds, cnt = idx.knnQueryBatch(Q, k)
    st = idx.stats()
    print("terms", terms, "tiles", st["fast_tiles"], "precise", st["fast_tiles_precise"], "fallback", st["fast_tiles_fallback"],
          "ids==oracle", float((ids[:64] == opos).mean()), flush=True)
