"""-m gpu: bench.py's one-process-per-GPU path under the test runner.  Two ranks on the ONE GPU of
the test box, collective over gloo (RCCL refuses two ranks on a device): row shards, packed all-gather of per-shard top-k,
strided merge -- everything but the transport is what the 8-GPU scaling run executes.  The sharded answer must be the
unsharded one."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, tmp_path, gloo):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "NMSLIB_GPU_DEVICE", "NMSLIB_GPU_SHARDS"):
        env.pop(v, None)
    if gloo:
        env["BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=str(tmp_path),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_bruteforce_equals_one_rank(tmp_path):
    common = ["--workload", "bruteforce", "--n", "200000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    one = _bench(["--gpus", "1", "--dump", str(tmp_path / "one.npz")] + common, tmp_path, False)
    two = _bench(["--gpus", "2", "--dump", str(tmp_path / "two.npz")] + common, tmp_path, True)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["rows_per_gpu"] == 100000 and "all-gather" in two["config"]["sharding"]
    assert one["recall_at_k"] == 1.0 and two["recall_at_k"] == 1.0
    assert two["distances_match_reference"] is True
    a, b = np.load(tmp_path / "one.npz"), np.load(tmp_path / "two.npz")
    np.testing.assert_array_equal(a["ids"], b["ids"])
    np.testing.assert_array_equal(a["dists"], b["dists"])


def test_two_ranks_cos768x_weak_scaling_shape(tmp_path):
    """C5's shape at a small size: every rank generates and indexes only its own rows (HNSW, cosine, 768-D); the
    ground truth is the same protocol with the exact scan."""
    two = _bench(["--gpus", "2", "--workload", "cos768x", "--rows-per-gpu", "20000", "--batch", "512", "--rank", "16",
                  "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--dump", str(tmp_path / "c.npz")], tmp_path, True)
    assert two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert two["config"]["rows"] == 40000 and two["config"]["rows_per_gpu"] == 20000
    assert two["recall_at_k"] is not None and two["recall_at_k"] >= 0.9, two["recall_at_k"]
    ids = np.load(tmp_path / "c.npz")["ids"]
    assert ids.shape == (512, 10) and ids.min() >= 0 and ids.max() < 40000
    assert (ids >= 20000).any() and (ids < 20000).any()          # both shards contribute
