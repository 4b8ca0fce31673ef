"""The multi-GPU protocol on CPU: world_size 2 over gloo.  Each rank owns a row shard, answers the
whole batch from it (the CPU oracle stands in for the GPU search here -- this test is about the
partition / all-gather layout / merge order, not about kernels), gathers the per-shard top-k and
merges; the merged result must equal the oracle's scan of the unsharded corpus."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nmslib_zig_amd.shard import all_gather_topk, all_gather_topk_packed, merge_topk_reference, shard_range
from tests import orc, refio


def _worker(rank, world, port, space, n, k, q_out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if space == "l2sqr_sift":
            X, Q = refio.s_sift_like(n, 5), refio.s_sift_like(12, 6)
        else:
            X, Q = refio.s_lowrank(n, 32, 5), refio.s_lowrank(12, 32, 6)
        lo, hi = shard_range(rank, world, n)
        pos, d, cnt = orc.seq_search(space, X[lo:hi], Q, k)
        gid = np.where(pos >= 0, pos + lo, -1).astype(np.int32)       # global ids
        g_d, g_i = all_gather_topk(dist, torch.from_numpy(d), torch.from_numpy(gid))
        assert tuple(g_d.shape) == (world, Q.shape[0], k)
        m_d, m_i = merge_topk_reference(g_d.numpy(), g_i.numpy(), k)
        full_pos, full_d, _ = orc.seq_search(space, X, Q, k)
        ok = np.array_equal(m_i, full_pos) and np.array_equal(m_d, full_d)
        # the packed form bench.py uses: ids and distance bit patterns in ONE all-gather
        pack = torch.from_numpy(np.stack([gid, d.astype(np.float32).view(np.int32)]))
        g = all_gather_topk_packed(dist, pack).numpy()
        assert g.shape == (world, 2, Q.shape[0], k)
        p_d, p_i = merge_topk_reference(g[:, 1].copy().view(np.float32), g[:, 0], k)
        ok = ok and np.array_equal(p_i, full_pos) and np.array_equal(p_d, full_d)
        if rank == 0:
            q_out.put(bool(ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("space,n,k", [("l2", 1001, 10), ("l2sqr_sift", 777, 25), ("l2", 7, 10)])
def test_sharded_topk_equals_unsharded(space, n, k):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, space, n, k, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_shard_ranges_partition_the_corpus():
    for n in (0, 1, 7, 1000, 1_000_003):
        for w in (1, 2, 3, 8):
            r = [shard_range(i, w, n) for i in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
