"""Row sharding of one corpus over the GPUs of a node (one process per GPU).

k-NN over disjoint row shards is independent up to one exchange step: every rank answers the
whole query batch from its shard, the per-shard top-k lists ([Q][k] distances + global ids) are
all-gathered (RCCL over xGMI on GPUs: 8 B * Q * k per rank, latency-bound) and merged by
(distance, id).  The reference has no counterpart (it is single-process); SURVEY.md 8e.
"""
import numpy as np


def shard_range(rank, world, n):
    """Rows [lo, hi) owned by `rank`; global id = position in the unsharded corpus."""
    return rank * n // world, (rank + 1) * n // world


def all_gather_topk(dist, dists, ids):
    """dists/ids: [Q, k] tensors on this rank -> ([world, Q, k], [world, Q, k]) on every rank."""
    import torch
    world = dist.get_world_size()
    nq = dists.shape[0]
    # concatenated output form ([world*Q, k]): accepted by both RCCL and gloo
    g_d = torch.empty((world * nq,) + tuple(dists.shape[1:]), dtype=dists.dtype, device=dists.device)
    g_i = torch.empty((world * nq,) + tuple(ids.shape[1:]), dtype=ids.dtype, device=ids.device)
    dist.all_gather_into_tensor(g_d, dists.contiguous())
    dist.all_gather_into_tensor(g_i, ids.contiguous())
    return g_d.view((world,) + tuple(dists.shape)), g_i.view((world,) + tuple(ids.shape))


def all_gather_topk_packed(dist, pack):
    """pack: int32 [2, Q, k] on this rank (ids, then the float32 distances' bit patterns) ->
    int32 [world, 2, Q, k] on every rank with ONE collective (8 B * Q * k per rank)."""
    import torch
    world = dist.get_world_size()
    g = torch.empty((world * pack.shape[0],) + tuple(pack.shape[1:]), dtype=pack.dtype, device=pack.device)
    dist.all_gather_into_tensor(g, pack.contiguous())
    return g.view((world,) + tuple(pack.shape))


def merge_topk_reference(g_d, g_i, k):
    """Host statement of what nmslib_gpu_merge_topk computes (used by the CPU protocol test):
    per query the k smallest (distance, id) pairs over all shards; id < 0 entries are padding."""
    g_d, g_i = np.asarray(g_d), np.asarray(g_i)
    world, nq, kk = g_d.shape
    out_d = np.full((nq, k), np.inf, np.float32)
    out_i = np.full((nq, k), -1, np.int32)
    for q in range(nq):
        d = g_d[:, q, :].reshape(-1)
        i = g_i[:, q, :].reshape(-1)
        keep = i >= 0
        d, i = d[keep], i[keep]
        order = np.lexsort((i, d))[:k]
        out_d[q, :len(order)] = d[order]
        out_i[q, :len(order)] = i[order]
    return out_d, out_i
