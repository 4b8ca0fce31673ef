"""-m gpu: the HNSW search kernel through the C ABI.  Same-graph parity: graphs come either from
the reference's own saved index file (golden) or from this library's deterministic single-thread
build (which test_cabi_cpu.py proves equal to the reference's), so GPU results must match the
reference's SearchV1Merge item for item; floats within 1e-5 relative."""
import numpy as np
import pytest

import nmslib_zig_amd as nz
from tests import orc, refio
from tests.gpuutil import ULP1, close_rel, ids_match_modulo_near_ties, ids_match_modulo_ties, make_index, sim_close

pytestmark = pytest.mark.gpu


def test_search_reference_built_index_file(golden, tmp_path):
    p = str(tmp_path / "ref.idx")
    golden["hnsw_l2_index_file"].tofile(p)
    idx = nz.Index.load(p, load_data=False)
    qs = golden["f32_D128_queries"]
    for ef in (5, 20, 200):
        idx.setQueryTimeParams(efSearch=ef)
        ids, ds, cnt = idx.knnQueryBatch(qs, 10)
        want_i, want_d = golden[f"hnsw_l2_ef{ef}_v1merge_ids"], golden[f"hnsw_l2_ef{ef}_v1merge_dists"]
        valid = want_i >= 0
        np.testing.assert_array_equal(ids[valid], want_i[valid])
        assert close_rel(ds[valid], want_d[valid])
        np.testing.assert_array_equal(cnt, valid.sum(1))
    idx.close()


@pytest.mark.parametrize("space,D", [("l2", 128), ("cosinesimil", 100), ("negdotprod", 21), ("l1", 21)])
def test_golden_build_then_search(golden, space, D):
    idx = make_index(space, "hnsw", golden[f"f32_D{D}_base"], M=8, efConstruction=50, indexThreadQty=1)
    qs = golden[f"f32_D{D}_queries"]
    for ef in (5, 20, 200):
        idx.setQueryTimeParams(efSearch=ef)
        ids, ds, cnt = idx.knnQueryBatch(qs, 10)
        want_i, want_d = golden[f"hnsw_{space}_ef{ef}_v1merge_ids"], golden[f"hnsw_{space}_ef{ef}_v1merge_dists"]
        valid = want_i >= 0
        assert close_rel(ds[valid], want_d[valid])
        assert ids_match_modulo_ties(ids, ds, want_i, want_d) or (ids[valid] == want_i[valid]).mean() >= 0.999
    idx.close()


def test_cabi_default_ef_is_the_shims_200_and_l2_is_squared(golden):
    ext = golden["cabi_ids_in"]
    idx = make_index("l2", "hnsw", golden["f32_D128_base"], ext, M=8, efConstruction=50, indexThreadQty=1)
    ids, ds, cnt = idx.knnQueryBatch(golden["f32_D128_queries"], 10)
    np.testing.assert_array_equal(ids, golden["cabi_hnsw_l2_ids"])            # external ids, efSearch=200
    assert close_rel(ds, golden["cabi_hnsw_l2_dists"])                        # squared L2
    # ... while nmslib_get_distance stays sqrt (nmslib_c.cpp:1166): both quirks kept
    d01 = idx.getDistance(0, 1)
    assert abs(d01 - orc.space_distance("l2", golden["f32_D128_base"][0], golden["f32_D128_base"][1])) < 1e-4
    idx.close()


def test_generic_path_angular_and_u8(golden):
    idx = make_index("angulardist", "hnsw", golden["f32_D21_base"], M=8, efConstruction=50, indexThreadQty=1)
    idx.setQueryTimeParams(efSearch=20)
    ids, ds, cnt = idx.knnQueryBatch(golden["f32_D21_queries"], 10)
    want_d, want_i = golden["hnsw_angulardist_dists"], golden["hnsw_angulardist_ids"]
    # d = acos(s): the similarity within 4 ulp everywhere; 1e-5 relative on d where acos is well conditioned
    assert sim_close(np.cos(ds.astype(np.float64)), np.cos(want_d.astype(np.float64)), ulps=4)
    mid = (want_d > 0.3) & (want_d < 2.8)
    assert close_rel(ds[mid], want_d[mid], rtol=1e-5, atol=0.0)
    bad = ids_match_modulo_near_ties(ids, want_i, np.cos(want_d.astype(np.float64)), lambda v: 4 * ULP1, np.cos(ds.astype(np.float64)))
    assert not bad, bad[:5]
    idx.close()
    idx = make_index("l2sqr_sift", "hnsw", golden["u8_base"], M=8, efConstruction=50, indexThreadQty=1)
    idx.setQueryTimeParams(efSearch=150)
    ids, ds, cnt = idx.knnQueryBatch(golden["u8_queries"], 100)
    np.testing.assert_array_equal(ds, golden["hnsw_l2sqr_sift_dists"])        # integer distances: exact
    assert ids_match_modulo_ties(ids, ds, golden["hnsw_l2sqr_sift_ids"], golden["hnsw_l2sqr_sift_dists"])
    idx.close()


@pytest.mark.parametrize("space", ["l2", "cosinesimil", "negdotprod"])
def test_oracle_parity_same_graph_with_counters(space):
    """20k rows, M=16: ids, distances AND the work counters (distance computations, expansions)
    equal the oracle's on the same graph -- the kernel walks the same path."""
    import torch
    n, D, nq = 20000, 128, 256
    X, Q = refio.s_lowrank(n, D, 51), refio.s_lowrank(nq, D, 52)
    idx = make_index(space, "hnsw", X, M=16, efConstruction=100, indexThreadQty=1)
    g = orc.HnswGraph.build(space, X, 16, 100)
    for ef, k in ((32, 10), (128, 10), (200, 100)):
        idx.setQueryTimeParams(efSearch=ef)
        dq = torch.from_numpy(Q).cuda()
        d_ids = torch.empty((nq, k), dtype=torch.int32, device="cuda")
        d_ds = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        d_cnt = torch.empty((nq,), dtype=torch.int32, device="cuda")
        idx.knn_device(dq.data_ptr(), nq, D, k, d_ids.data_ptr(), d_ds.data_ptr(), d_cnt.data_ptr(),
                       torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        opos, odist, ocnt, ondc, ohops = g.search(Q, k, ef)
        ids, ds = d_ids.cpu().numpy(), d_ds.cpu().numpy()
        assert (ids == opos).mean() >= 0.999
        assert close_rel(ds, odist)
        ndc, hops, hops_up = (x.astype(np.int64) for x in idx.read_counters(nq))
        assert np.mean(ndc == ondc) >= 0.98 and abs(ndc.mean() / ondc.mean() - 1) < 0.01
        assert np.mean(hops == ohops) >= 0.98
    idx.close()


def test_visited_table_overflow_falls_back_to_bitset_not_cpu():
    """A tiny graph degree with a huge ef fills the LDS hash: the engine re-runs on the HBM bitset
    variant of the same kernel; results still equal the oracle."""
    n, D, nq = 30000, 32, 40
    X, Q = refio.s_gauss(n, D, 61), refio.s_gauss(nq, D, 62)
    idx = make_index("l2", "hnsw", X, M=6, efConstruction=40, indexThreadQty=1)
    g = orc.HnswGraph.build("l2", X, 6, 40)
    for ef in (700, 1000):
        idx.setQueryTimeParams(efSearch=ef, algoType="v1merge")
        ids, ds, cnt = idx.knnQueryBatch(Q, 10)
        opos, odist, _, _, _ = g.search(Q, 10, ef)
        assert (ids == opos).mean() >= 0.999 and close_rel(ds, odist)
    idx.close()


def test_multithreaded_build_recall_200k():
    """Default (all-thread) build at 200k rows: recall@10 against exact GPU brute force, NMSLIB's
    recall definition (eval_results.h:122-130)."""
    n, D, nq, k = 200_000, 128, 512, 10
    X, Q = refio.s_lowrank(n, D, 42), refio.s_lowrank(nq, D, 43)
    bf = make_index("l2", "seq_search", X)
    gt_i, gt_d, _ = bf.knnQueryBatch(Q, 32)
    bf.close()
    idx = make_index("l2", "hnsw", X, M=16, efConstruction=200)
    idx.setQueryTimeParams(efSearch=128)
    ids, ds, cnt = idx.knnQueryBatch(Q, k)
    rec = refio.recall_nmslib(ids, gt_i, gt_d ** 2, k)
    assert rec >= 0.97, rec                      # the reference reaches 0.976 at 200k / ef=200
    assert np.all(np.diff(ds, axis=1) >= 0)
    idx.close()


def test_wide_level0_lists_m32_and_m48_same_graph_parity():
    """M >= 32 (maxM0 = 2M > 62): level-0 lists span two words per lane.  Same graph as the reference
    (indexThreadQty=1 host build) -> ids, distances and counters equal the oracle's; the GPU builder
    reaches the same recall."""
    n, D, nq, k = 12000, 32, 128, 10
    X, Q = refio.s_gauss(n, D, 71), refio.s_gauss(nq, D, 72)     # iid data: lists fill up to maxM0
    for M in (32, 48):
        idx = make_index("l2", "hnsw", X, M=M, efConstruction=120, indexThreadQty=1)
        g = orc.HnswGraph.build("l2", X, M, 120)
        assert (g.links0()[:, 0] > 63).any()                        # the wide part is really exercised
        for ef in (40, 150):
            idx.setQueryTimeParams(efSearch=ef)
            ids, ds, cnt = idx.knnQueryBatch(Q, k)
            opos, odist, ocnt, ondc, ohops = g.search(Q, k, ef)
            assert (ids == opos).mean() >= 0.999 and close_rel(ds, odist)
            ndc, hops, _ = (x.astype(np.int64) for x in idx.read_counters(nq))
            assert abs(ndc.mean() / ondc.mean() - 1) < 0.01 and np.mean(hops == ohops) >= 0.97
        rec_host = (ids == opos).mean()
        idx.close()
        bf = make_index("l2", "brute_force", X)
        ei, ed, _ = bf.knnQueryBatch(Q, 2 * k)
        bf.close()
        rec = {}
        for mode in (0, 1):
            idx = make_index("l2", "hnsw", X, M=M, efConstruction=120, gpu_build=mode,
                             **({"indexThreadQty": 1} if mode == 0 else {}))
            idx.setQueryTimeParams(efSearch=60)
            ids, _, _ = idx.knnQueryBatch(Q, k)
            rec[mode] = refio.recall_nmslib(ids, ei, ed ** 2, k)
            idx.close()
        assert rec[1] >= rec[0] - 0.02, (M, rec)
    # (M = 64, maxM0 = 128: beyond two list words per lane -- since round 3 served by the chunked kernels,
    #  tests/test_gpu_hnsw_big.py::test_any_M_same_graph_same_walk)
    idx = make_index("l2", "hnsw", X[:300], M=64)
    ids, _, _ = idx.knnQueryBatch(X[:5], 1)
    assert ids[:, 0].tolist() == [0, 1, 2, 3, 4]
    idx.close()


def test_full_size_c3_1M_gpu_build_recall_and_self_queries():
    """BASELINE config 3 size (1M x 128, M=16, efConstruction=200, efSearch=128, k=10, Q=1024): the default
    (GPU-batched) build, recall against exact GPU brute force, plus properties that need no CPU scan."""
    n, D, nq, k = 1_000_000, 128, 1024, 10
    X, Q = refio.s_lowrank(n, D, 42), refio.s_lowrank(nq, D, 43)
    bf = make_index("l2", "seq_search", X)
    gt_i, gt_d, _ = bf.knnQueryBatch(Q, 32)
    bf.close()
    idx = make_index("l2", "hnsw", X, M=16, efConstruction=200)
    assert idx.stats()["build_seconds"] < 30                      # (the reference needs ~50 s on 16 threads)
    idx.setQueryTimeParams(efSearch=128)
    ids, ds, cnt = idx.knnQueryBatch(Q, k)
    assert np.all(cnt == k) and np.all(np.diff(ds, axis=1) >= 0)
    assert all(len(set(r)) == k for r in ids.tolist())
    rec = refio.recall_nmslib(ids, gt_i, gt_d ** 2, k)
    assert rec >= 0.99, rec                                       # reference graph: 0.9961 at the same efSearch
    # every returned distance is the squared L2 of that pair (spot check through nmslib_get_distance = sqrt)
    for q in (0, 500, 1023):
        d = orc.space_distance("l2", Q[q], X[int(ids[q, 0])])
        assert abs(d * d - ds[q, 0]) <= 1e-5 * max(1.0, ds[q, 0])
    # stored rows as queries find themselves first
    S = X[::3907][:256].copy()
    ids2, ds2, _ = idx.knnQueryBatch(S, 1)
    assert (ids2[:, 0] == np.arange(256) * 3907).mean() >= 0.995 and (ds2[:, 0] <= 1e-6).mean() >= 0.995
    idx.close()


def test_algotype_old_equals_reference_searchold(golden, tmp_path):
    """algoType=old runs the SearchOld kernel (hnsw_distfunc_opt.cc:46-150: candidate heap + closest queue +
    KNNQueue): on the reference's own graph file the results equal the reference's SearchOld outputs (golden)."""
    p = str(tmp_path / "ref.idx")
    golden["hnsw_l2_index_file"].tofile(p)
    idx = nz.Index.load(p, load_data=False)
    qs = golden["f32_D128_queries"]
    for ef in (5, 20, 200):          # ef=5 < k=10: the result queue outlives the closest queue (differs from V1Merge)
        idx.setQueryTimeParams(efSearch=ef, algoType="old")
        ids, ds, cnt = idx.knnQueryBatch(qs, 10)
        want_i, want_d = golden[f"hnsw_l2_ef{ef}_old_ids"], golden[f"hnsw_l2_ef{ef}_old_dists"]
        valid = want_i >= 0
        np.testing.assert_array_equal(ids[valid], want_i[valid])
        assert close_rel(ds[valid], want_d[valid])
        np.testing.assert_array_equal(cnt, valid.sum(1))
    idx.close()


@pytest.mark.parametrize("space,D", [("l2", 128), ("cosinesimil", 100), ("negdotprod", 21), ("l1", 21)])
def test_golden_build_then_search_old(golden, space, D):
    idx = make_index(space, "hnsw", golden[f"f32_D{D}_base"], M=8, efConstruction=50, indexThreadQty=1)
    qs = golden[f"f32_D{D}_queries"]
    for ef in (5, 20, 200):
        idx.setQueryTimeParams(efSearch=ef, algoType="old")
        ids, ds, cnt = idx.knnQueryBatch(qs, 10)
        want_i, want_d = golden[f"hnsw_{space}_ef{ef}_old_ids"], golden[f"hnsw_{space}_ef{ef}_old_dists"]
        valid = want_i >= 0
        assert close_rel(ds[valid], want_d[valid])
        assert ids_match_modulo_ties(ids, ds, want_i, want_d) or (ids[valid] == want_i[valid]).mean() >= 0.999
    idx.close()


def test_searchold_oracle_parity_with_counters_and_no_limits():
    """SearchOld against the oracle's SearchOld on the same graph: ids, distances and the work counters; ef beyond
    the V1Merge kernel's 1024 (HBM-resident candidate heap), k > ef, and the hybrid dispatch at ef >= 1000
    (Hnsw::Search, hnsw.cc:717-746)."""
    n, D, nq = 20000, 64, 128
    X, Q = refio.s_lowrank(n, D, 81), refio.s_lowrank(nq, D, 82)
    idx = make_index("l2", "hnsw", X, M=16, efConstruction=100, indexThreadQty=1)
    g = orc.HnswGraph.build("l2", X, 16, 100)
    for ef, k, algo in ((20, 10, "old"), (100, 10, "old"), (10, 50, "old"), (1500, 10, "old"), (1000, 10, "hybrid"),
                        (3000, 2500, "hybrid")):
        idx.setQueryTimeParams(efSearch=ef, algoType=algo)
        ids, ds, cnt = idx.knnQueryBatch(Q, k)
        opos, odist, ocnt, ondc, ohops = g.search(Q, k, ef, algo="old")
        np.testing.assert_array_equal(cnt, ocnt)
        assert (ids == opos).mean() >= 0.999, (ef, k, (ids == opos).mean())
        valid = opos >= 0                                    # k > found: both sides pad with -1 / +inf
        assert close_rel(ds[valid], odist[valid]) and np.isinf(ds[~valid]).all()
        ndc, hops, _ = (x.astype(np.int64) for x in idx.read_counters(nq))
        assert np.mean(ndc == ondc) >= 0.98 and abs(ndc.mean() / ondc.mean() - 1) < 0.01, (ef, k)
        assert np.mean(hops == ohops) >= 0.98, (ef, k)
    # hybrid below 1000 stays on V1Merge: counters of the V1Merge oracle
    idx.setQueryTimeParams(efSearch=999, algoType="hybrid")
    ids, ds, cnt = idx.knnQueryBatch(Q, 10)
    opos, odist, _, ondc, _ = g.search(Q, 10, 999, algo="v1merge")
    assert (ids == opos).mean() >= 0.999
    idx.close()


def test_searchold_exact_under_heavy_ties_u8():
    """Integer distances on a 3-symbol alphabet + duplicated rows: equal keys everywhere.  The candidate queue's pop
    order among equal keys is the binary heap's (libstdc++ push_heap/pop_heap), reproduced move for move, so ids
    AND counters equal the oracle exactly."""
    rng = np.random.default_rng(7)
    U = (rng.integers(0, 3, (6000, 128)) * 40).astype(np.uint8)
    U[500:560] = U[3]
    UQ = (rng.integers(0, 3, (48, 128)) * 40).astype(np.uint8)
    UQ[0] = U[3]
    idx = make_index("l2sqr_sift", "hnsw", U, M=8, efConstruction=60, indexThreadQty=1)
    g = orc.HnswGraph.build("l2sqr_sift", U, 8, 60)
    for ef, k in ((30, 10), (200, 100), (1200, 20)):
        idx.setQueryTimeParams(efSearch=ef, algoType="old")
        ids, ds, cnt = idx.knnQueryBatch(UQ, k)
        opos, odist, ocnt, ondc, ohops = g.search(UQ, k, ef, algo="old")
        np.testing.assert_array_equal(ds, odist)
        np.testing.assert_array_equal(ids, opos)
        ndc, hops, _ = (x.astype(np.int64) for x in idx.read_counters(len(UQ)))
        np.testing.assert_array_equal(ndc, ondc)
        np.testing.assert_array_equal(hops, ohops)
    idx.close()


def test_zig_batch_call_sequence_replayed_in_c(tmp_path):
    """zig/nmslib_gpu_batch.zig cannot be compiled here (no Zig toolchain): its exact C call sequence -- lib.zig's
    create-then-add order, nmslib_initialize_pool, ONE nmslib_knn_query_batch over a flat copy with caller-owned
    buffers, dense and packed uint8 -- is replayed by a plain C program against the library and must equal the
    reference's per-query sequence (lib.zig:889-931)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "zig_batch_sequence_check")
    libdir = os.path.join(root, "nmslib_zig_amd")
    subprocess.check_call(["gcc", "-std=c11", "-O1", os.path.join(root, "tests", "zig_batch_sequence_check.c"), "-o", exe,
                           "-L" + libdir, "-lnmslib_c", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "zig batch sequence ok" in out.stdout, out.stdout + out.stderr
