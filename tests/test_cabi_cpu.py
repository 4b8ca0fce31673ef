"""CPU-only checks of the drop-in boundary: the shared library loads and exports every symbol that
include/nmslib_c.h and include/nmslib_gpu.h declare, the host logic (parameters, data storage,
error codes, allocator ownership, persistence formats, graph construction) behaves like the
reference's shim, and -- without a GPU -- every k-NN entry fails loudly instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import nmslib_zig_amd as nz
from tests import refio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = nz.lib()
    declared = []
    for hdr in ("nmslib_c.h", "nmslib_gpu.h"):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        declared += re.findall(r"\b(nmslib_[a-z0-9_]+)\s*\(", txt)
    declared = sorted(set(declared))
    assert len([d for d in declared if not d.startswith("nmslib_gpu_")]) == 37
    for name in declared:
        assert hasattr(L, name), f"{name} is declared but not exported"
    assert sorted(nz.ABI_SYMBOLS_C + nz.ABI_SYMBOLS_GPU) == declared


def test_error_codes_and_null_handling():
    L = nz.lib()
    a = nz.TrackingAllocator()
    h = C.c_void_p()
    assert L.nmslib_index_create(None, None, b"hnsw", 0, 0, a.ref(), C.byref(h)) == 2
    assert L.nmslib_index_create(b"l2", None, b"hnsw", 0, 0, None, C.byref(h)) == 2
    # sparse / string / unknown spaces are not served: SPACE_INCOMPATIBLE, nothing leaked
    for space, dt in ((b"cosinesimil_sparse", 1), (b"leven", 3), (b"nonsense", 0), (b"l2", 2), (b"l2sqr_sift", 0)):
        assert L.nmslib_index_create(space, None, b"hnsw", dt, 0, a.ref(), C.byref(h)) == 5
    assert len(a.live) == 0
    assert L.nmslib_create_index(None, None, 0) == 2
    assert L.nmslib_data_qty(None) == 0
    assert L.nmslib_knn_query_fill(None, None, 0, 0, None, 0) == 2
    L.nmslib_index_destroy(None)
    L.nmslib_initialize_pool(None)


def test_last_error_detail_is_allocator_owned():
    a = nz.TrackingAllocator()
    L = nz.lib()
    h = C.c_void_p()
    assert L.nmslib_index_create(b"nonsense", None, b"hnsw", 0, 0, a.ref(), C.byref(h)) == 5
    d = nz.ErrorDetail()
    assert L.nmslib_get_last_error_detail(C.byref(d), a.ref()) == 0
    assert d.code == 5 and len(a.live) == 2
    assert b"nonsense" in C.string_at(d.message)
    L.nmslib_free_string(d.message, a.ref())
    L.nmslib_free_string(d.file, a.ref())
    assert len(a.live) == 0


def test_params_strictness_like_check_unused():
    idx = nz.Index("l2", "hnsw")
    with pytest.raises(nz.NmslibError) as e:       # AnyParamManager::CheckUnused, params.h:241-251
        idx.buildIndex(bogus=1)
    assert e.value.code == 8
    idx.buildIndex(M=8, efConstruction=50, maxM=8, maxM0=16, delaunay_type=2, post=0, indexThreadQty=1,
                   skip_optimized_index=0, searchMethod=0)
    with pytest.raises(nz.NmslibError):
        idx.setQueryTimeParams(ef=10, efSearch=20)  # synonyms, hnsw.cc:478-480
    with pytest.raises(nz.NmslibError):
        idx.setQueryTimeParams(algoType="fastest")
    idx.setQueryTimeParams(efSearch=128, algoType="v1merge")
    idx.close()
    assert len(idx.alloc.live) == 0
    bad = nz.Index("l2", "vptree")                 # method resolved at create_index, like the reference
    with pytest.raises(nz.NmslibError) as e:
        bad.buildIndex()
    assert e.value.code == 8
    bad.close()


def test_lib_zig_call_order_and_metadata():
    """lib.zig calls nmslib_create_index BEFORE pushing the rows (lib.zig:625-681); accepted."""
    idx = nz.Index("cosine", "hnsw", space_params={"dim": 4})     # alias + ignored 'dim' (lib.zig:530-533)
    idx.buildIndex()                                              # empty: parses params, returns
    X = np.eye(4, dtype=np.float32)[:3]
    L = nz.lib()
    ptrs = (C.c_void_p * 3)(*[X[i].ctypes.data for i in range(3)])
    ids = np.array([10, 20, 30], np.int32)
    assert L.nmslib_add_data_point_batch_pointers(idx.h, 0, ptrs, 3, 4, ids.ctypes.data, None) == 0
    assert idx.dataQty() == 3
    assert idx.getSpaceType() == "cosine" and idx.getMethod() == "hnsw"
    np.testing.assert_array_equal(idx.getDataPoint(1), X[1])
    with pytest.raises(nz.NmslibError) as e:
        idx.getDataPoint(10)
    assert e.value.code == 2                                      # lib.zig:1499-1515
    idx.setThreadPoolSize(4)
    assert idx.getThreadPoolSize() == 4                           # lib.zig:1518-1535
    with pytest.raises(nz.NmslibError):
        idx.setThreadPoolSize(0)
    # wrong data mode for this space
    assert L.nmslib_add_data_point_batch_pointers(idx.h, 2, ptrs, 3, 4, None, None) == 5
    assert L.nmslib_add_data_point_batch_pointers(idx.h, 1, ptrs, 3, 4, None, None) == 5
    # borrowed copy + its free function
    p, n, fn = C.c_void_p(), C.c_size_t(), C.c_void_p()
    assert L.nmslib_borrow_data_dense(idx.h, 2, C.byref(p), C.byref(n), C.byref(fn)) == 0
    got = np.frombuffer(C.string_at(p, n.value), np.float32)
    np.testing.assert_array_equal(got, X[2])
    C.CFUNCTYPE(None, C.c_void_p)(fn.value)(p)
    idx.close()
    assert len(idx.alloc.live) == 0


def test_range_query_on_hnsw_is_space_incompatible():
    idx = nz.Index("l2", "hnsw")                                  # lib.zig:1427-1455
    idx.buildIndex()
    q = np.zeros(4, np.float32)
    ids, ds = (C.c_int32 * 8)(), (C.c_float * 8)()
    r = nz.Result(ids, ds, 0, 8)
    assert nz.lib().nmslib_range_query_fill(idx.h, q.ctypes.data, 4, 1.0, C.byref(r), 0) == 5
    n = C.c_size_t()
    assert nz.lib().nmslib_range_query_get_size(idx.h, q.ctypes.data, 4, 1.0, C.byref(n), 0) == 0 and n.value == 128
    idx.close()


def test_unbuilt_index_reports_index_build_failed():
    idx = nz.Index("l2", "hnsw")
    idx.addDenseBatch(np.eye(4, dtype=np.float32))
    q = np.zeros(4, np.float32)
    ids, ds = (C.c_int32 * 8)(), (C.c_float * 8)()
    r = nz.Result(ids, ds, 0, 8)
    assert nz.lib().nmslib_knn_query_fill(idx.h, q.ctypes.data, 4, 2, C.byref(r), 0) == 8   # nmslib_c.cpp:963-967
    idx.close()


@pytest.mark.parametrize("space,D", [("l2", 128), ("cosinesimil", 100), ("negdotprod", 21), ("l1", 21)])
def test_host_graph_builder_and_index_file_match_reference(golden, tmp_path, space, D):
    """indexThreadQty=1 -> the adjacency of the reference's own deterministic build, written in the
    reference's optimized-index format (hnsw.cc:774-806).  gpu_defer=1 keeps this test GPU-free."""
    X = golden[f"f32_D{D}_base"]
    idx = nz.Index(space, "hnsw")
    idx.addDenseBatch(X)
    idx.buildIndex(M=8, efConstruction=50, indexThreadQty=1, gpu_defer=1)
    path = str(tmp_path / "idx")
    idx.save(path, True)
    P = refio.parse_optimized_index(path) if True else None
    mx, ep, maxM, maxM0, mem, off0, dfunc, smeth = (int(v) for v in golden[f"hnsw_{space}_meta"])
    assert (P["maxlevel"], P["enterpoint"], P["maxM"], P["maxM0"]) == (mx, ep, maxM, maxM0)
    assert (P["mem_per_obj"], P["off_level0"], P["dist_func"], P["search_method"]) == (mem, off0, dfunc, smeth)
    np.testing.assert_array_equal(P["levels"], golden[f"hnsw_{space}_levels"])
    np.testing.assert_array_equal(P["links0"], golden[f"hnsw_{space}_links0"])
    np.testing.assert_array_equal(P["up_off"], golden[f"hnsw_{space}_up_off"])
    np.testing.assert_array_equal(P["up_links"], golden[f"hnsw_{space}_up_links"])
    # .dat: size_t qty; {size_t len; 16-byte header + payload}  (space.cc:88-105)
    raw = open(path + ".dat", "rb").read()
    assert len(raw) == 8 + X.shape[0] * (8 + 16 + D * 4)
    idx.close()


def test_index_file_bytes_equal_reference_file_where_defined(golden, tmp_path):
    """Byte-for-byte against the reference's saved file, except the slots the reference leaves
    uninitialised (hnsw.cc:427,461-464)."""
    X = golden["f32_D128_base"]
    idx = nz.Index("l2", "hnsw")
    idx.addDenseBatch(X)
    idx.buildIndex(M=8, efConstruction=50, indexThreadQty=1, gpu_defer=1)
    path = str(tmp_path / "idx")
    idx.save(path, False)
    mine = np.fromfile(path, np.uint8)[:-32]          # minus this library's trailer
    ref = golden["hnsw_l2_index_file"]
    assert mine.size == ref.size
    refpath = str(tmp_path / "ref")
    ref.tofile(refpath)
    A, B = refio.parse_optimized_index(path), refio.parse_optimized_index(refpath)
    for k in ("ids", "datalen", "payload", "links0", "levels", "up_off", "up_links"):
        np.testing.assert_array_equal(A[k], B[k])
    assert np.array_equal(mine[:68], ref[:68])        # header
    idx.close()


def test_load_reference_index_file_roundtrip_cpu(golden, tmp_path):
    ref = golden["hnsw_l2_index_file"]
    p = str(tmp_path / "ref")
    ref.tofile(p)
    idx = nz.Index.load(p, load_data=False)
    assert idx.dataQty() == 300 and idx.getSpaceType() == "l2" and idx.getMethod() == "hnsw"
    np.testing.assert_array_equal(idx.getDataPoint(5), golden["f32_D128_base"][5])
    p2 = str(tmp_path / "again")
    idx.save(p2, True)
    A, B = refio.parse_optimized_index(p), refio.parse_optimized_index(p2)
    for k in ("ids", "payload", "links0", "levels", "up_links"):
        np.testing.assert_array_equal(A[k], B[k])
    idx.close()
    with pytest.raises(nz.NmslibError) as e:
        nz.Index.load(str(tmp_path / "missing"))
    assert e.value.code == 10


def test_u8_rows_carry_their_norm_and_sift_dim_is_enforced():
    idx = nz.Index("l2sqr_sift", "hnsw", data_type="DenseUInt8Vector", dist_type="Int")
    U = refio.s_sift_like(5, 3)
    idx.addUInt8Batch(U)
    n = C.c_size_t()
    assert nz.lib().nmslib_get_data_point_size(idx.h, 0, C.byref(n)) == 0 and n.value == 132
    buf = np.zeros(132, np.uint8)
    assert nz.lib().nmslib_get_data_point_fill(idx.h, 2, buf.ctypes.data, 132) == 0
    assert int(buf[128:].view(np.int32)[0]) == int((U[2].astype(np.int64) ** 2).sum())
    assert nz.lib().nmslib_get_data_point_fill(idx.h, 2, buf.ctypes.data, 100) == 4    # BUFFER_TOO_SMALL
    bad = np.zeros((2, 64), np.uint8)
    with pytest.raises(nz.NmslibError):                     # CHECK size == SIFT_DIM, space_l2sqr_sift.cc:138
        idx.addUInt8Batch(bad)
    idx.close()


def test_no_gpu_means_loud_failure_not_fallback():
    if nz.lib().nmslib_gpu_device_count() > 0:
        pytest.skip("a GPU is present")
    idx = nz.Index("l2", "seq_search")
    idx.addDenseBatch(np.eye(8, dtype=np.float32))
    with pytest.raises(nz.NmslibError) as e:
        idx.buildIndex()
    assert "no HIP device" in str(e.value)
    with pytest.raises(nz.NmslibError):
        idx.knnQueryBatch(np.eye(8, dtype=np.float32)[:2], 2)
    with pytest.raises(nz.NmslibError):
        idx.getDistance(0, 1)
    idx.close()


def test_cpp_host_mirror_compiles_and_links(tmp_path):
    """nmslib_zig_amd/host/nmslib.hpp (the C++ mirror of lib.zig) against the shared library."""
    import subprocess
    exe = str(tmp_path / "host_mirror_check")
    libdir = os.path.join(ROOT, "nmslib_zig_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "host_mirror_check.cpp"), "-o", exe,
                           "-L" + libdir, "-lnmslib_c", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "host mirror ok" in out.stdout, out.stdout + out.stderr
