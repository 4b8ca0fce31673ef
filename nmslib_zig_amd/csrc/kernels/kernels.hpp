// Host-callable launchers of the gfx950 kernels.  Plain pointers + hipStream_t; every
// function only enqueues work on `stream` and returns the launch status.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

namespace gfxknn {

// Space codes (host and device).  Names: include/factory/init_spaces.h:77-86,120.
enum SpaceCode : int {
    SP_L2 = 0,
    SP_L1 = 1,
    SP_LINF = 2,
    SP_COSINE = 3,
    SP_ANGULAR = 4,
    SP_NEGDOT = 5,
    SP_L2SQR_SIFT = 6,
    // search-time variants of the optimized HNSW index (hnsw.cc:70-102): squared L2 and
    // cosine over pre-normalised rows
    SP_L2SQR = 7,
    SP_NORMCOS = 8,
};

// ---- geometry shared by host and device ------------------------------------------------
constexpr int BF_TQ = 128;     // queries per workgroup (4 waves x 32)
constexpr int BF_BN = 64;      // base rows staged per step
constexpr int BF_KC = 128;     // K-chunk (floats) staged per step
constexpr int BF_MAX_K = 512;  // largest k the selection kernels support

struct BfPlan {
    int nq, qpad, nqt;         // queries, padded to BF_TQ, number of query tiles
    int n;                     // base rows
    int ldb;                   // row stride (floats for f32, bytes for u8)
    int nsplit, rows_per_split;
    int kprime, cap;           // per-(query,split) survivors / candidate buffer capacity
    int p2max;                 // power of two >= nsplit*kprime (rerank sort size)
    int xj, xm;                // counted-bound exchange between splits (bf_kernels.hip); xj = 0: off
    size_t lds_select, lds_rerank;
};
// Fills every field of the plan from (n, dim, nq, k).  is_u8 selects the integer path.
BfPlan bf_make_plan(int n, int dim, int nq, int k, bool is_u8, int qpad_multiple = BF_TQ);
inline size_t bf_cand_elems(const BfPlan& p) { return (size_t)p.qpad * p.nsplit * p.cap; }
// per-(query,split) survivor counts, then the per-query shared thresholds (gthr), then the
// per-(query,split) counted bounds (gq)
// (a second gthr/gq region behind the first: the exact-tail selection of the verified l2 path gets its own, so that one
//  clear at the start of a batch serves both selections -- see bf_f32_prep_kernel)
inline size_t bf_cnt_elems(const BfPlan& p) { return 3 * (size_t)p.qpad * p.nsplit + 2 * (size_t)p.qpad; }

// Row padding for the f32 device copy: multiple of 8 floats (two 16-byte half-wave loads).
inline int f32_row_stride(int dim) { return (dim + 7) & ~7; }

// ---- preparation -------------------------------------------------------------------------
// rowaux per space: L2 -> -0.5*||b||^2 ; COSINE/ANGULAR -> 1/||b|| (0 when ||b||^2 < 2*FLT_MIN);
// NEGDOT -> unused (0).
hipError_t launch_row_aux_f32(const float* base, int n, int ldb, int dim, int space, float* aux,
                              hipStream_t s);
// u8 brute force: rows_i8 [n_pad][128] = re-centred copy (x ^ 0x80), aux [n_pad] = 256*sum(a) - sum(a^2);
// n_pad = bf_u8_rows_padded(n), pad rows are zero with an aux that can never be selected (see bf_select_u8)
inline int bf_u8_rows_padded(int n) { return (n + BF_BN - 1) / BF_BN * BF_BN + BF_BN; }
hipError_t launch_prepare_u8(const uint8_t* base, int n, uint8_t* rows_i8, int32_t* aux, int32_t* auxh, hipStream_t s);
// Copy [rows][dim] -> [rows_pad][ld] with zero fill (elem = 4 or 1 bytes).
hipError_t launch_pad_rows(const void* src, int rows, int dim, void* dst, int rows_pad, int ld,
                           int elem_bytes, hipStream_t s);
// In-place L2 normalisation of rows (hnsw.cc:441-446, hnsw.h:486-497).
hipError_t launch_normalize_rows(float* rows, int n, int ld, int dim, hipStream_t s);

// Column sums in f64: stats[0..ldb) = per-column sums, stats[ldb] = sum of squares of all elements.
hipError_t launch_col_stats(const float* base, int n, int ldb, int dim, double* stats, hipStream_t s);
// dst = src - mean (columns < dim of the first rows_valid rows); used for the centred L2 selection copy and its queries.
hipError_t launch_center_rows(const float* src, const float* mean, int rows, int rows_valid, int ld, int dim, float* dst,
                              hipStream_t s);

// ---- brute force: selection (MFMA) + exact re-rank ---------------------------------------
// qaux_cosc != NULL (cosine / angular only): base and queries are CENTRED copies, aux holds three planes
// (launch_row_aux_cosc) and qaux_cosc the per-query constants (launch_query_aux_cosc).
hipError_t launch_bf_select_f32(const BfPlan& p, int space, const float* base, const float* aux,
                                const float* queries_padded, const float* qaux_cosc, unsigned long long* cand,
                                int* cand_cnt, hipStream_t s);
hipError_t launch_row_aux_cosc(const float* orig, const float* centred, int n, int ldb, int dim, double mu_norm, float* aux,
                               hipStream_t s);
hipError_t launch_query_aux_cosc(const float* orig, const float* centred, int nq, int ldb, int dim, double mu_norm,
                                 float* qaux, hipStream_t s);
hipError_t launch_bf_select_u8(const BfPlan& p, const uint8_t* base_i8, const int32_t* aux,
                               const uint8_t* queries_padded, unsigned long long* cand,
                               int* cand_cnt, hipStream_t s);
// ... over every tile_stride-th 64-row tile only, and/or only for the query-tile groups flagged in tile_fail
hipError_t launch_bf_select_u8_ex(const BfPlan& p, const uint8_t* base_i8, const int32_t* aux,
                                  const uint8_t* queries_padded, unsigned long long* cand, int* cand_cnt,
                                  int tile_stride, const int* tile_fail, int fail_group, hipStream_t s, bool cleared = false);

// uint8 fast path for large batches (bf_kernels.hip: sample pass -> fixed-threshold scan -> list re-rank with
// verification -> adaptive fallback for flagged tile groups).  Exact like the adaptive path.
struct BfU8Fast {
    bool use;
    int qg;                    // query groups of 32 per wave: a workgroup serves 128 * qg queries
    int qpad, nqt;             // queries padded to 128 * qg; scan query tiles
    int stride, r;             // sample = every stride-th tile; threshold = r-th best score of the sample
    int nsplit, tps, caph;     // scan: row splits, 64-row tiles per split, list capacity per (query, split, half)
    int p2max;
    size_t lds_scan, lds_thr, lds_rerank;
    int s_nsplit, s_tps;       // sample pass: splits and sample tiles per split
    BfPlan fallback;           // plan of the adaptive kernel for the fallback
};
inline size_t bf_u8_top8_elems(const BfU8Fast& f) { return (size_t)f.qpad * f.s_nsplit * 2 * 8; }
BfU8Fast bf_u8_fast_plan(int n, int nq, int k);
inline size_t bf_u8_list_elems(const BfU8Fast& f) { return (size_t)f.qpad * f.nsplit * 2 * f.caph; }
inline size_t bf_u8_listcnt_elems(const BfU8Fast& f) { return (size_t)f.qpad * f.nsplit * 2; }
hipError_t launch_bf_u8_fast(const BfU8Fast& f, int n, int nq, int k, const uint8_t* base_u8, const uint8_t* base_i8,
                             const int32_t* aux, const int32_t* auxh, const uint8_t* queries_padded,
                             int* top8, unsigned long long* cand_fb, int* cnt_fb,
                             int* thr, uint32_t* list, int* list_cnt, int* tile_fail, const int32_t* ext_ids,
                             int32_t* out_ids, float* out_dists, int32_t* out_cnt, hipEvent_t scan_begin,
                             hipEvent_t scan_end, hipStream_t s, const uint8_t* queries_raw);

hipError_t launch_bf_select_f32_ex(const BfPlan& p, int space, const float* base, const float* aux,
                                   const float* queries_padded, const float* qaux_cosc, unsigned long long* cand,
                                   int* cand_cnt, const int* tile_fail, int fail_group, hipStream_t s, bool cleared = false);

// f32 fast path for large batches at D <= 128 (bf_kernels.hip: split-bf16 MFMA selection with sample-fixed thresholds,
// exact f32 re-rank with verification, adaptive fallback).  Exact like the adaptive path.
struct BfF32Fast {
    bool use;
    int mode;                  // 0 l2, 1 negdotprod (and centred cosine / angular, see cosc), 2 cosine / angular (uncentred)
    int qg;                    // a scan workgroup serves 256 * qg queries (= one query tile)
    bool cosc;                 // centred cosine / angular: mode 1 over augmented rows of sel_dim = dim + 3 columns
    int sel_dim;               // columns of the selection rows / queries (dim, or dim + 3)
    int kch;                   // rows longer than 128: chunks of 128 dimensions per row (1 = the classic shape)
    int dp;                    // 128 * kch: row length of the bf16 tiles
    int tq;                    // queries per scan workgroup / query tile: 256 * qg, or 128 when kch > 1
    int qpad, nqt;             // queries padded to tq; scan query tiles
    int stride, r;
    int rcap;                  // one-product scan: its threshold may sit as low as the rcap-th best sample score
    bool force_precise;        // NMSLIB_GPU_F32_TERMS=3: every tile through the split-product scan
    int nsplit, tps, caph, p2max;
    int s_nsplit, s_tps;
    size_t lds_scan, lds_scan1, lds_thr, lds_rerank;
    BfPlan fallback;
};
BfF32Fast bf_f32_fast_plan(int n, int dim, int nq, int k, int space, bool cosine_centred);
// workspace behind `thr`: [qpad] split-product thresholds, [qpad] one-product thresholds; behind `tile_fail`: [nqt]
// fallback flags, [nqt] precise flags
inline size_t bf_f32_thr_bytes(const BfF32Fast& f) { return (size_t)f.qpad * 8 + (size_t)f.nqt * 8 + 64; }
inline size_t bf_f32_list_elems(const BfF32Fast& f) { return (size_t)f.qpad * f.nsplit * 2 * f.caph; }
inline size_t bf_f32_listcnt_elems(const BfF32Fast& f) { return (size_t)f.qpad * f.nsplit * 2; }
inline size_t bf_f32_top8_elems(const BfF32Fast& f) { return (size_t)f.qpad * f.s_nsplit * 2 * 8; }
inline int bf_f32_rows_padded(int n) { return (n + BF_BN - 1) / BF_BN * BF_BN + BF_BN; }
// rows (or queries) -> bf16 hi / lo tiles [rows_pad][128]; auxp [rows_pad] = aux, aux_pad behind `rows` (optional)
// largest row norm -> *out (device float)
// out (16 bytes): [0] largest norm, [1] largest bf16 residual, [2] largest |element|, [3] largest fp16 residual of scale16 * row
hipError_t launch_row_maxnorm(const float* rows, int n, int ld, int dim, bool relative_residual, float* out, hipStream_t s,
                              float scale16 = 0.f);
// hi / lo / auxp: bf16 tiles of the split-product scan (nullable); h16 / auxp16: fp16(scale * row) tiles and start values
// (aux * aux16_mul) of the one-product scan (nullable)
hipError_t launch_split_bf16(const float* src, int rows, int rows_pad, int ld, int dim, void* hi, void* lo,
                             const float* aux, float aux_pad, float* auxp, hipStream_t s, int dp = 128, void* h16 = nullptr,
                             float scale = 1.f, float* auxp16 = nullptr, float aux16_mul = 1.f);
// the one-product scan's side of a float fast-path batch: fp16 tiles of the rows (built at finalize), their start values,
// the power-of-two scale, the rows' largest fp16 residual (scaled units), and the workspace of the batch's fp16 queries
struct BfF16Side {
    const void* base_h16;
    const float* auxp16;
    void* q_h16;
    float scale;       // rows
    float bres16;
    float scale_q;     // queries (l2: the rows' scale -- the start values carry the product; centred cosine: rows are divided by
                       // their norm, queries are not)
};
hipError_t launch_bf_f32_fast(const BfF32Fast& f, int space, int n, int dim, int ldb, int nq, int k, const float* base_orig,
                              const float* sel_rows, const float* aux, const void* base_hi, const void* base_lo,
                              const float* auxp, float bmax, float bres, const float* queries_orig, const float* queries_sel,
                              void* q_hi, void* q_lo, float* top8, unsigned long long* cand_fb, int* cnt_fb, float* thr,
                              uint32_t* list, int* list_cnt, int* tile_fail, int* flags_fb, const int32_t* ext_ids, int32_t* out_ids,
                              float* out_dists, int32_t* out_cnt, hipEvent_t scan_begin, hipEvent_t scan_end,
                              hipStream_t s, const float* queries_raw = nullptr, float* queries_pad_out = nullptr,
                              const float* qaux_cosc = nullptr, const float* queries_centred = nullptr, int sel_ld = 0,
                              const BfF16Side& h16 = BfF16Side{});
// centred cosine / angular on the fast path: augmented rows / queries (see row_aug_cosc_kernel)
hipError_t launch_row_aug_cosc(const float* orig, const float* centred, int n, int ldb, int dim, double mu_norm, float lambda,
                               float* out, int ldo, int* zero_rows, hipStream_t s);
hipError_t launch_query_aug_cosc(const float* centred, const float* qaux, int nq, int qpad, int ldb, int dim, float lambda,
                                 float* out, int ldo, hipStream_t s);

// The adaptive f32 path end to end (selection, re-rank; l2: verification + exact tail).  flags: [p.nqt] ints.
hipError_t launch_bf_adaptive_f32(const BfPlan& p, int space, int dim, int k, const float* base_orig, const float* sel_rows,
                                  const float* aux, const float* queries_orig, const float* queries_sel,
                                  const float* qaux_cosc, float bmax, unsigned long long* cand, int* cand_cnt, int* flags,
                                  const int32_t* ext_ids, int32_t* out_ids, float* out_dists, int32_t* out_cnt,
                                  const int* gate, int gate_tiles, hipStream_t s, bool cleared = false);
hipError_t launch_bf_rerank_verify(const BfPlan& p, int space, int dim, int k, const void* base,
                                   const void* queries_padded, const unsigned long long* cand,
                                   const int* cand_cnt, const int32_t* ext_ids, int32_t* out_ids,
                                   float* out_dists, int32_t* out_cnt, const int* tile_fail, int fail_queries,
                                   int* verify_flags, const float* queries_sel, float bmax, hipStream_t s);
hipError_t launch_bf_select_direct_f32_ex(const BfPlan& p, int space, const float* base, const float* queries_padded,
                                          unsigned long long* cand, int* cand_cnt, const int* tile_fail, int fail_group,
                                          hipStream_t s, bool cleared_second_region = false);
// Direct (VALU) selection for spaces with no inner-product form (l1, linf).
hipError_t launch_bf_select_direct_f32(const BfPlan& p, int space, const float* base,
                                       const float* queries_padded, unsigned long long* cand,
                                       int* cand_cnt, hipStream_t s);
// Exact distances of the survivors in the reference's formula, (dist, position) order, top k.
hipError_t launch_bf_rerank(const BfPlan& p, int space, int dim, int k, const void* base,
                            const void* queries_padded, const unsigned long long* cand,
                            const int* cand_cnt, const int32_t* ext_ids, int32_t* out_ids,
                            float* out_dists, int32_t* out_cnt, hipStream_t s);

hipError_t launch_bf_rerank_ex(const BfPlan& p, int space, int dim, int k, const void* base,
                               const void* queries_padded, const unsigned long long* cand,
                               const int* cand_cnt, const int32_t* ext_ids, int32_t* out_ids,
                               float* out_dists, int32_t* out_cnt, const int* tile_fail, int fail_queries,
                               hipStream_t s);

// one pair, one wave (nmslib_get_distance)
hipError_t launch_pair_distance(int space, const void* a, const void* b, int dim, float* out,
                                hipStream_t s);

// ---- HNSW search -------------------------------------------------------------------------
struct HnswDeviceGraph {
    const void* rows;          // f32 [n][ldv] or u8 [n][128]
    const int32_t* row_norm;   // u8: sum of squares per row
    const int32_t* links0;     // [n][maxM0+1]  (count, ids...)
    const int64_t* up_off;     // [n] offset into up_links (ints) or -1
    const int32_t* up_links;   // per node: level blocks of (maxM+1) ints
    const int32_t* ext_ids;    // internal position -> external id
    int n, dim, ldv;
    int maxM, maxM0, maxlevel, enterpoint;
    int space;                 // search-time SpaceCode (SP_L2SQR, SP_NORMCOS, ...)
    int normalize_query;       // cosine on the optimized index
};
struct HnswSearchPlan {
    int nq, k, ef, cap;        // cap = max(ef, k)
    int table_size;            // LDS visited hash entries (power of two); 0 -> global bitset
    size_t lds_bytes;
    size_t bitset_words;       // per query, when table_size == 0
    // SearchOld kernel only (hnsw_make_plan_old): candidate heap and queue placement
    int heap_lds, heap_cap;    // heap entries in LDS / in total per query (the rest lives in the HBM workspace)
    int a_in_lds, r_in_lds;    // closest-queue values (ef floats) / result queue (k pairs) in LDS?
};
// entries of the frontier arrays: a multiple of 64 that holds the longest adjacency list (level 0 or above)
inline int hnsw_nbcap(const HnswDeviceGraph& g) {
    const int longest = g.maxM0 > g.maxM ? g.maxM0 : g.maxM;
    return longest <= 62 ? 64 : (longest + 64) / 64 * 64;
}

// the LDS search kernels hold frontiers of up to this many neighbours (lists up to maxM0 = 254, i.e. M <= 127); longer lists
// go to the HBM-array kernel
constexpr int HNSW_NBCAP_LDS = 256;

HnswSearchPlan hnsw_make_plan(const HnswDeviceGraph& g, int nq, int k, int ef, bool force_bitset);
// Plan of the SearchOld kernel (hnsw_distfunc_opt.cc:46-150): no limit on ef or k.  heap_cap = 0 picks the default
// bound on the candidate heap (queries that outgrow it report status 2 and are retried with heap_cap = n).
HnswSearchPlan hnsw_make_plan_old(const HnswDeviceGraph& g, int nq, int k, int ef, bool force_bitset, int heap_cap);
// per-query HBM workspace of the SearchOld kernel, in bytes (0 = nothing spills)
inline size_t hnsw_old_ws_a(const HnswSearchPlan& p) { return p.a_in_lds ? 0 : (size_t)p.ef * 4; }
inline size_t hnsw_old_ws_r(const HnswSearchPlan& p) { return p.r_in_lds ? 0 : (size_t)p.k * 8; }
inline size_t hnsw_old_ws_heap(const HnswSearchPlan& p) { return (size_t)(p.heap_cap - p.heap_lds) * 8; }
hipError_t launch_hnsw_search_old(const HnswDeviceGraph& g, const HnswSearchPlan& p, const void* queries,
                                  uint32_t* bitset, void* ws_a, void* ws_r, void* ws_heap, int32_t* out_ids,
                                  float* out_dists, int32_t* out_cnt, int32_t* out_ndc, int32_t* out_hops,
                                  int32_t* out_hops_up, int32_t* status, hipStream_t s);
// queries: [nq][dim] f32 (row stride dim) or u8 [nq][128].  status[q] != 0 -> visited table
// overflowed (caller re-runs those with the bitset variant).
hipError_t launch_hnsw_search(const HnswDeviceGraph& g, const HnswSearchPlan& p,
                              const void* queries, uint32_t* bitset, int32_t* out_ids,
                              float* out_dists, int32_t* out_cnt, int32_t* out_ndc,
                              int32_t* out_hops, int32_t* out_hops_up, int32_t* status,
                              hipStream_t s);

// SearchV1Merge with max(ef, k) beyond the LDS kernels' 1024 items: the sorted array in a per-query HBM workspace
// (ws_keys / ws_idu: [nq][max(ef, k)]), visited set = HBM bitset [nq][ceil(n / 32)] (cleared by the caller).
hipError_t launch_hnsw_search_big(const HnswDeviceGraph& g, int nq, int k, int ef, const void* queries, uint32_t* bitset,
                                  float* ws_keys, int32_t* ws_idu, int32_t* out_ids, float* out_dists, int32_t* out_cnt,
                                  int32_t* out_ndc, int32_t* out_hops, int32_t* out_hops_up, int32_t* status, hipStream_t s);

// Visited-table overflow handled on the device (no host round trip):
//   1. LDS-table plan:  fix_slots = 0, fix_list/fix_count given -> overflowed queries are appended to fix_list;
//   2. bitset plan:     fix_slots = S > 0 -> S workgroups walk fix_list (count read on the device), each clearing and
//      using its own bitset slot (bitset must hold S * bitset_words words) and overwriting those queries' outputs.
hipError_t launch_hnsw_search_fix(const HnswDeviceGraph& g, const HnswSearchPlan& p, const void* queries,
                                  uint32_t* bitset, int fix_slots, int32_t* fix_list, int32_t* fix_count,
                                  int32_t* out_ids, float* out_dists, int32_t* out_cnt, int32_t* out_ndc,
                                  int32_t* out_hops, int32_t* out_hops_up, int32_t* status, hipStream_t s);

// Construction-mode search (hnsw_build): queries are stored rows (query_rows), the best-first phase runs on
// `level`, start_nodes[q] >= 0 gives the start node (else descend from the entry point to level+1).
hipError_t launch_hnsw_search_ex(const HnswDeviceGraph& g, const HnswSearchPlan& p, const void* queries,
                                 const int32_t* query_rows, const int32_t* start_nodes, int level,
                                 uint32_t* bitset, int32_t* out_ids, float* out_dists, int32_t* out_cnt,
                                 int32_t* out_ndc, int32_t* out_hops, int32_t* out_hops_up, int32_t* status,
                                 hipStream_t s);

// ---- HNSW construction on the GPU (hnsw_build_kernels.hip) ------------------------------------
struct HnswBuildGraph {          // mutable twin of HnswDeviceGraph
    HnswDeviceGraph g;           // rows, links0, up_off, up_links (written by the link kernels)
    int32_t* links0;             // same memory as g.links0, non-const
    int32_t* up_links;
    int M, delaunay;
};
// starts[i] = first (closest) candidate of pair src[i] (src[i] < 0 or empty -> -1 = descend from the entry point)
hipError_t launch_hnsw_build_starts(const int32_t* src, const int32_t* cand_ids, const int32_t* cand_n, int stride,
                                    int32_t* starts, int m, hipStream_t s);
// Heuristic neighbour selection for the `npts` new nodes listed in pts at `level`: reads the sorted candidates
// (cand_ids/cand_d/cand_n, stride `stride`), writes each new node's forward list, and one reverse-link request per
// selected neighbour into the node's own M slots: req_key [npts][M] = target << 32 | new node (unused: ~0), req_dist.
hipError_t launch_hnsw_build_select(const HnswBuildGraph& bg, int level, const int32_t* pts, int npts,
                                    const int32_t* cand_ids, const float* cand_d, const int32_t* cand_n,
                                    int stride, const int32_t* extra_ids, const float* extra_d,
                                    const int32_t* extra_n, unsigned long long* req_key, float* req_dist,
                                    hipStream_t s);
// Batch-mates: for every new node of the level slice, the EARLIER nodes of the slice that are closer than its worst
// candidate (at most 64, ascending): extra_ids/extra_d [npts][64], extra_n [npts].  Merged by the select kernel.
hipError_t launch_hnsw_build_mates(const HnswBuildGraph& bg, int level, const int32_t* pts, int npts,
                                   const float* cand_d, const int32_t* cand_n, int stride, int32_t* extra_ids,
                                   float* extra_d, int32_t* extra_n, hipStream_t s);
// Device radix sort of the requests by (target, new node) + the index of each target's first request (active/nactive).
size_t hnsw_build_sort_temp_bytes(int max_requests, int n);
hipError_t launch_hnsw_build_sort_requests(const unsigned long long* req_key, const float* req_dist, int total,
                                           unsigned long long* key_sorted, float* dist_sorted, void* temp,
                                           size_t temp_bytes, int32_t* active, int32_t* nactive, hipStream_t s);
// Apply the sorted reverse links target by target (addFriendlevel + shrink, hnsw.h:258-314), any number per target.
hipError_t launch_hnsw_build_link(const HnswBuildGraph& bg, int level, const int32_t* active,
                                  const int32_t* nactive, int max_active, const unsigned long long* key_sorted,
                                  const float* dist_sorted, int total, hipStream_t s);

// ---- range search on the brute-force index (range_kernels.hip) ----------------------------------
// dist_ws: [n] floats; count_ws: [ceil(n/1024) + 1] ints, the last one receives the number of matches.
// Matches (distance <= radius) are written in position order, the first `capacity` of them.
inline size_t range_count_elems(int n) { return (size_t)(n + 1023) / 1024 + 1; }
hipError_t launch_range_search(int space, const void* rows, int ld, int n, const void* query_padded, int dim,
                               float radius, const int32_t* ext_ids, float* dist_ws, int* count_ws, int capacity,
                               int32_t* out_ids, float* out_dists, hipStream_t s);

// Exact scan for k > BF_MAX_K: per query one pass with the reference formula + one stable device radix sort of
// (distance, position).  dist_ws [n] floats, key_ws [4][n] u32, temp from bf_bigk_temp_bytes(n).
size_t bf_bigk_temp_bytes(int n);
hipError_t launch_bf_bigk(int space, const void* rows, int ld, int n, const void* queries_padded, size_t query_stride_bytes,
                          int nq, int dim, int k, const int32_t* ext_ids, float* dist_ws, uint32_t* key_ws, void* temp,
                          size_t temp_bytes, int32_t* out_ids, float* out_dists, int32_t* out_cnt, hipStream_t s);

// ---- shard merge ---------------------------------------------------------------------------
// shard s's lists start at dists_in + s*shard_stride / ids_in + s*shard_stride (elements)
hipError_t launch_merge_topk(const float* dists_in, const int32_t* ids_in, size_t shard_stride, int nshards, int nq,
                             int k, float* dists_out, int32_t* ids_out, hipStream_t s);
// ... with the number of valid results per query (cnt_out, optional) and a final id map: ids in the lists are
// positions, ids_out[i] = ext_ids[position] (ext_ids optional)
hipError_t launch_merge_topk_ex(const float* dists_in, const int32_t* ids_in, size_t shard_stride, int nshards, int nq,
                                int k, float* dists_out, int32_t* ids_out, int32_t* cnt_out, const int32_t* ext_ids,
                                hipStream_t s);

}  // namespace gfxknn
