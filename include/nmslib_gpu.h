/*
 * nmslib_gpu.h -- device-resident extensions of the nmslib_c.h boundary.
 *
 * The reference's hot entry points (nmslib_knn_query_fill / _batch,
 * nmslib_c.cpp:941-1031) take host pointers and per-query result structs.  A
 * caller that already keeps its query batch in HBM (a serving loop, bench.py,
 * the multi-GPU shard merge) uses these entry points instead: same index handle,
 * same semantics, but plain *device* pointers in and out and an explicit HIP
 * stream, so nothing crosses PCIe inside the call.  No torch / C++ types in any
 * signature: pointers, sizes, and a `void*` that is a hipStream_t.
 *
 * Results layout: ids [query_count][k] int32 (external ids, -1 padding) and
 * distances [query_count][k] float32 (+inf padding), ascending per query, exactly
 * what nmslib_knn_query_fill would write into each nmslib_result_t
 * (extract_knn_results, nmslib_c.cpp:293-328); counts [query_count] int32.
 */
#ifndef NMSLIB_GPU_H
#define NMSLIB_GPU_H

#include "nmslib_c.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Number of HIP devices visible; 0 when the runtime finds none. */
int nmslib_gpu_device_count(void);

/* Make sure rows are resident in HBM and the index is built (what
 * nmslib_initialize_pool does lazily).  Returns the usual error codes. */
nmslib_error_t nmslib_gpu_finalize(nmslib_index_handle_t index);

/* Batched k-NN with queries already in device memory.
 *   d_queries : [query_count][elem_count] float32 (dense) or uint8 (l2sqr_sift), row-major
 *   d_ids / d_dists / d_counts : outputs in device memory (d_counts may be NULL)
 *   stream    : hipStream_t (NULL = default stream); the call only enqueues work -- with ONE exception: HNSW queries
 *               that the reference answers with SearchOld (algoType=old, or the default hybrid with efSearch >= 1000,
 *               hnsw.cc:724) read a per-query status word back and wait for the stream before returning (queues that
 *               outgrew their workspace are re-run with larger ones, at most twice).  Every other path -- brute force,
 *               both fast paths, SearchV1Merge incl. its visited-table overflow -- never blocks the caller.
 * Replaces the per-query loop of nmslib_knn_query_batch (nmslib_c.cpp:1015-1023). */
nmslib_error_t nmslib_gpu_knn_query_batch_device(nmslib_index_handle_t index,
                                                 const void* d_queries, size_t query_count,
                                                 size_t elem_count, size_t k, int32_t* d_ids,
                                                 float* d_dists, int32_t* d_counts,
                                                 void* stream);

/* Per-query work counters of the most recent HNSW batch on this index (device
 * pointers, valid until the next batch; NULL for brute force):
 *   ndc  = distance computations (the DIST_CALC counters of hnsw_distfunc_opt.cc:76-78,128-130)
 *   hops = level-0 expansions, hops_up = upper-level adjacency reads. */
nmslib_error_t nmslib_gpu_last_batch_counters(nmslib_index_handle_t index,
                                              const int32_t** d_ndc, const int32_t** d_hops,
                                              const int32_t** d_hops_up);

/* Merge per-shard top-k lists (after an all-gather across ranks): for each query,
 * `nshards` ascending lists of k (distance, id) pairs -> the k smallest by
 * (distance, id).  Layout in: [nshards][query_count][k]; out: [query_count][k]. */
nmslib_error_t nmslib_gpu_merge_topk(const float* d_dists_in, const int32_t* d_ids_in,
                                     size_t nshards, size_t query_count, size_t k,
                                     float* d_dists_out, int32_t* d_ids_out, void* stream);

/* Same merge for per-shard lists that are not back to back: shard s starts at d_dists_in + s*shard_stride and
 * d_ids_in + s*shard_stride (elements).  Lets ONE all-gather move ids and distances together: every rank
 * contributes a packed [2][query_count][k] block (ids, then the distances' bit patterns), the gathered buffer is
 * [nshards][2][query_count][k] and shard_stride = 2*query_count*k. */
nmslib_error_t nmslib_gpu_merge_topk_strided(const float* d_dists_in, const int32_t* d_ids_in,
                                             size_t shard_stride, size_t nshards, size_t query_count,
                                             size_t k, float* d_dists_out, int32_t* d_ids_out,
                                             void* stream);

/* Live timing of the dominant kernel (bf_select_* for brute force, hnsw_search for HNSW) with
 * HIP events recorded on the stream the kernel is launched on.  enable != 0 starts recording
 * (one event pair per batch); a call with total_ms / launches non-NULL waits for the recorded
 * events, returns their summed duration and launch count, and clears the record. */
nmslib_error_t nmslib_gpu_kernel_timing(nmslib_index_handle_t index, int enable, double* total_ms,
                                        uint64_t* launches);

/* Engine statistics of the last finalize/build (host values). */
typedef struct {
    double upload_seconds;   /* host -> HBM copy of the rows            */
    double build_seconds;    /* index construction (HNSW graph)        */
    size_t hbm_bytes;        /* bytes resident in HBM for this index   */
    size_t rows;
    size_t dim;
    size_t shards;         /* row shards behind this handle (index parameter gpu_shards); 1 = single GPU */
    size_t last_path;      /* selection path of the last brute-force batch: 0 adaptive f32 MFMA, 1 bf16 fast path,
                              2 adaptive int8, 3 int8 fast path; 4 = HNSW */
    /* fast paths, last slice of the last batch (reading them waits for the stream): query tiles in the slice, tiles the
       threshold kernel sent through the split-bf16-product scan instead of the one-product scan (float rows only), and
       tiles whose verification failed and were redone by the adaptive kernel */
    size_t fast_tiles;
    size_t fast_tiles_precise;
    size_t fast_tiles_fallback;
    /* HNSW with the LDS visited table, last batch (reading it waits for the stream): queries whose table filled up and
       that the HBM-bitset kernel answered again.  0 on every BASELINE configuration. */
    size_t hnsw_redone;
} nmslib_gpu_stats_t;
nmslib_error_t nmslib_gpu_get_stats(nmslib_index_handle_t index, nmslib_gpu_stats_t* out);

#ifdef __cplusplus
}
#endif
#endif /* NMSLIB_GPU_H */
