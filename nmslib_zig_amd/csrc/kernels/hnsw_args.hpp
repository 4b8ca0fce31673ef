// Kernel arguments shared by the HNSW search kernels (hnsw_kernels.hip, hnsw_mw_kernels.hip).
#pragma once
#include "common_dev.hpp"
#include "kernels.hpp"

namespace gfxknn {

struct HnswArgs {
    HnswDeviceGraph g;
    const void* queries;
    uint32_t* bitset;
    size_t bitset_words;
    int32_t* out_ids;
    float* out_dists;
    int32_t* out_cnt;
    int32_t* out_ndc;
    int32_t* out_hops;
    int32_t* out_hops_up;
    int32_t* status;
    int nq, k, ef, cap;
    int nbcap;  // SearchOld / HBM-array SearchV1Merge with wide lists: entries of the frontier arrays (hnsw_nbcap)
    int capa;  // cap rounded up to a multiple of 4 (keeps the LDS carve-up 16-byte aligned)
    // construction mode (hnsw_build_kernels.hip): the query is a stored row, the best-first phase runs
    // on `level`, and the start node is given (or found by descending from the entry point to level+1)
    const int32_t* query_rows;   // [nq] row index of each query, or NULL (external queries)
    const int32_t* start_nodes;  // [nq] start node (>= 0) or -1 = descend from the entry point; or NULL
    int level;
    int table_size, table_shift;
    int prof;  // NMSLIB_HNSW_PROF: accumulate per-phase cycles into g_hnsw_prof (experiments only)
    // visited-table overflow without the host: the LDS-table launch appends overflowed queries to fix_list; the
    // bitset launch that follows (fix_mode) walks that list, each workgroup with its own bitset slot
    int32_t* fix_list;
    int32_t* fix_count;
    int fix_mode;
    int no_pipe;  // NMSLIB_HNSW_PIPE=0: the one-wave kernel without its software pipeline (diagnostics)
};

constexpr uint32_t HT_EMPTY = 0xFFFFFFFFu;  // free slot of the LDS visited table

// hnsw_mw_kernels.hip: one workgroup per query (control wave + gather waves); sa_emax = sorted-array items per lane
// (2 or 4).  Same results, bit for bit, as hnsw_search_kernel<SPACE, false, sa_emax, false>.
hipError_t launch_hnsw_search_mw(const HnswArgs& a, size_t lds_bytes, int sa_emax, hipStream_t s);
// control-wave phase cycles accumulated since the last call (NMSLIB_HNSW_PROF), then cleared
void hnsw_mw_read_prof(unsigned long long out[12]);

}  // namespace gfxknn
