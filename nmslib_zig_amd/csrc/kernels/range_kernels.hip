// Range search on the brute-force index: RangeQuery::CheckAndAddToResult (src/rangequery.cc:67-76) keeps
// every object with distance <= radius, in scan (= insertion) order; the shim then copies the first
// `capacity` of them with distances recomputed by IndexTimeDistance (nmslib_c.cpp:1104-1113).
//   1. range_dist   : the reference distance formula for every row, one wave per row (HBM-bound:
//                     one pass over the base);
//   2. range_count  : matches per 1024-row block;  3. range_scan: exclusive scan of the block counts;
//   4. range_scatter: matches written in position order at their global rank, the first `capacity` only.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "common_dev.hpp"
#include "kernels.hpp"

namespace gfxknn {

__global__ __launch_bounds__(256) void range_dist_kernel(int space, const void* rows, int ld, int n,
                                                         const void* query, int dim, float* dist) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < n; row += gridDim.x * 4) {
        float d;
        if (space == SP_L2SQR_SIFT)
            d = (float)wave_exact_distance_u8(reinterpret_cast<const uint8_t*>(rows) + (size_t)row * 128,
                                              reinterpret_cast<const uint8_t*>(query), lane);
        else
            d = wave_exact_distance_f32(space, reinterpret_cast<const float*>(rows) + (size_t)row * ld,
                                        reinterpret_cast<const float*>(query), dim, lane);
        if (lane == 0) dist[row] = d;
    }
}

constexpr int kRangeBlockRows = 1024;

__device__ __forceinline__ int block_exclusive_scan(int v, int* total) {
    // 256 threads: wave scans + 4 wave totals through LDS
    __shared__ int wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return base + incl - v;
}

__global__ __launch_bounds__(256) void range_count_kernel(const float* dist, int n, float radius, int* block_counts) {
    const int row0 = blockIdx.x * kRangeBlockRows + threadIdx.x * 4;
    int c = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) c += (row0 + i < n && dist[row0 + i] <= radius) ? 1 : 0;
    int total;
    (void)block_exclusive_scan(c, &total);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}

// single workgroup: exclusive scan of nb block counts (in place), total -> block_counts[nb]
__global__ __launch_bounds__(256) void range_scan_kernel(int* block_counts, int nb) {
    int carry = 0;
    for (int base = 0; base < nb; base += 256) {
        const int i = base + threadIdx.x;
        const int v = i < nb ? block_counts[i] : 0;
        int total;
        const int ex = block_exclusive_scan(v, &total);
        if (i < nb) block_counts[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) block_counts[nb] = carry;
}

__global__ __launch_bounds__(256) void range_scatter_kernel(const float* dist, int n, float radius,
                                                            const int* block_offsets, const int32_t* ext_ids,
                                                            int capacity, int32_t* out_ids, float* out_dists) {
    const int row0 = blockIdx.x * kRangeBlockRows + threadIdx.x * 4;
    float d[4];
    int c = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        d[i] = row0 + i < n ? dist[row0 + i] : INFINITY;
        c += (row0 + i < n && d[i] <= radius) ? 1 : 0;
    }
    int total;
    int at = block_offsets[blockIdx.x] + block_exclusive_scan(c, &total);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (row0 + i < n && d[i] <= radius) {
            if (at < capacity) {
                out_ids[at] = ext_ids[row0 + i];
                out_dists[at] = d[i];
            }
            ++at;
        }
    }
}

hipError_t launch_range_search(int space, const void* rows, int ld, int n, const void* query_padded, int dim,
                               float radius, const int32_t* ext_ids, float* dist_ws, int* count_ws, int capacity,
                               int32_t* out_ids, float* out_dists, hipStream_t s) {
    if (n <= 0) return hipMemsetAsync(count_ws, 0, 4, s);
    int grid = (n + 3) / 4;
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(range_dist_kernel, dim3(grid), dim3(256), 0, s, space, rows, ld, n, query_padded, dim, dist_ws);
    const int nb = (n + kRangeBlockRows - 1) / kRangeBlockRows;
    hipLaunchKernelGGL(range_count_kernel, dim3(nb), dim3(256), 0, s, dist_ws, n, radius, count_ws);
    hipLaunchKernelGGL(range_scan_kernel, dim3(1), dim3(256), 0, s, count_ws, nb);
    hipLaunchKernelGGL(range_scatter_kernel, dim3(nb), dim3(256), 0, s, dist_ws, n, radius, count_ws, ext_ids,
                       capacity, out_ids, out_dists);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Exact scan for k beyond the selection kernels' capacity (k > 512; the reference has no limit on k, knnquery.cc:66-75):
// per query, the reference formula for every row (range_dist_kernel), one stable device radix sort of
// (distance, position) -- a stable sort by distance IS the canonical (distance, position) order -- and the first k.
// One pass over the base + one sort of N pairs per query: slow (~0.5 ms per query at 1M rows), exact, unlimited.
// ---------------------------------------------------------------------------------------------------------------
__global__ void bigk_keys_kernel(const float* dist, int n, uint32_t* keys, uint32_t* vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = f32_ord(dist[i]);
    vals[i] = (uint32_t)i;
}
__global__ void bigk_emit_kernel(const uint32_t* keys, const uint32_t* vals, int n, int k, const int32_t* ext_ids,
                                 int32_t* out_ids, float* out_dists, int32_t* out_cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && out_cnt) *out_cnt = k < n ? k : n;
    if (i >= k) return;
    if (i < n) {
        out_ids[i] = ext_ids ? ext_ids[vals[i]] : (int32_t)vals[i];
        out_dists[i] = ord_f32(keys[i]);
    } else {
        out_ids[i] = -1;
        out_dists[i] = INFINITY;
    }
}

size_t bf_bigk_temp_bytes(int n) {
    size_t tmp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                    (uint32_t*)nullptr, (size_t)(n > 0 ? n : 1), 0, 32, nullptr, false);
    return tmp + 256;
}

hipError_t launch_bf_bigk(int space, const void* rows, int ld, int n, const void* queries_padded, size_t query_stride_bytes,
                          int nq, int dim, int k, const int32_t* ext_ids, float* dist_ws, uint32_t* key_ws /* [4][n] */,
                          void* temp, size_t temp_bytes, int32_t* out_ids, float* out_dists, int32_t* out_cnt,
                          hipStream_t s) {
    uint32_t* k_in = key_ws;
    uint32_t* k_out = key_ws + (size_t)n;
    uint32_t* v_in = key_ws + 2 * (size_t)n;
    uint32_t* v_out = key_ws + 3 * (size_t)n;
    for (int q = 0; q < nq; ++q) {
        int32_t* oi = out_ids + (size_t)q * k;
        float* od = out_dists + (size_t)q * k;
        int32_t* oc = out_cnt ? out_cnt + q : nullptr;
        if (n > 0) {
            int grid = (n + 3) / 4;
            if (grid > 65536) grid = 65536;
            hipLaunchKernelGGL(range_dist_kernel, dim3(grid), dim3(256), 0, s, space, rows, ld, n,
                               static_cast<const char*>(queries_padded) + (size_t)q * query_stride_bytes, dim, dist_ws);
            hipLaunchKernelGGL(bigk_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dist_ws, n, k_in, v_in);
            hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, k_in, k_out, v_in, v_out, (size_t)n, 0, 32, s, false);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(bigk_emit_kernel, dim3((k + 255) / 256), dim3(256), 0, s, k_out, v_out, n, k, ext_ids, oi, od, oc);
    }
    return hipGetLastError();
}

}  // namespace gfxknn
