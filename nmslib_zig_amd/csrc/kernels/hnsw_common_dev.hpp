// Device helpers shared by the HNSW search and construction kernels (wave64).
#pragma once
#include "common_dev.hpp"

namespace gfxknn {

template <int SPACE>
struct DistTraits {
    static constexpr bool kU8 = (SPACE == SP_L2SQR_SIFT);
    static constexpr bool kThree = (SPACE == SP_COSINE || SPACE == SP_ANGULAR);
    static constexpr bool kMax = (SPACE == SP_LINF);
};

template <int SPACE>
__device__ __forceinline__ void accum4(const f32x4& q, const f32x4& b, float& s0, float& s1, float& s2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if constexpr (SPACE == SP_L2SQR || SPACE == SP_L2) {
            const float t = q[j] - b[j];
            s0 = fmaf(t, t, s0);
        } else if constexpr (SPACE == SP_L1) {
            s0 += fabsf(q[j] - b[j]);
        } else if constexpr (SPACE == SP_LINF) {
            s0 = fmaxf(s0, fabsf(q[j] - b[j]));
        } else if constexpr (SPACE == SP_NORMCOS || SPACE == SP_NEGDOT) {
            s0 = fmaf(q[j], b[j], s0);
        } else {  // cosine / angular on raw rows: dot, |row|^2, |query|^2
            s0 = fmaf(b[j], q[j], s0);
            s1 = fmaf(b[j], b[j], s1);
            s2 = fmaf(q[j], q[j], s2);
        }
    }
}

template <int SPACE>
__device__ __forceinline__ float finish_dist(float s0, float s1, float s2) {
    if constexpr (SPACE == SP_L2) return sqrtf(s0);
    else if constexpr (SPACE == SP_NEGDOT) return -s0;
    else if constexpr (SPACE == SP_NORMCOS) {
        const float c = fmaxf(-1.0f, fminf(1.0f, s0));
        return fmaxf(0.0f, 1.0f - c);
    } else if constexpr (SPACE == SP_COSINE) return fmaxf(0.0f, 1.0f - normdot_finish(s0, s1, s2));
    else if constexpr (SPACE == SP_ANGULAR) return acosf(normdot_finish(s0, s1, s2));
    else return s0;
}

// 8-lane reductions with DPP (VALU speed; __shfl_xor would go through the LDS crossbar):
// row_half_mirror pairs lane i with 7-i inside each group of 8, then quad_perm swaps 1 and 2 apart.
template <typename T>
__device__ __forceinline__ T dpp_half_mirror(T v) {
    return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
}
template <typename T>
__device__ __forceinline__ T dpp_quad_xor1(T v) {  // quad_perm [1,0,3,2]
    return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
template <typename T>
__device__ __forceinline__ T dpp_quad_xor2(T v) {  // quad_perm [2,3,0,1]
    return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float group8_sum(float v) {
    v += dpp_half_mirror(v);
    v += dpp_quad_xor1(v);
    v += dpp_quad_xor2(v);
    return v;
}
__device__ __forceinline__ float group8_max(float v) {
    v = fmaxf(v, dpp_half_mirror(v));
    v = fmaxf(v, dpp_quad_xor1(v));
    v = fmaxf(v, dpp_quad_xor2(v));
    return v;
}
__device__ __forceinline__ int group8_sum_i(int v) {
    v += dpp_half_mirror(v);
    v += dpp_quad_xor1(v);
    v += dpp_quad_xor2(v);
    return v;
}

// Distances of the query to the m rows listed in nbr[0..m) -> nd[0..m).
// 8 lanes per row; 8 rows per pass; 4 passes issued together (32 rows in flight).
template <int SPACE>
__device__ __forceinline__ void frontier_distances(const HnswDeviceGraph& g, const float* qv,
                                                   const uint8_t* qb, int qnorm, const int* nbr,
                                                   float* nd, int m, int lane) {
    const int g8 = lane >> 3, sub = lane & 7;
    for (int base_i = 0; base_i < m; base_i += 32) {
        int ids[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            // slots past m re-read the last row (a cache hit) instead of branching: a load under its own
            // exec-masked branch is followed by a full vmcnt(0) wait, which serialises the whole gather
            const int idx = base_i + p * 8 + g8;
            ids[p] = nbr[idx < m ? idx : m - 1];
        }
        if constexpr (DistTraits<SPACE>::kU8) {
            // 128-byte rows: one 16-byte load per lane; exact integer n1 + n2 - 2*dot
            const i32x4 qq = *reinterpret_cast<const i32x4*>(qb + sub * 16);
            int dots[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                int dsum = 0;
                const i32x4 bb = *reinterpret_cast<const i32x4*>(
                    reinterpret_cast<const uint8_t*>(g.rows) + (size_t)ids[p] * 128 + sub * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) dsum = __builtin_amdgcn_udot4(qq[j], bb[j], dsum, false);
                dots[p] = dsum;
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int dot = group8_sum_i(dots[p]);
                const int idx = base_i + p * 8 + g8;
                if (sub == 0 && idx < m) nd[idx] = (float)(g.row_norm[ids[p]] + qnorm - 2 * dot);
            }
        } else {
            float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
            const float* rows = reinterpret_cast<const float*>(g.rows);
            // 128 floats of the 4 rows per pass: all 16 row loads are issued before the first one is waited for
            // (random 512-byte gathers are latency-bound: the loads in flight per wave are the throughput).
            // Out-of-range tails read a clamped address and are zeroed on both sides, so no load sits under
            // a divergent branch.
            const float* rp[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) rp[p] = rows + (size_t)ids[p] * g.ldv;
            const int dlast = g.ldv - 4;
            for (int d0 = sub * 4; d0 < g.ldv; d0 += 128) {
                f32x4 bb[4][4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int d = d0 + 32 * it;
                    const int dc = d < dlast ? d : dlast;
#pragma unroll
                    for (int p = 0; p < 4; ++p) bb[it][p] = *reinterpret_cast<const f32x4*>(rp[p] + dc);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int d = d0 + 32 * it;
                    const bool ok = d < g.ldv;
                    const int dc = d < dlast ? d : dlast;
                    f32x4 qq = *reinterpret_cast<const f32x4*>(qv + dc);
                    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                    qq = ok ? qq : zero;
#pragma unroll
                    for (int p = 0; p < 4; ++p) accum4<SPACE>(qq, ok ? bb[it][p] : zero, s0[p], s1[p], s2[p]);
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float r0, r1 = 0.f, r2 = 0.f;
                if constexpr (DistTraits<SPACE>::kMax) r0 = group8_max(s0[p]);
                else r0 = group8_sum(s0[p]);
                if constexpr (DistTraits<SPACE>::kThree) {
                    r1 = group8_sum(s1[p]);
                    r2 = group8_sum(s2[p]);
                }
                const int idx = base_i + p * 8 + g8;
                if (sub == 0 && idx < m) nd[idx] = finish_dist<SPACE>(r0, r1, r2);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Split-phase form of frontier_distances for the software-pipelined search loop: issue() requests the first 128 floats
// (u8: the whole row) of the first 32 rows and returns; finish() consumes them, fetches whatever is left (longer rows,
// rows 32..m) and writes nd[0..m).  Same arithmetic, same order as frontier_distances (bit-identical results).
template <int SPACE>
struct FrontierLoads {
    f32x4 bb[4][4];
    i32x4 bu[4];
    int ids[4];
};

template <int SPACE>
__device__ __forceinline__ void frontier_issue(FrontierLoads<SPACE>& L, const HnswDeviceGraph& g, const int* nbr, int m,
                                               int lane) {
    const int g8 = lane >> 3, sub = lane & 7;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int idx = p * 8 + g8;
        L.ids[p] = nbr[idx < m ? idx : m - 1];
    }
    if constexpr (DistTraits<SPACE>::kU8) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
            L.bu[p] = *reinterpret_cast<const i32x4*>(reinterpret_cast<const uint8_t*>(g.rows) + (size_t)L.ids[p] * 128 +
                                                       sub * 16);
    } else {
        const float* rows = reinterpret_cast<const float*>(g.rows);
        const int dlast = g.ldv - 4;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int d = sub * 4 + 32 * it;
            const int dc = d < dlast ? d : dlast;
#pragma unroll
            for (int p = 0; p < 4; ++p) L.bb[it][p] = *reinterpret_cast<const f32x4*>(rows + (size_t)L.ids[p] * g.ldv + dc);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int SPACE>
__device__ __forceinline__ void frontier_finish(FrontierLoads<SPACE>& L, const HnswDeviceGraph& g, const float* qv,
                                                const uint8_t* qb, int qnorm, const int* nbr, float* nd, int m,
                                                int lane) {
    const int g8 = lane >> 3, sub = lane & 7;
    if constexpr (DistTraits<SPACE>::kU8) {
        const i32x4 qq = *reinterpret_cast<const i32x4*>(qb + sub * 16);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int dsum = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) dsum = __builtin_amdgcn_udot4(qq[j], L.bu[p][j], dsum, false);
            const int dot = group8_sum_i(dsum);
            const int idx = p * 8 + g8;
            if (sub == 0 && idx < m) nd[idx] = (float)(g.row_norm[L.ids[p]] + qnorm - 2 * dot);
        }
    } else {
        float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        const float* rows = reinterpret_cast<const float*>(g.rows);
        const int dlast = g.ldv - 4;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        for (int d0 = sub * 4; d0 < g.ldv; d0 += 128) {
            if (d0 != sub * 4) {  // chunks after the first: fetched here (rows longer than 128 floats)
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int d = d0 + 32 * it;
                    const int dc = d < dlast ? d : dlast;
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        L.bb[it][p] = *reinterpret_cast<const f32x4*>(rows + (size_t)L.ids[p] * g.ldv + dc);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int d = d0 + 32 * it;
                const bool ok = d < g.ldv;
                const int dc = d < dlast ? d : dlast;
                f32x4 qq = *reinterpret_cast<const f32x4*>(qv + dc);
                qq = ok ? qq : zero;
#pragma unroll
                for (int p = 0; p < 4; ++p) accum4<SPACE>(qq, ok ? L.bb[it][p] : zero, s0[p], s1[p], s2[p]);
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float r0, r1 = 0.f, r2 = 0.f;
            if constexpr (DistTraits<SPACE>::kMax) r0 = group8_max(s0[p]);
            else r0 = group8_sum(s0[p]);
            if constexpr (DistTraits<SPACE>::kThree) {
                r1 = group8_sum(s1[p]);
                r2 = group8_sum(s2[p]);
            }
            const int idx = p * 8 + g8;
            if (sub == 0 && idx < m) nd[idx] = finish_dist<SPACE>(r0, r1, r2);
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (m > 32) frontier_distances<SPACE>(g, qv, qb, qnorm, nbr + 32, nd + 32, m - 32, lane);
}

}  // namespace gfxknn
