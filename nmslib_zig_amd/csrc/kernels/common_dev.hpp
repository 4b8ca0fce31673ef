// Shared device helpers for the gfx950 k-NN kernels (wave64 everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace gfxknn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

// Order-preserving map float -> uint32 (larger float => larger uint).
__device__ __forceinline__ uint32_t f32_ord(float f) {
    uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float ord_f32(uint32_t o) {
    uint32_t b = o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    return __uint_as_float(b);
}
__device__ __forceinline__ uint32_t i32_ord(int v) { return (uint32_t)v ^ 0x80000000u; }
__device__ __forceinline__ int ord_i32(uint32_t o) { return (int)(o ^ 0x80000000u); }

// "bigger is better" selection key: (score, then LOWER position wins).
__device__ __forceinline__ u64 make_sel_key(uint32_t score_ord, uint32_t pos) {
    return ((u64)score_ord << 32) | (u64)(0xFFFFFFFFu - pos);
}
__device__ __forceinline__ uint32_t sel_key_pos(u64 k) { return 0xFFFFFFFFu - (uint32_t)k; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// Bitonic sort of P (power of two) 64-bit keys in LDS by ONE wave (LDS operations of a
// wave execute in order, so no barrier is needed between the stages; the wave_barrier
// only pins the compiler's order).  descending != 0 -> largest first.
__device__ __forceinline__ void wave_bitonic_u64(u64* s, int P, int lane, bool descending) {
    for (int k2 = 2; k2 <= P; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (P >> 1); t += 64) {
                int i = 2 * t - (t & (j - 1));
                int l = i + j;
                u64 a = s[i], b = s[l];
                bool up = ((i & k2) == 0) != descending;  // ascending run?
                bool sw = up ? (a > b) : (a < b);
                if (sw) {
                    s[i] = b;
                    s[l] = a;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// Same network run by a whole workgroup (blockDim.x threads), ascending.
__device__ __forceinline__ void block_bitonic_u64_asc(u64* s, int P, int tid, int nthreads) {
    for (int k2 = 2; k2 <= P; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += nthreads) {
                int i = 2 * t - (t & (j - 1));
                int l = i + j;
                u64 a = s[i], b = s[l];
                bool up = ((i & k2) == 0);
                bool sw = up ? (a > b) : (a < b);
                if (sw) {
                    s[i] = b;
                    s[l] = a;
                }
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------
// Exact ("direct formula") distance of one stored row to one query, computed by one wave.
// These are the values the reference returns to the caller:
//   SP_L2      sqrt(sum (a-b)^2)              distcomp_lp.cc:304-371
//   SP_L1      sum |a-b|                      distcomp_lp.cc:190-251
//   SP_LINF    max |a-b|                      distcomp_lp.cc:77-139
//   SP_COSINE  max(0, 1 - normdot)            distcomp_scalar.cc:83-168,267-271
//   SP_ANGULAR acos(normdot)                  distcomp_scalar.cc:254-258
//   SP_NEGDOT  -dot                           distcomp_scalar.cc:193-245, space_scalar.cc:59-68
//   SP_L2SQR   sum (a-b)^2                    hnsw_distfunc_opt_impl_inline.h:42-122
//   SP_NORMCOS max(0, 1 - clamp(dot))         hnsw.cc:78-81 (rows pre-normalised)
// Accumulation is f32 with a fixed lane-strided order (deterministic); it differs from the
// reference's SSE/AVX lane order only in rounding (bar: 1e-5 relative).
// Every lane returns the same value.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float normdot_finish(float dot, float n1, float n2) {
    const float eps = 1.17549435e-38f * 2.0f;  // numeric_limits<float>::min() * 2
    if (n1 < eps || n2 < eps) return 0.0f;
    float v = dot / sqrtf(n1) / sqrtf(n2);
    return fmaxf(-1.0f, fminf(1.0f, v));
}

__device__ __forceinline__ float wave_exact_distance_f32(int space, const float* __restrict__ a,
                                                         const float* __restrict__ q, int dim,
                                                         int lane) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if (space == SP_L2 || space == SP_L2SQR) {
        for (int d = lane; d < dim; d += 64) {
            float t = a[d] - q[d];
            s0 = fmaf(t, t, s0);
        }
        s0 = wave_sum(s0);
        return space == SP_L2 ? sqrtf(s0) : s0;
    } else if (space == SP_L1) {
        for (int d = lane; d < dim; d += 64) s0 += fabsf(a[d] - q[d]);
        return wave_sum(s0);
    } else if (space == SP_LINF) {
        for (int d = lane; d < dim; d += 64) s0 = fmaxf(s0, fabsf(a[d] - q[d]));
        return wave_max(s0);
    } else if (space == SP_NEGDOT || space == SP_NORMCOS) {
        for (int d = lane; d < dim; d += 64) s0 = fmaf(a[d], q[d], s0);
        s0 = wave_sum(s0);
        if (space == SP_NEGDOT) return -s0;
        float c = fmaxf(-1.0f, fminf(1.0f, s0));
        return fmaxf(0.0f, 1.0f - c);
    } else {  // SP_COSINE, SP_ANGULAR
        for (int d = lane; d < dim; d += 64) {
            float x = a[d], y = q[d];
            s0 = fmaf(x, y, s0);
            s1 = fmaf(x, x, s1);
            s2 = fmaf(y, y, s2);
        }
        s0 = wave_sum(s0);
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        float sim = normdot_finish(s0, s1, s2);
        if (space == SP_ANGULAR) return acosf(sim);
        return fmaxf(0.0f, 1.0f - sim);
    }
}

// Four rows at once, dim <= 256: the same per-lane accumulation order and the same reductions as
// wave_exact_distance_f32 (bit-identical results), but all row loads are issued before the first is used --
// a re-rank over thousands of survivors is otherwise one exposed HBM latency per row.  Lanes past `dim`
// read element 0 and contribute zeros on both sides.
__device__ __forceinline__ void wave_exact_distance_f32_x4(int space, const float* const a[4],
                                                           const float* __restrict__ q, int dim, int lane,
                                                           float out[4]) {
    float va[4][4], vq[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int d = lane + 64 * c;
        const bool ok = d < dim;
        const int di = ok ? d : 0;
        const float qv = q[di];
        vq[c] = ok ? qv : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = a[r][di];
            va[r][c] = ok ? x : 0.f;
        }
    }
    const int nch = (dim + 63) >> 6;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        if (space == SP_L2 || space == SP_L2SQR) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < nch) {
                    const float t = va[r][c] - vq[c];
                    s0 = fmaf(t, t, s0);
                }
            s0 = wave_sum(s0);
            out[r] = space == SP_L2 ? sqrtf(s0) : s0;
        } else if (space == SP_L1) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < nch) s0 += fabsf(va[r][c] - vq[c]);
            out[r] = wave_sum(s0);
        } else if (space == SP_LINF) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < nch) s0 = fmaxf(s0, fabsf(va[r][c] - vq[c]));
            out[r] = wave_max(s0);
        } else if (space == SP_NEGDOT || space == SP_NORMCOS) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < nch) s0 = fmaf(va[r][c], vq[c], s0);
            s0 = wave_sum(s0);
            if (space == SP_NEGDOT) {
                out[r] = -s0;
            } else {
                const float cc = fmaxf(-1.0f, fminf(1.0f, s0));
                out[r] = fmaxf(0.0f, 1.0f - cc);
            }
        } else {  // SP_COSINE, SP_ANGULAR
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < nch) {
                    const float x = va[r][c], y = vq[c];
                    s0 = fmaf(x, y, s0);
                    s1 = fmaf(x, x, s1);
                    s2 = fmaf(y, y, s2);
                }
            s0 = wave_sum(s0);
            s1 = wave_sum(s1);
            s2 = wave_sum(s2);
            const float sim = normdot_finish(s0, s1, s2);
            out[r] = space == SP_ANGULAR ? acosf(sim) : fmaxf(0.0f, 1.0f - sim);
        }
    }
}

// uint8 SIFT: exact integer squared L2 (distcomp_l2sqr_sift.cc:41-50 gives the same
// integer as sum (a-b)^2).  128 bytes per row, 2 per lane.
__device__ __forceinline__ int wave_exact_distance_u8(const uint8_t* __restrict__ a,
                                                      const uint8_t* __restrict__ q, int lane) {
    int d0 = (int)a[2 * lane] - (int)q[2 * lane];
    int d1 = (int)a[2 * lane + 1] - (int)q[2 * lane + 1];
    return wave_sum_i(d0 * d0 + d1 * d1);
}

}  // namespace gfxknn
