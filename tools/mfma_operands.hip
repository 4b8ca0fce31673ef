// Which part of bf_select's inner loop costs MFMA issue slots?  One wave per SIMD (256 workgroups of 256
// threads), 64-MFMA blocks like block_mfma:
//   v0: A and B constant registers            v1: B from 64 distinct registers (the query fragments)
//   v2: v1 + A from LDS (ds_read_b128 two groups ahead, padded rows)   v3: v2 with both LDS and 2 waves/SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(256, 2) void loop(float* out, const float* in, int iters) {
    __shared__ __attribute__((aligned(16))) float tile[64 * 132];
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    for (int i = threadIdx.x; i < 64 * 132; i += 256) tile[i] = in[i & 1023];
    __syncthreads();
    f32x4 bq[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) bq[t] = *reinterpret_cast<const f32x4*>(in + 64 * t + 4 * h + (l31 & 7) * 8);
    f32x16 n = {0};
    const float* ap = tile + l31 * 132 + 4 * h;
    for (int it = 0; it < iters; ++it) {
        if (V == 0) {
#pragma unroll
            for (int tt = 0; tt < 16; ++tt)
#pragma unroll
                for (int j = 0; j < 4; ++j) n = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[0][0], bq[0][1], n, 0, 0, 0);
        } else if (V == 1) {
#pragma unroll
            for (int tt = 0; tt < 16; ++tt)
#pragma unroll
                for (int j = 0; j < 4; ++j) n = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[0][0], bq[tt][j], n, 0, 0, 0);
        } else {
            const float* a2 = ap + ((it & 1) ? 32 * 132 : 0);
            f32x4 c0 = *reinterpret_cast<const f32x4*>(a2), c1 = *reinterpret_cast<const f32x4*>(a2 + 8), c2 = c1;
#pragma unroll
            for (int tt = 0; tt < 16; ++tt) {
                if (tt + 2 < 16) c2 = *reinterpret_cast<const f32x4*>(a2 + 8 * (tt + 2));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) n = __builtin_amdgcn_mfma_f32_32x32x2f32(c0[j], bq[tt][j], n, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                c0 = c1;
                c1 = c2;
            }
        }
    }
    float s = 0;
    for (int j = 0; j < 16; ++j) s += n[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int V>
void run(int grid, float* out, const float* in, const char* what) {
    const int iters = 2048;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(loop<V>, dim3(grid), dim3(256), 0, 0, out, in, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("%-52s waves/SIMD %d : %.3f ms  %.1f TFLOP/s\n", what, grid / 256, best,
           (double)grid * 4 * iters * 64 * 4096.0 / best * 1e-9);
}
int main() {
    float *out, *in;
    hipMalloc(&out, 1024 * 256 * 4);
    hipMalloc(&in, 4096 * 4);
    hipMemset(in, 0, 4096 * 4);
    run<0>(256, out, in, "A, B constant registers");
    run<1>(256, out, in, "B from 64 registers");
    run<2>(256, out, in, "B from 64 registers, A from LDS (2 groups ahead)");
    run<2>(512, out, in, "B from 64 registers, A from LDS (2 groups ahead)");
    return 0;
}
