// How many independent accumulation chains / waves per SIMD does v_mfma_f32_32x32x2_f32 need to fill the pipe?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ __launch_bounds__(256, 2) void loop(float* out, int iters) {
    f32x16 a[CH];
    for (int c = 0; c < CH; ++c) a[c] = f32x16{0};
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 32 / CH; ++j)
#pragma unroll
            for (int c = 0; c < CH; ++c) a[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CH; ++c)
        for (int j = 0; j < 16; ++j) s += a[c][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH>
void run(int grid, float* out) {
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(loop<CH>, dim3(grid), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("chains %d  waves/SIMD %d : %.3f ms  %.1f TFLOP/s\n", CH, grid / 256, best,
           (double)grid * 4 * iters * 32 * 4096.0 / best * 1e-9);
}
int main() {
    float* out;
    hipMalloc(&out, 1024 * 256 * 4);
    run<1>(256, out); run<2>(256, out); run<4>(256, out);
    run<1>(512, out); run<2>(512, out); run<4>(512, out);
    return 0;
}
