// Practical ceiling of the matrix pipe on this part: a register-only loop of dependent-free MFMAs
// (no LDS, no memory), same occupancy as bf_select (2 workgroups of 256 threads per CU).
// Prints achieved TFLOP/s / TOP/s and the shader clock seen by the kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void f32_loop(float* out, int iters, long long* clk) {
    f32x16 a0 = {0}, a1 = {0};
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    const long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        }
    }
    const long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0;
    for (int j = 0; j < 16; ++j) s += a0[j] + a1[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = c1 - c0;
        clk[1] = r1 - r0;
    }
}
__global__ __launch_bounds__(256, 2) void i8_loop(int* out, int iters) {
    i32x16 a0 = {0}, a1 = {0};
    i32x4 x = {(int)threadIdx.x, 1, 2, 3}, y = {(int)blockIdx.x, 5, 6, 7};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(y, x, a1, 0, 0, 0);
        }
    }
    int s = 0;
    for (int j = 0; j < 16; ++j) s += a0[j] + a1[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    const int grid = 512, iters = 4096;
    float* out;
    long long* clk;
    hipMalloc(&out, grid * 256 * 4);
    hipMalloc(&clk, 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(f32_loop, dim3(grid), dim3(256), 0, 0, out, iters, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        long long h[2];
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double flops = (double)grid * 4 * iters * 32 * 4096.0;
        printf("f32 32x32x2 : %.3f ms  %.1f TFLOP/s   shader clock %.0f MHz (cycles %lld / 100MHz ticks %lld)\n", ms,
               flops / ms * 1e-9, (double)h[0] / ((double)h[1] / 100.0), h[0], h[1]);
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(i8_loop, dim3(grid), dim3(256), 0, 0, (int*)out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double ops = (double)grid * 4 * iters * 32 * (32.0 * 32 * 32 * 2);
        printf("i8 32x32x32 : %.3f ms  %.1f TOP/s\n", ms, ops / ms * 1e-9);
    }
    return 0;
}
