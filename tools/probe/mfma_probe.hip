// Micro-benchmark (gfx950): how many VALU / LDS / s_nop instructions ride in the shadow of one
// v_mfma_f32_32x32x16_bf16 issued by the SAME wave (one wave per SIMD)?  Prints clocks per MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_probe.hip -o mfma_probe && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND, int NEXTRA, int CHAINS>
__global__ __launch_bounds__(256) void probe(float* out, long long* clk, int iters) {
    __shared__ float lds[4096];
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    uint32_t la = (uint32_t)(threadIdx.x * 16) & 4095;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b));
#pragma unroll
            for (int e = 0; e < NEXTRA; ++e) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x[e % 8]));
                if (KIND == 1) asm volatile("s_nop 0");
                if (KIND == 2) { float4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(la)); }
                if (KIND == 3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[e % 8]) : "v"(x[(e + 1) % 8]), "v"(x[(e + 2) % 8]));
            }
        }
        if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    long long t1 = clock64();
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15");
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int KIND, int NEXTRA, int CHAINS>
void run(const char* name, float* out, long long* clk, int wgs) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KIND, NEXTRA, CHAINS><<<wgs, 256>>>(out, clk, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<KIND, NEXTRA, CHAINS><<<wgs, 256>>>(out, clk, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c0; hipMemcpy(&c0, clk, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * CHAINS;
    printf("%-10s extra=%d chains=%d wgs=%d : %.1f ns/mfma  (s_memtime-clock %.1f ticks/mfma)\n", name, NEXTRA, CHAINS, wgs,
           ms * 1e6 / n, (double)c0 / n);
}

int main() {
    float* out; long long* clk;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&clk, 1024 * 8);
    for (int wgs : {256, 512}) {
        run<0, 0, 4>("none", out, clk, wgs);
        run<0, 0, 1>("none", out, clk, wgs);
        run<0, 2, 4>("v_add", out, clk, wgs);
        run<0, 4, 4>("v_add", out, clk, wgs);
        run<0, 6, 4>("v_add", out, clk, wgs);
        run<0, 8, 4>("v_add", out, clk, wgs);
        run<0, 12, 4>("v_add", out, clk, wgs);
        run<3, 4, 4>("v_max3", out, clk, wgs);
        run<3, 8, 4>("v_max3", out, clk, wgs);
        run<1, 4, 4>("s_nop", out, clk, wgs);
        run<1, 8, 4>("s_nop", out, clk, wgs);
        run<2, 1, 4>("ds_read", out, clk, wgs);
        run<2, 2, 4>("ds_read", out, clk, wgs);
    }
    return 0;
}
