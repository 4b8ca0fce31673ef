// Micro-benchmark 2 (gfx950): the f32 scan kernel's K-step shape -- 12 MFMAs (4 accumulators x 3 products), B operands in
// AGPRs, 2 ds_read_b128 two steps ahead with counted waits -- variant by variant.  One wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// V: 0 = B in VGPR, group-major; 1 = B in AGPR, group-major; 2 = B in AGPR, term-major; 3 = 1 + LDS reads;
//    4 = 3 + 8 v_max3 per 12 MFMAs spread (2 per group); 5 = 3 + 12 clumped v_max3 after the 12 MFMAs
template <int V>
__global__ __launch_bounds__(256) void probe(float* out, long long* clk, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    bf16x8 a[4], bv[8];
    for (int s = 0; s < 4; ++s) for (int i = 0; i < 8; ++i) a[s][i] = (__bf16)(threadIdx.x * 0.001f + i + s);
    for (int s = 0; s < 8; ++s) for (int i = 0; i < 8; ++i) bv[s][i] = (__bf16)(i * 0.5f + s);
    bf16x8 ba[8];
    for (int s = 0; s < 8; ++s) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(((int*)&ba[s])[0]) : "v"(((int*)&bv[s])[0]));
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i;
    __syncthreads();
    const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds + (threadIdx.x & 63) * 16;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            if (V >= 3) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(a[(kc + 2) % 4]) : "v"(la + kc * 1024) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(a[(kc + 3) % 4]) : "v"(la + kc * 1024) : "memory");
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a[kc % 4]), "+v"(a[(kc + 1) % 4]));
            }
#define M(c, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(A), "v"(B))
#define MA(c, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(A), "a"(B))
#define X2(e) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[e]) : "v"(x[(e + 1) % 8]), "v"(x[(e + 2) % 8]))
            if (V == 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) { M(g, a[kc % 4], bv[2 * g]); M(g, a[(kc + 1) % 4], bv[2 * g]); M(g, a[kc % 4], bv[2 * g + 1]); }
            } else if (V == 2) {
#pragma unroll
                for (int g = 0; g < 4; ++g) MA(g, a[kc % 4], ba[2 * g]);
#pragma unroll
                for (int g = 0; g < 4; ++g) MA(g, a[(kc + 1) % 4], ba[2 * g]);
#pragma unroll
                for (int g = 0; g < 4; ++g) MA(g, a[kc % 4], ba[2 * g + 1]);
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    MA(g, a[kc % 4], ba[2 * g]); MA(g, a[(kc + 1) % 4], ba[2 * g]); MA(g, a[kc % 4], ba[2 * g + 1]);
                    if (V == 4) { X2(2 * g); X2(2 * g + 1); }
                }
                if (V == 5) {
#pragma unroll
                    for (int e = 0; e < 12; ++e) X2(e % 8);
                }
            }
        }
    }
    long long t1 = clock64();
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15");
    float s = 0;
    for (int c = 0; c < 4; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int V>
void run(const char* name, float* out, long long* clk) {
    const int iters = 500, wgs = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<V><<<wgs, 256>>>(out, clk, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<V><<<wgs, 256>>>(out, clk, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c0; hipMemcpy(&c0, clk, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 48;
    printf("%-44s : %.1f ns/mfma  %.1f clk/mfma\n", name, ms * 1e6 / n, (double)c0 / n);
}

int main() {
    float* out; long long* clk;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&clk, 1024 * 8);
    run<0>("B in VGPR, group-major", out, clk);
    run<1>("B in AGPR, group-major", out, clk);
    run<2>("B in AGPR, term-major", out, clk);
    run<3>("B in AGPR, group-major, + 2 ds_read/12", out, clk);
    run<4>("  + 8 v_max3 spread", out, clk);
    run<5>("  + 12 v_max3 clumped", out, clk);
    return 0;
}
