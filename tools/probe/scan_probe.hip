// Micro-benchmark (gfx950): the steady-state K-step of the one-product scan -- two waves per SIMD (512 threads), per K-step
// one ds_read_b128 three steps ahead, a counted wait, two v_mfma_f32_32x32x16_f16 on two accumulator chains -- with its
// ingredients switched on one by one.  Prints MFMA-pipe utilisation (32 clocks per MFMA and SIMD = 100 %).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/scan_probe.hip -o /tmp/scan_probe && /tmp/scan_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// FLAGS: 1 = LDS fragment reads + counted waits, 2 = B operand in AGPRs, 4 = check VALU (8 v_max3 + cmp per 4 K-steps),
//        8 = s_barrier every 64 K-steps, 16 = fresh C operand every 8 K-steps (start values from other registers),
//        64 = two accumulator sets alternating between blocks of 8 K-steps, the check reads the OTHER set (the scores the
//             previous block's MFMAs just wrote) instead of idle registers; 128 = the block's start values read from LDS
//             (4 more ds_read_b128 per block into the C tuple, waits counted accordingly)
//        32 = the row stream: per 16 K-steps (one 64-row stage) every wave requests 2 KB by LDS-DMA from a 2 MB slice of a
//             512 MB buffer (scalar-base form), waited for with vmcnt(0) at the barrier
template <int FLAGS, int NW>
__global__ __launch_bounds__(NW * 64) void probe(float* out, long long* clk, int ksteps, const char* stream) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63;
    f16x8 qv[2], fr[4];
    for (int i = 0; i < 8; ++i) { qv[0][i] = (_Float16)(lane * 0.01f + i); qv[1][i] = (_Float16)(i * 0.5f); }
    for (int s = 0; s < 4; ++s) fr[s] = qv[0];
    f16x8 qa0 = qv[0], qa1 = qv[1];
    if (FLAGS & 2) { asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %0\n v_accvgpr_write_b32 a2, %0\n v_accvgpr_write_b32 a3, %0\n"
                                  "v_accvgpr_write_b32 a4, %0\n v_accvgpr_write_b32 a5, %0\n v_accvgpr_write_b32 a6, %0\n v_accvgpr_write_b32 a7, %0" :: "v"(lane) : "a0","a1","a2","a3","a4","a5","a6","a7"); }
    f32x16 accs[2][2], iv;
    for (int i = 0; i < 16; ++i) { accs[0][0][i] = 0.f; accs[0][1][i] = 0.f; accs[1][0][i] = 0.f; accs[1][1][i] = 0.f; iv[i] = (float)i; }
    float m = 0.f;
    f32x4 ivq[4];
    for (int i = 0; i < 4; ++i) ivq[i] = f32x4{1.f, 2.f, 3.f, 4.f};
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = (float)i;
    __syncthreads();
    // (the scan's conflict-free image: 32 rows x 256 B, 16-byte chunks swizzled by the row)
    const int l31 = lane & 31, hh = lane >> 5;
    uint32_t lad[8];
    for (int kc = 0; kc < 8; ++kc)
        lad[kc] = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + (uint32_t)(l31 * 256 + (((kc * 2 + hh) ^ (l31 & 15)) * 16));
    if (FLAGS & 1) for (int s = 0; s < 3; ++s) asm volatile("ds_read_b128 %0, %1" : "=v"(fr[s]) : "v"(lad[s]) : "memory");
    const int wave_u = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const uint32_t dma_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + 16384u + (uint32_t)wave_u * 2048u;
    const uint32_t dma_voff = (uint32_t)lane * 16u;
    unsigned long long sbase = (unsigned long long)(uintptr_t)stream + (unsigned long long)blockIdx.x * (2u << 20) + (unsigned long long)wave_u * 2048ull;
    int stage = 0;
    long long t0 = clock64();
    for (int k00 = 0; k00 < ksteps; k00 += 16) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const int k0 = k00 + 8 * blk;
        f32x16& acc0 = accs[(FLAGS & 64) ? blk : 0][0];
        f32x16& acc1 = accs[(FLAGS & 64) ? blk : 0][1];
        const f32x16& chk = (FLAGS & 64) ? accs[blk ^ 1][(0)] : iv;
        if ((FLAGS & 32) && (k0 & 15) == 0) {
            const unsigned long long sb = sbase + (unsigned long long)((stage & 127) * 16384);
            const uint32_t m0v = dma_lds + (uint32_t)((stage % 3) * 16384);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(dma_voff), "s"(sb), "s"(m0v) : "memory");
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(dma_voff), "s"(sb + 1024), "s"(m0v + 1024u) : "memory");
            ++stage;
        }
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            if (FLAGS & 1) {
                if ((FLAGS & 128) && kc == 6)
                    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\tds_read_b128 %2, %4 offset:64\n\tds_read_b128 %3, %4 offset:96"
                                 : "=v"(ivq[0]), "=v"(ivq[1]), "=v"(ivq[2]), "=v"(ivq[3]) : "v"(lad[0]) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(fr[(kc + 3) % 4]) : "v"(lad[(kc + 3) & 7]) : "memory");
                if ((FLAGS & 128) && kc >= 6) asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(fr[kc % 4]));
                else if ((FLAGS & 128) && kc == 0) {
                    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fr[kc % 4]), "+v"(ivq[0]), "+v"(ivq[1]), "+v"(ivq[2]), "+v"(ivq[3]));
                    for (int i = 0; i < 16; ++i) iv[i] = ivq[i >> 2][i & 3];
                } else asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fr[kc % 4]));
            }
            const bool fresh = (FLAGS & 16) && kc == 0;
            if (FLAGS & 2) {
                if (fresh) {
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[0:3], %2" : "=&v"(acc0) : "v"(fr[kc % 4]), "v"(iv));
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[4:7], %2" : "=&v"(acc1) : "v"(fr[kc % 4]), "v"(iv));
                } else {
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[0:3], %0" : "+v"(acc0) : "v"(fr[kc % 4]));
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[4:7], %0" : "+v"(acc1) : "v"(fr[kc % 4]));
                }
            } else {
                if (fresh) {
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(acc0) : "v"(fr[kc % 4]), "v"(qa0), "v"(iv));
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(acc1) : "v"(fr[kc % 4]), "v"(qa1), "v"(iv));
                } else {
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fr[kc % 4]), "v"(qa0));
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(fr[kc % 4]), "v"(qa1));
                }
            }
            if ((FLAGS & 4) && (kc & 3) == 2) {   // the check of "the previous block": 8 VALU + a compare on other registers
                float a0, a1, a2, a3, a4, r0, r1;
                asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(a0) : "v"(chk[0]), "v"(chk[1]), "v"(chk[2]));
                asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(a1) : "v"(chk[3]), "v"(chk[4]), "v"(chk[5]));
                asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(a2) : "v"(chk[6]), "v"(chk[7]), "v"(chk[8]));
                asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(a3) : "v"(chk[9]), "v"(chk[10]), "v"(chk[11]));
                asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(a4) : "v"(chk[12]), "v"(chk[13]), "v"(chk[14]));
                asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(a0), "v"(a1), "v"(a2));
                asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r1) : "v"(a3), "v"(a4), "v"(chk[15]));
                asm volatile("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(r0), "v"(r1));
                unsigned long long trig = __builtin_amdgcn_ballot_w64(m >= 1e30f);
                asm volatile("" : "+s"(trig));
                if (trig) out[0] = m;
            }
        }
        if ((FLAGS & 8) && (k0 & 63) == 56) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    long long t1 = clock64();
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15");
    float s = m;
    for (int i = 0; i < 16; ++i) s += accs[0][0][i] + accs[0][1][i] + accs[1][0][i] + accs[1][1][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

static char* g_stream = nullptr;
template <int FLAGS, int NW>
void run(const char* name, float* out, long long* clk) {
    const int ksteps = 40000, wgs = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<FLAGS, NW><<<wgs, NW * 64>>>(out, clk, 64, g_stream);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<FLAGS, NW><<<wgs, NW * 64>>>(out, clk, ksteps, g_stream);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c0; hipMemcpy(&c0, clk, 8, hipMemcpyDeviceToHost);
    // per SIMD: NW / 4 waves x 2 MFMAs per K-step
    const double mfma_per_simd = (double)ksteps * 2.0 * (NW / 4);
    const double ns_per = ms * 1e6 / mfma_per_simd;
    printf("%-46s waves/SIMD %d : %6.2f ns per MFMA and SIMD  = %5.1f %% of the 2.4 GHz pipe (13.33 ns), %6.1f TFLOP/s chip; wall-clock ticks/MFMA %.2f\n",
           name, NW / 4, ns_per, 100.0 * 13.333 / ns_per, 2.0 * 32 * 32 * 16 * 1024.0 / ns_per / 1e3, (double)c0 / mfma_per_simd);
}

int main() {
    float* out; long long* clk;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 8);
    hipMalloc(&g_stream, 512u << 20); hipMemset(g_stream, 1, 512u << 20);
    run<0, 8>("MFMA only (B in VGPR)", out, clk);
    run<2, 8>("MFMA only (B in AGPR)", out, clk);
    run<2, 4>("MFMA only (B in AGPR)", out, clk);
    run<3, 8>("+ LDS reads 3 ahead, counted waits", out, clk);
    run<3, 4>("+ LDS reads 3 ahead, counted waits", out, clk);
    run<7, 8>("+ check VALU", out, clk);
    run<15, 8>("+ barrier every 64 K-steps", out, clk);
    run<31, 8>("+ fresh C operand every 8 K-steps", out, clk);
    run<31, 4>("+ fresh C operand every 8 K-steps", out, clk);
    run<63, 8>("+ the row stream by LDS-DMA (16 KB per stage)", out, clk);
    run<63 - 4, 8>("  ... without the check VALU", out, clk);
    run<35, 8>("MFMA + LDS reads + row stream only", out, clk);
    run<63 + 64, 8>("all + the check reads the previous block's accumulators", out, clk);
    run<63 + 64 + 128, 8>("all + start values read from LDS per block", out, clk);
    run<63 + 64 + 128, 4>("all + start values read from LDS per block", out, clk);
    return 0;
}
