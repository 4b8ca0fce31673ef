// How much does an empty kernel cost in a stream of dependent launches -- ordinary vs cooperative launch?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/coop_launch_probe.hip -o /tmp/coop_probe && /tmp/coop_probe
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;
__global__ void k_plain(const int* flag, int* out) {
    if (*flag == 0) return;
    out[blockIdx.x] = 1;
}
__global__ void k_coop(const int* flag, int* out) {
    if (*flag == 0) return;
    out[blockIdx.x] = 1;
    cg::this_grid().sync();
    out[blockIdx.x] += out[(blockIdx.x + 1) % gridDim.x];
}
__global__ void k_work(float* x, int n) {   // a little real work between the empties
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = x[i] * 1.0001f + 1.f;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    int *flag, *out; float* x;
    CK(hipMalloc(&flag, 4)); CK(hipMalloc(&out, 4096 * 4)); CK(hipMalloc(&x, 1 << 22));
    CK(hipMemset(flag, 0, 4)); CK(hipMemset(out, 0, 4096 * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 2000;
    for (int mode = 0; mode < 4; ++mode) {
        for (int warm = 0; warm < 2; ++warm) {
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r) {
                hipLaunchKernelGGL(k_work, dim3(1024), dim3(256), 0, s, x, 1 << 20);
                if (mode == 1) for (int j = 0; j < 4; ++j) hipLaunchKernelGGL(k_plain, dim3(1024), dim3(256), 0, s, flag, out);
                if (mode == 2) hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, s, flag, out);
                if (mode == 3) {
                    void* args[] = {&flag, &out};
                    CK(hipLaunchCooperativeKernel((void*)k_coop, dim3(256), dim3(256), args, 0, s));
                }
            }
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (warm) printf("mode %d (%s): %.2f us per iteration\n", mode,
                             mode == 0 ? "work only" : mode == 1 ? "work + 4 plain empties of 1024 blocks" : mode == 2 ? "work + 1 plain empty of 256 blocks" : "work + 1 cooperative empty of 256 blocks",
                             ms * 1000.f / reps);
        }
    }
    return 0;
}
