// Compiled and run by tests/test_cabi_cpu.py: the C++ mirror of lib.zig links against
// libnmslib_c.so and walks the GPU-free part of the reference's "Index dense vector workflow"
// and metadata tests (lib.zig:1273-1312,1518-1558).
#include <cstdio>

#include "../nmslib_zig_amd/host/nmslib.hpp"

int main() {
    using namespace nmslib;
    Index idx("cosine", "hnsw");
    const float rows[3][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}};
    const int32_t ids[3] = {10, 20, 30};
    idx.addDenseBatch(&rows[0][0], 3, 4, ids);
    if (idx.dataQty() != 3) return 1;
    if (idx.getSpaceType() != "cosine" || idx.getMethod() != "hnsw") return 2;
    idx.setThreadPoolSize(4);
    if (idx.getThreadPoolSize() != 4) return 3;
    try {
        idx.setThreadPoolSize(0);
        return 4;
    } catch (const Error& e) {
        if (e.code != NMSLIB_ERROR_INVALID_ARGUMENT) return 5;
    }
    {
        TrackingAllocator a;
        Params p(a);
        p.add("M", 8).add("mult", 0.5).add("algoType", std::string("v1merge"));
    }
    std::puts("host mirror ok");
    return 0;
}
