// HNSW construction on the GPU: batched insertion.
//
// Restates Hnsw::add (src/method/hnsw.cc:534-609) for a whole batch of new nodes at once.  Per
// level, top down, three launches:
//   1. search    - kSearchElementsWithAttemptsLevel (hnsw.cc:611-708) for every new node of the
//                  batch = the search kernel in construction mode (hnsw_kernels.hip): the query is
//                  a stored row, ef = efConstruction, the whole result set is returned;
//   2. select    - HnswNode::getNeighborsByHeuristic2 (include/method/hnsw.h:129-169) on each result
//                  set -> the new node's forward list, plus one queued reverse-link request per
//                  selected neighbour;
//   3. link      - HnswNode::addFriendlevel (hnsw.h:258-314) target by target: append, or shrink
//                  with the same heuristic when the list is full.
// The nodes of one batch do not see each other while searching (like the reference's concurrent
// inserts, which lock one node at a time).  Reverse-link requests are written to fixed slots (new node x M),
// sorted by (target, new node) with one device radix sort, and applied per target in that order: any
// number of requests per target, no atomics, no dependence on scheduling -> the construction is
// deterministic.  One wavefront per node/target; no workgroup ever waits for another inside a launch.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "hnsw_common_dev.hpp"
#include "kernels.hpp"

namespace gfxknn {

struct BuildArgs {
    HnswDeviceGraph g;
    int32_t* links0;
    int32_t* up_links;
    int M, delaunay, level;
    const int32_t* pts;
    int npts;
    const int32_t* cand_ids;
    const float* cand_d;
    const int32_t* cand_n;
    int stride;
    u64* req_key;        // select: [npts][M] slots, key = target << 32 | new node (pad = ~0); link: the sorted keys
    float* req_dist;     // distance(target, new node), same order as req_key
    int req_total;       // link: number of slots (valid entries sort to the front)
    int32_t* active;     // link: index of the first request of each target
    int32_t* nactive;
    // batch-mates (hnsw_build_mates_kernel): per new node up to XCAP earlier nodes of the same batch that are closer
    // than its worst candidate, ascending by distance
    int32_t* extra_ids;
    float* extra_d;
    int32_t* extra_n;
};

constexpr int XCAP = 64;

__device__ __forceinline__ int32_t* adj_list(const BuildArgs& a, int node) {
    if (a.level == 0) return a.links0 + (size_t)node * (a.g.maxM0 + 1);
    return a.up_links + a.g.up_off[node] + (int64_t)(a.level - 1) * (a.g.maxM + 1);
}

// Stored row `node` becomes the "query" of frontier_distances (LDS copy; u8: bytes + norm).
template <int SPACE>
__device__ __forceinline__ void stage_row(const HnswDeviceGraph& g, int node, float* qv, int& qnorm, int lane) {
    if constexpr (DistTraits<SPACE>::kU8) {
        const uint8_t* src = reinterpret_cast<const uint8_t*>(g.rows) + (size_t)node * 128;
        reinterpret_cast<uint16_t*>(qv)[lane] = reinterpret_cast<const uint16_t*>(src)[lane];
        qnorm = g.row_norm[node];
    } else {
        const float* src = reinterpret_cast<const float*>(g.rows) + (size_t)node * g.ldv;
        for (int d = lane; d < g.ldv; d += 64) qv[d] = src[d];
        qnorm = 0;
    }
    __builtin_amdgcn_wave_barrier();
}

// getNeighborsByHeuristic2 over candidates sorted by ascending distance to the centre node.
// A candidate is kept unless some already kept node is strictly closer to it than the centre is.
// Fewer than NN candidates: all are kept (hnsw.h:133-135).  delaunay_type 0: the NN closest.
template <int SPACE>
__device__ __forceinline__ int heuristic2(const HnswDeviceGraph& g, const int* cid, const float* cd, int nc, int NN,
                                          int delaunay, float* qv, float* nd, int* kept_id, float* kept_d, int lane) {
    if (nc < NN || delaunay == 0) {
        const int n = nc < NN ? nc : NN;
        for (int i = lane; i < n; i += 64) {
            kept_id[i] = cid[i];
            kept_d[i] = cd[i];
        }
        __builtin_amdgcn_wave_barrier();
        return n;
    }
    int nk = 0;
    for (int i = 0; i < nc && nk < NN; ++i) {
        const int c = cid[i];
        const float dc = cd[i];
        bool good = true;
        if (nk > 0) {
            int qnorm;
            stage_row<SPACE>(g, c, qv, qnorm, lane);
            frontier_distances<SPACE>(g, qv, reinterpret_cast<const uint8_t*>(qv), qnorm, kept_id, nd, nk, lane);
            bool bad = false;
            for (int j = lane; j < nk; j += 64) bad |= nd[j] < dc;
            good = !__any(bad);
        }
        if (good) {
            if (lane == 0) {
                kept_id[nk] = c;
                kept_d[nk] = dc;
            }
            nk++;
        }
        __builtin_amdgcn_wave_barrier();
    }
    return nk;
}

template <int SPACE>
__global__ __launch_bounds__(64) void hnsw_build_select_kernel(BuildArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const HnswDeviceGraph& g = a.g;
    const int q = blockIdx.x, lane = threadIdx.x;
    const int qfloats = DistTraits<SPACE>::kU8 ? 32 : g.ldv;
    float* qv = reinterpret_cast<float*>(smem);                 // [ldv]
    float* nd = qv + qfloats;                                   // [64]
    int* kept_id = reinterpret_cast<int*>(nd + 64);             // [64]
    float* kept_d = reinterpret_cast<float*>(kept_id + 64);     // [64]
    int* lc_id = reinterpret_cast<int*>(kept_d + 64);           // [stride]
    float* lc_d = reinterpret_cast<float*>(lc_id + a.stride);   // [stride]

    const int p = a.pts[q];
    int nc = a.cand_n[q];
    const int nx = a.extra_n ? a.extra_n[q] : 0;
    if (nx == 0) {
        for (int i = lane; i < nc; i += 64) {
            lc_id[i] = a.cand_ids[(size_t)q * a.stride + i];
            lc_d[i] = a.cand_d[(size_t)q * a.stride + i];
        }
    } else {
        // merge the graph-search results with the close batch-mates (both ascending; equal distances: graph
        // results first), keep the `stride` (= efConstruction) closest: what kSearchElementsWithAttemptsLevel would
        // have returned had the earlier nodes of this batch already been linked (hnsw.cc:611-708)
        const float xd = lane < nx ? a.extra_d[(size_t)q * XCAP + lane] : INFINITY;
        const int xi = lane < nx ? a.extra_ids[(size_t)q * XCAP + lane] : -1;
        int below = 0;  // graph candidates <= xd
        for (int base = 0; base < nc; base += 64) {
            const int i = base + lane;
            const float cd = i < nc ? a.cand_d[(size_t)q * a.stride + i] : INFINITY;
            for (int t = 0; t < nx; ++t) {
                const float xt = __shfl(xd, t, 64);
                const int c = __popcll(__ballot(i < nc && cd <= xt));
                if (lane == t) below += c;
            }
        }
        for (int base = 0; base < nc; base += 64) {
            const int i = base + lane;
            const float cd = i < nc ? a.cand_d[(size_t)q * a.stride + i] : INFINITY;
            const int cid = i < nc ? a.cand_ids[(size_t)q * a.stride + i] : -1;
            int sh = 0;  // batch-mates strictly closer than this candidate (uniform loop: shuffles need every lane)
            for (int t = 0; t < nx; ++t) sh += (__shfl(xd, t, 64) < cd) ? 1 : 0;
            if (i < nc && i + sh < a.stride) {
                lc_id[i + sh] = cid;
                lc_d[i + sh] = cd;
            }
        }
        if (lane < nx && below + lane < a.stride) {
            lc_id[below + lane] = xi;
            lc_d[below + lane] = xd;
        }
        nc = nc + nx < a.stride ? nc + nx : a.stride;
    }
    __builtin_amdgcn_wave_barrier();
    const int nk = heuristic2<SPACE>(g, lc_id, lc_d, nc, a.M, a.delaunay, qv, nd, kept_id, kept_d, lane);

    // forward list of the new node: link() is called farthest first (hnsw.cc:597-601)
    int32_t* L = adj_list(a, p);
    if (lane == 0) L[0] = nk;
    if (lane < nk) L[1 + lane] = kept_id[nk - 1 - lane];
    // one reverse-link request per selected neighbour, in this node's own slots
    if (lane < nk) {
        a.req_key[(size_t)q * a.M + lane] = ((u64)(uint32_t)kept_id[lane] << 32) | (uint32_t)p;
        a.req_dist[(size_t)q * a.M + lane] = kept_d[lane];
    }
}

// Batch-mates: the nodes of one batch are searched against the graph as it was BEFORE the batch, so a node never meets
// the earlier nodes of its own batch -- harmless for shuffled data, fatal for data whose insertion order has locality
// (near-duplicates arriving together end up linked only to their common old neighbours).  This kernel gives node j what
// sequential insertion gives it: every EARLIER node of the batch (same level slice) that is closer than its worst
// graph-search candidate, by exhaustive distance evaluation (one wave per node, 32 rows per gather; the batch's rows
// are L2 / Infinity-Cache resident).  The select kernel merges them into the candidate list.
template <int SPACE>
__global__ __launch_bounds__(64) void hnsw_build_mates_kernel(BuildArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const HnswDeviceGraph& g = a.g;
    const int q = blockIdx.x, lane = threadIdx.x;
    const int qfloats = DistTraits<SPACE>::kU8 ? 32 : g.ldv;
    float* qv = reinterpret_cast<float*>(smem);                 // [ldv]
    int* nbr = reinterpret_cast<int*>(qv + qfloats);            // [32]
    float* nd = reinterpret_cast<float*>(nbr + 32);             // [32]
    int* xid = reinterpret_cast<int*>(nd + 32);                 // [XCAP + 32] kept + newly accepted
    float* xdv = reinterpret_cast<float*>(xid + XCAP + 32);     // [XCAP + 32]
    int* tid_ = reinterpret_cast<int*>(xdv + XCAP + 32);        // [XCAP + 32] rank-sort scratch
    float* tdv = reinterpret_cast<float*>(tid_ + XCAP + 32);    // [XCAP + 32]

    int nx = 0;
    if (q > 0) {
        const int p = a.pts[q];
        const int nc = a.cand_n[q];
        float thr = nc >= a.stride ? a.cand_d[(size_t)q * a.stride + nc - 1] : INFINITY;  // full list: must beat its worst
        int qnorm;
        stage_row<SPACE>(g, p, qv, qnorm, lane);
        for (int base = 0; base < q; base += 32) {
            const int m = min(32, q - base);
            if (lane < m) nbr[lane] = a.pts[base + lane];
            __builtin_amdgcn_wave_barrier();
            frontier_distances<SPACE>(g, qv, reinterpret_cast<const uint8_t*>(qv), qnorm, nbr, nd, m, lane);
            const float d = lane < m ? nd[lane] : INFINITY;
            const bool acc = lane < m && d < thr;
            const u64 am = __ballot(acc);
            if (am == 0) continue;
            const int na = __popcll(am);
            if (acc) {
                const int slot = nx + __popcll(am & ((1ull << lane) - 1ull));
                xid[slot] = nbr[lane];
                xdv[slot] = d;
            }
            __builtin_amdgcn_wave_barrier();
            // rank-sort the nx + na entries (distance, then arrival order), keep the XCAP closest
            const int tot = nx + na;
            for (int i = lane; i < tot; i += 64) {
                tid_[i] = xid[i];
                tdv[i] = xdv[i];
            }
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < tot; i += 64) {
                const float di = tdv[i];
                int r = 0;
                for (int j = 0; j < tot; ++j) {
                    const float dj = tdv[j];
                    r += (dj < di || (dj == di && j < i)) ? 1 : 0;
                }
                if (r < XCAP) {
                    xid[r] = tid_[i];
                    xdv[r] = di;
                }
            }
            nx = tot < XCAP ? tot : XCAP;
            __builtin_amdgcn_wave_barrier();
            if (nx == XCAP) thr = fminf(thr, xdv[XCAP - 1]);
        }
    }
    if (lane == 0) a.extra_n[q] = nx;
    if (lane < nx) {
        a.extra_ids[(size_t)q * XCAP + lane] = xid[lane];
        a.extra_d[(size_t)q * XCAP + lane] = xdv[lane];
    }
}

// first request of every target in the sorted list -> active[]
__global__ void hnsw_build_heads_kernel(const u64* keys, int total, int32_t* active, int32_t* nactive) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const u64 k = keys[i];
    if (k == ~0ull) return;
    if (i == 0 || (uint32_t)(keys[i - 1] >> 32) != (uint32_t)(k >> 32)) active[atomicAdd(nactive, 1)] = i;
}

template <int SPACE>
__global__ __launch_bounds__(64) void hnsw_build_link_kernel(BuildArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const HnswDeviceGraph& g = a.g;
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= *a.nactive) return;
    constexpr int W = 128;  // list width: maxM0 <= 126 friends + the new node
    const int qfloats = DistTraits<SPACE>::kU8 ? 32 : g.ldv;
    float* qv = reinterpret_cast<float*>(smem);                 // [ldv]
    float* nd = qv + qfloats;                                   // [W]
    int* kept_id = reinterpret_cast<int*>(nd + W);              // [W]
    float* kept_d = reinterpret_cast<float*>(kept_id + W);      // [W]
    int* fl = reinterpret_cast<int*>(kept_d + W);               // [W] current friends
    int* sc_id = fl + W;                                        // [W] friends + new node, sorted
    float* sc_d = reinterpret_cast<float*>(sc_id + W);          // [W]
    float* td = sc_d + W;                                       // [W] unsorted distances of the same

    const int first = a.active[blockIdx.x];
    const int t = (int)(uint32_t)(a.req_key[first] >> 32);
    int32_t* L = adj_list(a, t);
    const int maxsz = a.level > 0 ? g.maxM : g.maxM0;
    int cnt = L[0];
    for (int i = lane; i < cnt; i += 64) fl[i] = L[1 + i];
    __builtin_amdgcn_wave_barrier();

    // this target's requests: consecutive in the sorted list, ascending new-node id
    for (int r = first; r < a.req_total; ++r) {
        const u64 rk = a.req_key[r];
        if ((int)(uint32_t)(rk >> 32) != t || rk == ~0ull) break;
        const int p = (int)(uint32_t)rk;
        const float dp = a.req_dist[r];
        if (cnt < maxsz) {
            if (lane == 0) fl[cnt] = p;
            cnt++;
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        // full: distances of the centre t to its friends, the new node joins, heuristic over cnt+1
        int qnorm;
        stage_row<SPACE>(g, t, qv, qnorm, lane);
        frontier_distances<SPACE>(g, qv, reinterpret_cast<const uint8_t*>(qv), qnorm, fl, nd, cnt, lane);
        const int n1 = cnt + 1;
        for (int i = lane; i < cnt; i += 64) td[i] = nd[i];
        if (lane == 0) {
            td[cnt] = dp;
            fl[cnt] = p;   // (slot cnt <= 126 is free: the list is rebuilt below)
        }
        __builtin_amdgcn_wave_barrier();
        // rank sort by (distance, list position), up to two items per lane
        const float da = lane < n1 ? td[lane] : INFINITY, db = lane + 64 < n1 ? td[lane + 64] : INFINITY;
        int ra = 0, rb = 0;
        for (int j = 0; j < n1; ++j) {
            const float dj = td[j];
            ra += (dj < da || (dj == da && j < lane)) ? 1 : 0;
            rb += (dj < db || (dj == db && j < lane + 64)) ? 1 : 0;
        }
        if (lane < n1) {
            sc_id[ra] = fl[lane];
            sc_d[ra] = da;
        }
        if (lane + 64 < n1) {
            sc_id[rb] = fl[lane + 64];
            sc_d[rb] = db;
        }
        __builtin_amdgcn_wave_barrier();
        const int nk = heuristic2<SPACE>(g, sc_id, sc_d, n1, n1 - 1, a.delaunay, qv, nd, kept_id, kept_d, lane);
        // refilled farthest first (hnsw.h:295-300)
        for (int i = lane; i < nk; i += 64) fl[i] = kept_id[nk - 1 - i];
        cnt = nk;
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) L[0] = cnt;
    for (int i = lane; i < maxsz; i += 64) L[1 + i] = i < cnt ? fl[i] : 0;
}

// start node of each (node, level) search = closest result of the same node's search one level up
__global__ void hnsw_build_starts_kernel(const int32_t* src, const int32_t* cand_ids, const int32_t* cand_n,
                                         int stride, int32_t* starts, int m) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int sp = src[i];
    starts[i] = (sp >= 0 && cand_n[sp] > 0) ? cand_ids[(size_t)sp * stride] : -1;
}

hipError_t launch_hnsw_build_starts(const int32_t* src, const int32_t* cand_ids, const int32_t* cand_n, int stride,
                                    int32_t* starts, int m, hipStream_t s) {
    if (m <= 0) return hipSuccess;
    hipLaunchKernelGGL(hnsw_build_starts_kernel, dim3((m + 255) / 256), dim3(256), 0, s, src, cand_ids, cand_n,
                       stride, starts, m);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
template <typename Kern>
static hipError_t launch_build(Kern kern, const BuildArgs& a, int grid, size_t lds, hipStream_t s) {
    if (grid <= 0) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, s, a);
    return hipGetLastError();
}

#define BUILD_DISPATCH(KERNEL, SPACEVAR, ARGS, GRID, LDS, STREAM)                                  \
    switch (SPACEVAR) {                                                                            \
        case SP_L2SQR: return launch_build(KERNEL<SP_L2SQR>, ARGS, GRID, LDS, STREAM);             \
        case SP_L2: return launch_build(KERNEL<SP_L2>, ARGS, GRID, LDS, STREAM);                   \
        case SP_L1: return launch_build(KERNEL<SP_L1>, ARGS, GRID, LDS, STREAM);                   \
        case SP_LINF: return launch_build(KERNEL<SP_LINF>, ARGS, GRID, LDS, STREAM);               \
        case SP_NORMCOS: return launch_build(KERNEL<SP_NORMCOS>, ARGS, GRID, LDS, STREAM);         \
        case SP_COSINE: return launch_build(KERNEL<SP_COSINE>, ARGS, GRID, LDS, STREAM);           \
        case SP_ANGULAR: return launch_build(KERNEL<SP_ANGULAR>, ARGS, GRID, LDS, STREAM);         \
        case SP_NEGDOT: return launch_build(KERNEL<SP_NEGDOT>, ARGS, GRID, LDS, STREAM);           \
        case SP_L2SQR_SIFT: return launch_build(KERNEL<SP_L2SQR_SIFT>, ARGS, GRID, LDS, STREAM);   \
        default: return hipErrorInvalidValue;                                                      \
    }

static BuildArgs base_args(const HnswBuildGraph& bg, int level) {
    BuildArgs a{};
    a.g = bg.g;
    a.links0 = bg.links0;
    a.up_links = bg.up_links;
    a.M = bg.M;
    a.delaunay = bg.delaunay;
    a.level = level;
    return a;
}

hipError_t launch_hnsw_build_select(const HnswBuildGraph& bg, int level, const int32_t* pts, int npts,
                                    const int32_t* cand_ids, const float* cand_d, const int32_t* cand_n,
                                    int stride, const int32_t* extra_ids, const float* extra_d,
                                    const int32_t* extra_n, unsigned long long* req_key, float* req_dist,
                                    hipStream_t s) {
    if (npts <= 0) return hipSuccess;
    // unused slots keep the pad key ~0 and sort behind every real request
    hipError_t e = hipMemsetAsync(req_key, 0xFF, (size_t)npts * bg.M * 8, s);
    if (e != hipSuccess) return e;
    BuildArgs a = base_args(bg, level);
    a.pts = pts;
    a.npts = npts;
    a.cand_ids = cand_ids;
    a.cand_d = cand_d;
    a.cand_n = cand_n;
    a.stride = stride;
    a.req_key = req_key;
    a.req_dist = req_dist;
    a.extra_ids = const_cast<int32_t*>(extra_ids);
    a.extra_d = const_cast<float*>(extra_d);
    a.extra_n = const_cast<int32_t*>(extra_n);
    const size_t qbytes = bg.g.space == SP_L2SQR_SIFT ? 128 : (size_t)bg.g.ldv * 4;
    const size_t lds = qbytes + 3 * 64 * 4 + (size_t)stride * 8 + 16;
    BUILD_DISPATCH(hnsw_build_select_kernel, bg.g.space, a, npts, lds, s)
}

hipError_t launch_hnsw_build_mates(const HnswBuildGraph& bg, int level, const int32_t* pts, int npts,
                                   const float* cand_d, const int32_t* cand_n, int stride, int32_t* extra_ids,
                                   float* extra_d, int32_t* extra_n, hipStream_t s) {
    if (npts <= 0) return hipSuccess;
    BuildArgs a = base_args(bg, level);
    a.pts = pts;
    a.npts = npts;
    a.cand_d = cand_d;
    a.cand_n = cand_n;
    a.stride = stride;
    a.extra_ids = extra_ids;
    a.extra_d = extra_d;
    a.extra_n = extra_n;
    const size_t qbytes = bg.g.space == SP_L2SQR_SIFT ? 128 : (size_t)bg.g.ldv * 4;
    const size_t lds = qbytes + 2 * 32 * 4 + 4 * (size_t)(XCAP + 32) * 4 + 16;
    BUILD_DISPATCH(hnsw_build_mates_kernel, bg.g.space, a, npts, lds, s)
}

size_t hnsw_build_sort_temp_bytes(int max_requests, int n) {
    size_t tmp = 0;
    int bits = 32;
    while (bits < 64 && (1ull << (bits - 32)) < (unsigned long long)n + 1) ++bits;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                    (float*)nullptr, (float*)nullptr, (size_t)max_requests, 0, 64, nullptr, false);
    (void)bits;
    return tmp + 256;
}

// (target, new node) order of the requests of one level + the first request of every target
hipError_t launch_hnsw_build_sort_requests(const unsigned long long* req_key, const float* req_dist, int total,
                                           unsigned long long* key_sorted, float* dist_sorted, void* temp,
                                           size_t temp_bytes, int32_t* active, int32_t* nactive, hipStream_t s) {
    if (total <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(nactive, 0, 4, s);
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_pairs(temp, temp_bytes, req_key, key_sorted, req_dist, dist_sorted, (size_t)total, 0, 64, s,
                                  false);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(hnsw_build_heads_kernel, dim3((total + 255) / 256), dim3(256), 0, s, key_sorted, total, active,
                       nactive);
    return hipGetLastError();
}

hipError_t launch_hnsw_build_link(const HnswBuildGraph& bg, int level, const int32_t* active,
                                  const int32_t* nactive, int max_active, const unsigned long long* key_sorted,
                                  const float* dist_sorted, int total, hipStream_t s) {
    BuildArgs a = base_args(bg, level);
    a.active = const_cast<int32_t*>(active);
    a.nactive = const_cast<int32_t*>(nactive);
    a.req_key = const_cast<u64*>(key_sorted);
    a.req_dist = const_cast<float*>(dist_sorted);
    a.req_total = total;
    const size_t qbytes = bg.g.space == SP_L2SQR_SIFT ? 128 : (size_t)bg.g.ldv * 4;
    const size_t lds = qbytes + 7 * 128 * 4 + 16;
    BUILD_DISPATCH(hnsw_build_link_kernel, bg.g.space, a, max_active, lds, s)
}

}  // namespace gfxknn
