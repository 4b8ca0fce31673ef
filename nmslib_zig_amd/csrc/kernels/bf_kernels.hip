// Brute-force (sequential-search) path on gfx950.
//
// Replaces SeqSearch::Search (src/method/seqsearch.cc:143-150), i.e. one virtual
// HiddenDistance call per (query, object) + KNNQueue::Push, by
//   1. bf_select_*  : a tiled Q x B^T contraction on the matrix cores
//                     (v_mfma_f32_32x32x2_f32 / v_mfma_i32_32x32x32_i8) whose epilogue keeps,
//                     per (query, row-split), the k' best rows by a ranking score that is
//                     monotone in the reference distance;
//   2. bf_rerank    : the reference's own distance formula on those survivors, canonical
//                     (distance, position) order, top k  (KNNQuery::CheckAndAddToResult,
//                     knnquery.cc:66-75 + extract_knn_results, nmslib_c.cpp:293-328).
//
// Work decomposition: workgroup = 128 queries (4 waves x 32) x one row split.  A wave keeps
// its 32 queries as the MFMA B operand in registers (D <= 128) and streams base rows through
// a double-buffered, padded LDS tile (64 rows x <=128 floats per step).  With A = base rows
// and B = queries, accumulator register r of lane l is the score of query (l & 31) against
// row (r&3) + 8*(r>>2) + 4*(l>>5): every lane owns ONE query, so the running threshold of
// that query is a lane-local register and the epilogue is one compare per element.
// Survivors are appended to a per-(query, split) buffer in HBM (L2-resident); when a buffer
// nears capacity the owning wave sorts it in LDS, keeps k', and raises the threshold.
#include <cstdio>
#include <type_traits>
#include <vector>
#include <cstdlib>

#include "common_dev.hpp"
#include "kernels.hpp"

namespace gfxknn {

enum BfMode : int { BF_L2 = 0, BF_DOT = 1, BF_COS = 2, BF_L1 = 3, BF_LINF = 4, BF_COSC = 5, BF_L2D = 6 };  // L2D: sum (a-b)^2 by VALU

struct BfArgs {
    const float* base;
    const float* aux;
    const float* qaux;  // BF_COSC: [qpad][4] = |q'|^2, |q| - |mu|, |q|, 1 (0 for a zero-norm query)
    int aux_stride;     // BF_COSC: aux holds three planes of this many floats (-|b'|^2, |b| - |mu|, 1/|b|)
    const float* queries;
    u64* cand;
    int* cand_cnt;
    uint32_t* gthr;  // [qpad] best published threshold per query (order-preserving uint, 0 = none)
    uint32_t* gq;    // [qpad][nsplit] counted bounds: 2*xj rows of that split score >= the value (0 = none)
    int xj, xm;      // counted-bound exchange: per-lane rank published / rank taken over the splits (xj = 0: off)
    int n, ldb, nqt, nsplit, rows_per_split, kprime, cap;
    int kcs;      // floats staged per K-chunk = min(ldb, 128)
    int nchunks;  // ceil(ldb / 128)
    int dbg;      // NMSLIB_GPU_DEBUG bits (timing experiments only): 1 skip epilogue, 2 skip staging, 4 no stagger, 8 no barrier
    const int* tile_fail;  // fallback launch of the f32 fast path: query tiles whose group flag is 0 have nothing to do
    int fail_group;
};

// row handled by accumulator register r of a lane in half h (C/D map of the 32x32 MFMAs)
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Candidate buffer of one (query, split): `cap` keys, split into two halves of cap/2.  The two
// lanes that own a query (l and l+32) append to their own half with a private register counter:
// no atomics and no cross-lane traffic on the append path.
//
// Compaction (wave-cooperative, rare): gather n0 keys of half 0 and n1 keys of half 1, keep the
// k' largest in half 0 (sorted, best first), return the new threshold as an order-preserving
// uint32 (0 when fewer than k' keys exist).
__device__ __forceinline__ u64 wave_max_u64(u64 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const u64 other = __shfl_xor(v, o, 64);
        v = other > v ? other : v;
    }
    return v;
}

// ascending bitonic sort of 128 64-bit keys held two per lane (element 2 * lane + i), registers and shuffles only: no LDS,
// no barrier.  Afterwards lane L holds elements 2L and 2L + 1 of the sorted order.
__device__ __forceinline__ void wave_sort128_u64(u64& k0, u64& k1, int lane) {
#pragma unroll
    for (int k2 = 2; k2 <= 128; k2 <<= 1) {
        const bool asc = ((2 * lane) & k2) == 0;   // (both elements of a lane lie in the same run: k2 >= 2)
#pragma unroll
        for (int j = k2 >> 1; j >= 1; j >>= 1) {
            if (j == 1) {
                const u64 lo = k0 < k1 ? k0 : k1, hi = k0 < k1 ? k1 : k0;
                k0 = asc ? lo : hi;
                k1 = asc ? hi : lo;
            } else {
                const int lj = j >> 1;   // the partner element sits in lane ^ lj, same slot
                const bool keep_min = ((lane & lj) == 0) == asc;
                const u64 o0 = __shfl_xor(k0, lj, 64), o1 = __shfl_xor(k1, lj, 64);
                k0 = keep_min ? (o0 < k0 ? o0 : k0) : (o0 > k0 ? o0 : k0);
                k1 = keep_min ? (o1 < k1 ? o1 : k1) : (o1 > k1 ? o1 : k1);
            }
        }
    }
}

__device__ __forceinline__ u64 gather_key(const u64* g, int idx, int n0, int n1, int half) {
    if (idx < n0) return g[idx];
    if (idx - n0 < n1) return g[half + idx - n0];
    return 0ull;
}

// n0 + n1 <= 256 and k' <= 32: four keys per lane in registers, k' rounds of (lane max, wave
// max, retire the winner).  Keys are unique (they embed the position): one retires per round.
__device__ __forceinline__ uint32_t compact_small(u64* g, int n0, int n1, int half, int kprime, int lane,
                                                  int* keep_out) {
    const int n = n0 + n1;
    u64 r0 = gather_key(g, lane, n0, n1, half);
    u64 r1 = gather_key(g, lane + 64, n0, n1, half);
    u64 r2 = gather_key(g, lane + 128, n0, n1, half);
    u64 r3 = gather_key(g, lane + 192, n0, n1, half);
    const int keep = n < kprime ? n : kprime;
    u64 mine = 0ull;
    for (int t = 0; t < keep; ++t) {
        const u64 a01 = r0 > r1 ? r0 : r1, a23 = r2 > r3 ? r2 : r3;
        const u64 w = wave_max_u64(a01 > a23 ? a01 : a23);
        r0 = r0 == w ? 0ull : r0;
        r1 = r1 == w ? 0ull : r1;
        r2 = r2 == w ? 0ull : r2;
        r3 = r3 == w ? 0ull : r3;
        if (lane == t) mine = w;
    }
    if (lane < keep) g[lane] = mine;
    const u64 kth = __shfl(mine, kprime - 1, 64);
    *keep_out = keep;
    return n >= kprime ? (uint32_t)(kth >> 32) : 0u;
}

// general case: bitonic sort of the gathered keys in this wave's LDS scratch
__device__ __forceinline__ uint32_t compact_sort(u64* g, int n0, int n1, int half, int kprime, u64* scratch,
                                                 int lane, int* keep_out) {
    const int n = n0 + n1;
    const int P = next_pow2(n < 2 ? 2 : n);
    for (int i = lane; i < P; i += 64) scratch[i] = gather_key(g, i, n0, n1, half);
    __builtin_amdgcn_wave_barrier();
    wave_bitonic_u64(scratch, P, lane, /*descending=*/true);
    const int keep = n < kprime ? n : kprime;
    for (int i = lane; i < keep; i += 64) g[i] = scratch[i];
    const uint32_t thr = (n >= kprime) ? (uint32_t)(scratch[kprime - 1] >> 32) : 0u;
    *keep_out = keep;
    __builtin_amdgcn_wave_barrier();
    return thr;
}

__device__ __forceinline__ bool need_pair(bool need) {  // either lane of the query asks
    const u64 m = __ballot(need);
    const int lane = threadIdx.x & 63;
    return ((m | (m >> 32) | (m << 32)) >> lane) & 1ull;
}

// Compact every query whose lanes ask for it (need), update the per-lane counters (mycnt) and
// return, per lane, the new threshold of ITS query as ord bits (0 = unchanged).
__device__ __forceinline__ uint32_t compact_queries(u64* gbase /* query 0 of this wave */, size_t qstride, int cap,
                                                    int kprime, u64* scratch, bool need, int& mycnt, int lane) {
    const int half = cap >> 1, l31 = lane & 31;
    if (__any(need)) {
        // cheap case, all queries at once: the two halves together hold no more than k' keys ->
        // the upper lane appends its keys behind the lower lane's, nothing is dropped
        const int n_other = __shfl_xor(mycnt, 32, 64);
        const bool fits = need_pair(need) && (mycnt + n_other <= kprime);
        if (fits) {
            if (lane >= 32) {
                u64* g = gbase + (size_t)l31 * qstride;
                for (int i = 0; i < mycnt; ++i) g[n_other + i] = g[half + i];
                mycnt = 0;
            } else {
                mycnt += n_other;
            }
            need = false;
        }
    }
    u64 m = __ballot(need);
    m = (m | (m >> 32)) & 0xFFFFFFFFull;  // either lane of a query may ask
    uint32_t my_thr = 0u;
    if (m) {
        // this wave's own appended keys must be visible to its loads below
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    while (m) {
        const int q = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int n0 = __shfl(mycnt, q, 64), n1 = __shfl(mycnt, q + 32, 64);
        u64* g = gbase + (size_t)q * qstride;
        int keep;
        uint32_t t;
        if (cap <= 256 && kprime <= 32) t = compact_small(g, n0, n1, half, kprime, lane, &keep);
        else t = compact_sort(g, n0, n1, half, kprime, scratch, lane, &keep);
        if (l31 == q) {
            mycnt = (lane < 32) ? keep : 0;
            my_thr = t;
        }
    }
    return my_thr;
}

// Counted bound shared by the row splits of one query.  Every split publishes a score v_s such
// that at least 2*xj of ITS rows score >= v_s (the xj-th best seen by each of the query's two
// lanes).  If T is the xm-th largest published value, 2*xj*xm >= k' rows of the whole base score
// >= T, so the global k'-th best is >= T and every row scoring below T can be dropped -- a bound
// of global quality (pass probability ~ k'/rows seen by ALL splits) where a split's own k'-th
// best only gives k'/rows seen by THAT split.  Values only grow, so a stale read just prunes less;
// loads/stores are agent-scope because the splits of a query run on different XCDs (own L2s).
// Both lanes of a query call this; each scans half of the splits, then the two top lists merge.
__device__ __forceinline__ uint32_t counted_bound(uint32_t* row, int nsplit, int split, int h, uint32_t pu,
                                                  int xm) {
    if (h == 0) __hip_atomic_store(row + split, pu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t tm[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) tm[i] = 0u;
    auto ins = [&](uint32_t v) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t hi = tm[i] > v ? tm[i] : v;
            v = tm[i] > v ? v : tm[i];
            tm[i] = hi;
        }
    };
    const int per = nsplit >> 2;  // 64-bit words per lane (nsplit is a multiple of 8)
    const u64* r64 = reinterpret_cast<const u64*>(row) + h * per;
    for (int i = 0; i < per; ++i) {
        const u64 w = __hip_atomic_load(r64 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int s0 = 2 * (h * per + i);
        // own slot: the value in the register (the store above may not be visible yet, and must not count twice)
        ins(s0 == split ? pu : (uint32_t)w);
        ins(s0 + 1 == split ? pu : (uint32_t)(w >> 32));
    }
    uint32_t o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = __shfl_xor(tm[i], 32, 64);
#pragma unroll
    for (int i = 0; i < 8; ++i) ins(o[i]);
    uint32_t t = tm[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) t = (i == xm - 1) ? tm[i] : t;
    return t;
}

// FULL: every staged K-chunk holds exactly 128 floats (ldb % 128 == 0): the MFMA loop is
// branch-free.  ONE: the whole row is one chunk (ldb == 128): the query fragments are loaded
// once, outside the row loop.
//
// Software pipeline inside ONE wave (MFMA modes): the scores of stage s stay in one accumulator
// set while the MFMAs of stage s+1 fill the other; the selection epilogue of stage s (compare,
// rare append) is issued between those MFMAs, two elements per 8-MFMA group.  A 32x32x2 f32
// MFMA keeps the matrix pipe busy for 64 cycles but the issue port for only a few, so the
// epilogue's VALU/SALU work rides in the MFMA shadow instead of idling the pipe.
__device__ long long g_bf_clk[2];  // NMSLIB_GPU_DEBUG & 1024: shader cycles / 100 MHz ticks of block 0
__device__ long long g_bf_trace[2048][3];  // ... and per block: start tick, end tick, HW_ID

template <int MODE, bool FULL, bool ONE>
__global__ __launch_bounds__(256, 2) void bf_select_f32_kernel(BfArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long clk_c0 = __builtin_readcyclecounter(), clk_r0 = wall_clock64();
    const int h = lane >> 5, l31 = lane & 31;

    // XCD-aware mapping: blocks b and b+8 share an XCD (round-robin dispatch), so the
    // nqt query tiles of one row split get ids that differ by multiples of 8 and re-read
    // that split's rows from the same L2.
    const int b = blockIdx.x;
    const int xcd = b & 7, rest = b >> 3;
    const int qt = rest % a.nqt;
    const int split = (rest / a.nqt) * 8 + xcd;
    if (split >= a.nsplit) return;
    if (a.tile_fail && a.tile_fail[qt / a.fail_group] == 0) return;

    constexpr bool kDirect = (MODE == BF_L1 || MODE == BF_LINF || MODE == BF_L2D);
    constexpr bool kDelay = !kDirect;  // epilogue of stage s overlapped with the MFMAs of s+1
    // BF_COSC = cosine / angular on data with a large common offset: the tile holds CENTRED rows b' = b - mu and the
    // score is -(1 - cos)|q| rebuilt from small quantities only,
    //     1 - cos(q,b) = (|q'-b'|^2 - (|q|-|b|)^2) / (2|q||b|),    |q'-b'|^2 = |q'|^2 + |b'|^2 - 2 q'.b',
    // so the f32 rounding of the MFMA dot product is relative to |q'||b'| (the spread), not to |q||b| (the offset).
    constexpr bool kCosC = (MODE == BF_COSC);
    constexpr int kAuxN = kCosC ? 3 : 1;
    constexpr bool kAux = (MODE == BF_L2 || MODE == BF_COS || kCosC);

    const int kcs = FULL ? BF_KC : a.kcs;
    const int lds_stride = kcs + 4;  // +16 B pad: conflict-free ds_read_b128 of 32 rows
    float* tile = reinterpret_cast<float*>(smem);                      // [2][BN][lds_stride]
    float* auxs = tile + 2 * BF_BN * lds_stride;                       // [4][kAuxN][BN], 3 in rotation
    u64* scratch = reinterpret_cast<u64*>(auxs + 4 * BF_BN * 3) + (size_t)wave * a.cap;  // [4][cap]

    const int qidx = qt * BF_TQ + wave * 32 + l31;  // this lane's query (row of the padded batch)
    const int half = a.cap >> 1;
    // this lane's half of its query's candidate buffer, and its private fill counter
    u64* candq = a.cand + ((size_t)qidx * a.nsplit + split) * a.cap + (size_t)h * half;
    int mycnt = 0;
    u64* wave_cand = a.cand + ((size_t)(qt * BF_TQ + wave * 32) * a.nsplit + split) * a.cap;  // query 0 of this wave
    const size_t qstride = (size_t)a.nsplit * a.cap;
    // Small k' (<= 16): each lane also keeps the 8 best scores IT has seen, sorted, in
    // registers.  The 16 values held by the two lanes of a query are all >= min(their 8th
    // bests), so that minimum never exceeds the query's k'-th best score: a threshold that
    // tightens continuously with no memory traffic and no compaction.
    const bool use_top8 = a.kprime <= 16;
    const bool track8 = use_top8 || a.xj > 0;  // the counted-bound exchange publishes t8[xj-1]
    float t8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t8[i] = -INFINITY;
    float thr = -INFINITY;  // lane-local threshold of this lane's query

    const int r_begin = split * a.rows_per_split;
    const int r_end = min(a.n, r_begin + a.rows_per_split);
    const int nstages = r_end > r_begin ? (r_end - r_begin + BF_BN - 1) / BF_BN : 0;
    const int nchunks = ONE ? 1 : a.nchunks;
    const int nsteps = nstages * nchunks;

    // staging map: 32 threads per row (16 B each), 8 rows per pass, 8 passes = 64 rows
    const int sc = tid & 31, sr = tid >> 5;
    f32x4 stg[8];
    float stg_aux[kAuxN];
    float cq2 = 0.f, cqnp = 0.f, cqn = 0.f, cqf = 1.f;  // BF_COSC: this lane's query constants

    auto issue_loads = [&](int step) __attribute__((always_inline)) {
        const int stage = step / nchunks, kc = step - stage * nchunks;
        const int row0 = r_begin + stage * BF_BN;
        const int col0 = kc * BF_KC + sc * 4;
        const int kc_len = FULL ? BF_KC : min(kcs, a.ldb - kc * BF_KC);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = row0 + sr + 8 * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (sc * 4 < kc_len && row < r_end)
                v = *reinterpret_cast<const f32x4*>(a.base + (size_t)row * a.ldb + col0);
            stg[i] = v;
        }
        if (kAux && kc == 0 && tid < BF_BN) {
            // rows past the end of the split get -inf: L2 score = dot + (-inf), cosine score =
            // 0 * (-inf) = NaN -- neither can pass "s > thr", so those modes need no position test
            const int row = row0 + tid;
            stg_aux[0] = row < r_end ? a.aux[row] : -INFINITY;
            if constexpr (kCosC) {
                stg_aux[1] = row < r_end ? a.aux[(size_t)a.aux_stride + row] : 0.f;
                stg_aux[2] = row < r_end ? a.aux[2 * (size_t)a.aux_stride + row] : 1.f;
            }
        }
    };
    auto write_lds = [&](int step) __attribute__((always_inline)) {
        const int stage = step / nchunks, kc = step - stage * nchunks;
        float* t = tile + (step & 1) * BF_BN * lds_stride;
        if (sc * 4 < kcs) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<f32x4*>(t + (sr + 8 * i) * lds_stride + sc * 4) = stg[i];
        }
        // aux rotates over three buffers: the delayed epilogue of stage s still reads buffer
        // s % 3 while stage s+2's values are being written
        if (kAux && kc == 0 && tid < BF_BN) {
#pragma unroll
            for (int c = 0; c < kAuxN; ++c) auxs[((stage % 3) * kAuxN + c) * BF_BN + tid] = stg_aux[c];
        }
    };

    // one eighth of write_lds (rows sr + 8*i): issued between MFMA groups so that the staging of the
    // next tile costs no MFMA-free phase at the end of the stage
    auto write_lds_piece = [&](int step, int i) __attribute__((always_inline)) {
        float* t = tile + (step & 1) * BF_BN * lds_stride;
        if (sc * 4 < kcs) *reinterpret_cast<f32x4*>(t + (sr + 8 * i) * lds_stride + sc * 4) = stg[i];
    };
    auto write_lds_aux = [&](int step) __attribute__((always_inline)) {
        const int stage = step / nchunks, kc = step - stage * nchunks;
        if (kAux && kc == 0 && tid < BF_BN) {
#pragma unroll
            for (int c = 0; c < kAuxN; ++c) auxs[((stage % 3) * kAuxN + c) * BF_BN + tid] = stg_aux[c];
        }
    };
    // query fragments: lane (l31, h) holds dims 8t + 4h + {0..3} of its query, t = 0..15
    f32x4 bq[16];
    auto load_queries = [&](int kc) __attribute__((always_inline)) {
        const float* qrow = a.queries + (size_t)qidx * a.ldb + kc * BF_KC + 4 * h;
        const int kc_len = FULL ? BF_KC : min(kcs, a.ldb - kc * BF_KC);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (8 * t < kc_len) v = *reinterpret_cast<const f32x4*>(qrow + 8 * t);
            bq[t] = v;
        }
    };
    // direct (VALU) modes: every lane needs ALL dims of its query; the other half comes from
    // the partner lane (l ^ 32) once per K-chunk.  qlo/qhi = dims 8t+{0..3} / 8t+{4..7}.
    f32x4 qlo[kDirect ? 16 : 1], qhi[kDirect ? 16 : 1];
    auto spread_queries = [&]() __attribute__((always_inline)) {
        if constexpr (kDirect) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = __shfl_xor(bq[t][j], 32, 64);
                qlo[t] = h == 0 ? bq[t] : o;
                qhi[t] = h == 0 ? o : bq[t];
            }
        }
    };

    // Appended keys wait in four registers and are stored right AFTER the staging wait of the
    // stage (flush_pending): vmcnt counts stores and loads together and in order, so a store
    // issued between the tile prefetch and its vmcnt wait would put its whole latency on the
    // critical path of every stage.
    u64 pend0 = 0, pend1 = 0, pend2 = 0, pend3 = 0;
    int npend = 0;
    auto flush_pending = [&]() __attribute__((always_inline)) {
        const int base = mycnt - npend;
        if (npend > 0 && base < half) candq[base] = pend0;
        if (npend > 1 && base + 1 < half) candq[base + 1] = pend1;
        if (npend > 2 && base + 2 < half) candq[base + 2] = pend2;
        if (npend > 3 && base + 3 < half) candq[base + 3] = pend3;
        npend = 0;
    };
    // ---- selection epilogue ----
    auto score_of = [&](float acc, int row, const float* ax) __attribute__((always_inline)) -> float {
        if constexpr (kDirect) return -acc;  // smaller distance = better score
        else if constexpr (MODE == BF_COS) return acc * ax[row];
        else if constexpr (kCosC) {
            const float t = cqnp - ax[BF_BN + row];
            const float inv = ax[2 * BF_BN + row];
            const float v = fmaf(t, t, fmaf(2.f, acc, ax[row] - cq2)) * (0.5f * inv);
            return cqf * (inv == 0.f ? -cqn : v);  // zero-norm row: similarity 0 (distcomp_scalar.cc:83-168)
        } else return acc;
    };
    auto consider = [&](float s, int pos, bool lastst) __attribute__((always_inline)) {
        bool pass = s > thr;
        if constexpr (!kAux) pass = pass && (!lastst || pos < r_end);
        if (pass) {
            const u64 key = make_sel_key(f32_ord(s), (uint32_t)pos);
            if (npend < 4) {
                pend3 = pend2;
                pend2 = pend1;
                pend1 = pend0;
                pend0 = key;
                npend++;
            } else {
                // more than four hits in one stage (early rows only): flush, then keep going
                flush_pending();
                pend0 = key;
                npend = 1;
            }
            mycnt++;
            if (track8 && s > t8[7]) {
                float v = s;  // sorted insert, best first
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float hi = fmaxf(t8[i], v);
                    v = fminf(t8[i], v);
                    t8[i] = hi;
                }
            }
        }
    };
    // compaction when a half-buffer is nearly full, or final
    auto finish_stage = [&](bool lastst) __attribute__((always_inline)) {
        const bool need = lastst || (mycnt > half - BF_BN / 2);
        if (__any(need)) flush_pending();  // compaction reads the buffers
        if (lastst && !(a.dbg & 64)) {
            // Final pass, every lane on its own half-buffer: drop the keys that the final
            // threshold (own bound and the bounds published by the other row splits) rules out.
            // What is left usually fits in k' and is concatenated without any sorting.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const uint32_t t_ord = thr > -INFINITY ? f32_ord(thr) : 0u;
            const int n = mycnt < half ? mycnt : half;
            int w = 0;
            for (int i = 0; i < n; i += 4) {
                u64 k0 = candq[i], k1 = i + 1 < n ? candq[i + 1] : 0ull;
                u64 k2 = i + 2 < n ? candq[i + 2] : 0ull, k3 = i + 3 < n ? candq[i + 3] : 0ull;
                if ((uint32_t)(k0 >> 32) >= t_ord) candq[w++] = k0;
                if (k1 && (uint32_t)(k1 >> 32) >= t_ord) candq[w++] = k1;
                if (k2 && (uint32_t)(k2 >> 32) >= t_ord) candq[w++] = k2;
                if (k3 && (uint32_t)(k3 >> 32) >= t_ord) candq[w++] = k3;
            }
            mycnt = w;
        }
        const uint32_t t_ord = compact_queries(wave_cand, qstride, a.cap, a.kprime, scratch, need, mycnt, lane);
        if (t_ord != 0u) thr = fmaxf(thr, ord_f32(t_ord));
    };
    // The 16 scores of one finished 32-row block.  Fast path: one max over the block and one
    // compare; the element-wise path runs only when some lane of the wave has a hit.
    auto check_block = [&](const f32x16& o, int blk, int stage) __attribute__((always_inline)) {
        const float* ax = auxs + (stage % 3) * kAuxN * BF_BN;
        const int row0 = r_begin + stage * BF_BN + blk * 32;
        const bool lastst = stage == nstages - 1;
        float sc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = score_of(o[r], blk * 32 + acc_row(r, h), ax);
        float m01 = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
        float m23 = fmaxf(fmaxf(sc[4], sc[5]), fmaxf(sc[6], sc[7]));
        float m45 = fmaxf(fmaxf(sc[8], sc[9]), fmaxf(sc[10], sc[11]));
        float m67 = fmaxf(fmaxf(sc[12], sc[13]), fmaxf(sc[14], sc[15]));
        const float mx = fmaxf(fmaxf(m01, m23), fmaxf(m45, m67));
        if (__any(mx > thr) && !(a.dbg & 32)) {
            // hits are sparse: look only into the 4-row groups that hold one
            const float m4[4] = {m01, m23, m45, m67};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (__any(m4[g] > thr)) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) consider(sc[4 * g + j], row0 + acc_row(4 * g + j, h), lastst);
                }
            }
            if (use_top8) thr = fmaxf(thr, fminf(t8[7], __shfl_xor(t8[7], 32, 64)));
            if (__any(mycnt > half - BF_BN / 2)) finish_stage(false);
        }
    };
    // Thresholds are shared between the row splits of a query through gthr (agent-scope atomic
    // max): ANY split's bound on its k'-th best score also bounds the global k'-th best, so a
    // foreign bound T may prune with "score >= T" (strict compare against the next lower float).
    // A stale or missing value only prunes less.  The reply of one exchange is consumed at the
    // next one, so its latency is never waited for.
    uint32_t g_reply = 0u;
    auto exchange_thr = [&]() __attribute__((always_inline)) {
        const uint32_t g = __shfl(g_reply, l31, 64);
        if (g > 1u) thr = fmaxf(thr, ord_f32(g - 1u));
        if (h == 0 && a.gthr) {
            const uint32_t mine = thr > -INFINITY ? f32_ord(thr) : 0u;
            const uint32_t old = atomicMax(a.gthr + qidx, mine);
            g_reply = old > mine ? old : mine;
        }
    };

    auto exchange_counted = [&]() __attribute__((always_inline)) {
        float mine = t8[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) mine = (i == a.xj - 1) ? t8[i] : mine;
        const float pv = fminf(mine, __shfl_xor(mine, 32, 64));
        const uint32_t pu = pv > -INFINITY ? f32_ord(pv) : 0u;
        const uint32_t t = counted_bound(a.gq + (size_t)qidx * a.nsplit, a.nsplit, split, h, pu, a.xm);
        if (t > 1u) thr = fmaxf(thr, ord_f32(t - 1u));
    };

    // ---- one 32-row block of the staged tile: 64 MFMAs into n; behind the first of them, the 16
    //      scores of the previously finished block o ----
    auto block_mfma = [&](f32x16& n, const float* ap, const float* next_ap, f32x4& f0, f32x4& f1, int kc_len, bool epi, const f32x16& o, int o_blk, int o_stage, int wr_step) __attribute__((always_inline)) {
        // Fragment reads run TWO 8-float groups (8 MFMAs) ahead of their use: with eight waves per CU
        // reading 1 KB each, one group of slack does not cover the LDS latency.  f0/f1 = fragments 0/1 of
        // this block, requested by the caller; on return they hold fragments 0/1 of the block at next_ap.
        // The first group is peeled so that the (large, rarely taken) score check sits outside the unrolled loop.
        f32x4 c1 = f1, c2 = f1;
        if (16 < kc_len) c2 = *reinterpret_cast<const f32x4*>(ap + 16);
        __builtin_amdgcn_sched_barrier(0);
        n = __builtin_amdgcn_mfma_f32_32x32x2f32(f0[0], bq[0][0], n, 0, 0, 0);
        n = __builtin_amdgcn_mfma_f32_32x32x2f32(f0[1], bq[0][1], n, 0, 0, 0);
        n = __builtin_amdgcn_mfma_f32_32x32x2f32(f0[2], bq[0][2], n, 0, 0, 0);
        n = __builtin_amdgcn_mfma_f32_32x32x2f32(f0[3], bq[0][3], n, 0, 0, 0);
        if (epi) check_block(o, o_blk, o_stage);
#pragma unroll
        for (int tt = 1; tt < 16; ++tt) {
            f32x4 c3 = c2;
            if (tt + 2 < 16) {
                if (8 * (tt + 2) < kc_len) c3 = *reinterpret_cast<const f32x4*>(ap + 8 * (tt + 2));
            } else if (next_ap) {
                c3 = *reinterpret_cast<const f32x4*>(next_ap + 8 * (tt + 2 - 16));
            }
            // keep the read ABOVE the MFMAs of this group: the machine scheduler otherwise sinks it next
            // to its first use (register pressure heuristic) and exposes the whole LDS latency
            __builtin_amdgcn_sched_barrier(0);
            if (8 * tt < kc_len) {
                n = __builtin_amdgcn_mfma_f32_32x32x2f32(c1[0], bq[tt][0], n, 0, 0, 0);
                n = __builtin_amdgcn_mfma_f32_32x32x2f32(c1[1], bq[tt][1], n, 0, 0, 0);
                n = __builtin_amdgcn_mfma_f32_32x32x2f32(c1[2], bq[tt][2], n, 0, 0, 0);
                n = __builtin_amdgcn_mfma_f32_32x32x2f32(c1[3], bq[tt][3], n, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (FULL && wr_step >= 0 && (tt & 1)) write_lds_piece(wr_step, tt >> 1);
            c1 = c2;
            c2 = c3;
        }
        if (FULL && wr_step >= 0) write_lds_aux(wr_step);
        f0 = c1;
        f1 = c2;
    };
    // -0.5*||b||^2 folded in with one more MFMA: A = (aux | 0), B = (1 | 0)
    auto add_norm = [&](f32x16& n, int blk, int stage) __attribute__((always_inline)) {
        if constexpr (MODE == BF_L2) {
            const float* ax = auxs + (stage % 3) * BF_BN;
            const float one = h == 0 ? 1.f : 0.f;
            const float x = h == 0 ? ax[blk * 32 + l31] : 0.f;
            n = __builtin_amdgcn_mfma_f32_32x32x2f32(x, one, n, 0, 0, 0);
        }
    };
    // direct (VALU) modes: lanes of one half read the same LDS addresses (broadcast reads); each
    // lane accumulates full-dimension |a-b| for its own 16 rows of the block
    auto block_direct = [&](f32x16& n, const float* t, int blk, int kc_len) __attribute__((always_inline)) {
        if constexpr (kDirect) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* ap = t + (blk * 32 + acc_row(r, h)) * lds_stride;
                float s = n[r];
#pragma unroll
                for (int tt = 0; tt < 16; ++tt) {
                    if (8 * tt < kc_len) {
                        const f32x4 alo = *reinterpret_cast<const f32x4*>(ap + 8 * tt);
                        const f32x4 ahi = *reinterpret_cast<const f32x4*>(ap + 8 * tt + 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if constexpr (MODE == BF_L2D) {   // the reference's own form: differences first (no cancellation)
                                const float t0 = alo[j] - qlo[tt][j], t1 = ahi[j] - qhi[tt][j];
                                s = fmaf(t1, t1, fmaf(t0, t0, s));
                            } else {
                                const float d0 = fabsf(alo[j] - qlo[tt][j]);
                                const float d1 = fabsf(ahi[j] - qhi[tt][j]);
                                s = (MODE == BF_L1) ? (s + d0) + d1 : fmaxf(fmaxf(s, d0), d1);
                            }
                        }
                    }
                }
                n[r] = s;
            }
        }
    };

    // X holds block 0 of a stage, Y block 1.  MFMA modes: Y(stage s-1) is examined under the MFMAs
    // of X(stage s), X(stage s) under the MFMAs of Y(stage s).
    f32x16 accX, accY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        accX[i] = 0.f;
        accY[i] = 0.f;
    }

    if (nsteps > 0) {
        issue_loads(0);
        write_lds(0);
    }
    if constexpr (kCosC) {
        const f32x4 qa = *reinterpret_cast<const f32x4*>(a.qaux + (size_t)qidx * 4);
        cq2 = qa[0];
        cqnp = qa[1];
        cqn = qa[2];
        cqf = qa[3];
    }
    if (nchunks == 1) {
        load_queries(0);
        // Retire the query loads HERE, with the builtin the waitcnt pass models: otherwise it
        // keeps a progressive vmcnt(N) ladder inside the row loop (first use of each fragment)
        // that also drains the next tile's prefetch far too early.
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only
        spread_queries();
    }
    __syncthreads();

    const bool skip_epi = (a.dbg & 1) != 0;
    const int xmask = (a.dbg >> 12) ? ((1 << ((a.dbg >> 12) & 7)) - 1) : 3;  // threshold exchange every (xmask+1) stages
    for (int step = 0; step < nsteps; ++step) {
        const int stage = step / nchunks, kc = step - stage * nchunks;
        const bool have_next = step + 1 < nsteps;
        if (have_next && !((a.dbg & 2) && step > 0)) issue_loads(step + 1);
        if (!ONE && nchunks > 1) {
            load_queries(kc);
            spread_queries();
        }
        const int kc_len = FULL ? BF_KC : min(kcs, a.ldb - kc * BF_KC);
        const float* t = tile + (step & 1) * BF_BN * lds_stride;
        const bool first = kc == 0, lastc = kc == nchunks - 1;
        if constexpr (!kDirect) {
            const float* ap0 = t + l31 * lds_stride + 4 * h;
            if (first) {
#pragma unroll
                for (int i = 0; i < 16; ++i) accX[i] = 0.f;
            }
            f32x4 fr0 = *reinterpret_cast<const f32x4*>(ap0), fr1 = fr0;
            if (8 < kc_len) fr1 = *reinterpret_cast<const f32x4*>(ap0 + 8);
            block_mfma(accX, ap0, ap0 + 32 * lds_stride, fr0, fr1, kc_len, first && stage > 0 && !skip_epi, accY, 1,
                       stage - 1, -1);
            if (lastc) add_norm(accX, 0, stage);
            if (first) {
#pragma unroll
                for (int i = 0; i < 16; ++i) accY[i] = 0.f;
            }
            const bool stage_next = have_next && !((a.dbg & 2) && step > 0);
            block_mfma(accY, ap0 + 32 * lds_stride, nullptr, fr0, fr1, kc_len, lastc && !skip_epi, accX, 0, stage,
                       (FULL && stage_next) ? step + 1 : -1);
            if (lastc) add_norm(accY, 1, stage);
            if (skip_epi) asm volatile("" ::"v"(accX[0]), "v"(accY[0]), "v"(accX[15]), "v"(accY[15]));
        } else {
            if (first) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    accX[i] = 0.f;
                    accY[i] = 0.f;
                }
            }
            block_direct(accX, t, 0, kc_len);
            block_direct(accY, t, 1, kc_len);
            if (lastc && !skip_epi) {
                check_block(accX, 0, stage);
                check_block(accY, 1, stage);
                if (stage == nstages - 1) {
                    if (a.xj > 0 && !(a.dbg & (256 | 512))) exchange_counted();
                    finish_stage(true);
                }
            }
        }
        if ((!FULL || kDirect) && have_next && !((a.dbg & 2) && step > 0)) write_lds(step + 1);
        flush_pending();  // after the staging wait: these stores have a whole stage to retire
        if (lastc && !(a.dbg & 256)) {
            if (a.xj > 0 && !(a.dbg & 512)) {
                // the bound improves like 1/rows seen: exchange at stages 0,1,3,7,15,.. and every 32nd
                if ((stage & (stage + 1)) == 0 || (stage & 31) == 31) exchange_counted();
            } else if ((stage & xmask) == xmask) {
                exchange_thr();
            }
        }
        if (!(a.dbg & 8)) __syncthreads();
    }
    // drain: block 1 of the last stage has not been examined yet; then the final compaction
    if (kDelay && nstages > 0 && !skip_epi) {
        check_block(accY, 1, nstages - 1);
        if (a.xj > 0 && !(a.dbg & (256 | 512))) exchange_counted();
        if (!(a.dbg & 16)) finish_stage(true);
    }
    if (h == 0) a.cand_cnt[(size_t)qidx * a.nsplit + split] = mycnt;
    if ((a.dbg & 1024) && tid == 0) {
        if (blockIdx.x == 0) {
            g_bf_clk[0] = __builtin_readcyclecounter() - clk_c0;
            g_bf_clk[1] = wall_clock64() - clk_r0;
        }
        if (blockIdx.x < 2048) {
            g_bf_trace[blockIdx.x][0] = clk_r0;
            g_bf_trace[blockIdx.x][1] = wall_clock64();
            g_bf_trace[blockIdx.x][2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
        }
    }
}

// ---------------------------------------------------------------------------------------
// uint8 SIFT selection: v_mfma_i32_32x32x32_i8 on bytes re-centred to int8 (x ^ 0x80).
//   dot(a,b) = dot(a',b') + 128*sum(a) + 128*sum(b) - 128*128*128        (a' = a - 128)
//   dist     = ||a||^2 + ||b||^2 - 2 dot(a,b)          (distcomp_l2sqr_sift.cc:41-50)
// so for a fixed query, ranking by  score = 2*dot(a',b') + (256*sum(a) - ||a||^2)  is ranking
// by -dist, exactly, in int32.  aux[row] = 256*sum(a) - ||a||^2 is precomputed.  Rows of a
// split arrive in increasing position and the compare is strict, so ties keep the lowest
// positions: the survivors are exactly the (dist, position)-smallest k.
// ---------------------------------------------------------------------------------------
struct BfArgsU8 {
    const uint8_t* base_i8;  // [n_pad][128] re-centred copy (x ^ 0x80), n_pad = n rounded up to 64, zero rows behind n
    const int32_t* aux;      // [n_pad], kPadAux behind n
    const uint8_t* queries;  // [qpad][128]
    u64* cand;
    int* cand_cnt;
    uint32_t* gq;          // [qpad][nsplit] counted bounds (see counted_bound)
    int xj, xm;
    int n, nqt, nsplit, rows_per_split, kprime, cap;
    int dbg;
    int tile_stride;          // 1: every 64-row tile; S: only tiles 0, S, 2S, ... (sample pass of the fast path)
    const int* tile_fail;     // fallback launch of the fast path: query tiles whose flag is 0 have nothing to do
    int fail_group;           // ... flags are kept per group of this many 128-query tiles
};

// Same selection machinery as the f32 kernel (block-max prefilter, pending keys, per-lane top-8,
// counted bound between splits, wave-cooperative compaction), integer scores.
//
// Staging: the i8 MFMAs of a 64-row stage take only ~256 cycles per wave, far less than one
// global-load latency, so rows are streamed by LDS-DMA (global_load_lds_dwordx4, no VGPRs) into a
// ring of six 8 KB tiles, five stages ahead of their use; a counted vmcnt + raw s_barrier keeps four
// tiles in flight across every barrier.  The DMA writes LDS lane-linearly (1 KiB = 8 rows x 128 B per
// wave instruction), so the bank-conflict swizzle is applied on the SOURCE side: LDS chunk p of row r
// holds the row's 16-byte chunk p ^ ((r >> 1) & 7).  Rows come from a copy of the base that is already
// re-centred to int8 (x ^ 0x80) and padded to a multiple of 64 rows; aux is padded with kPadAux.
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ __launch_bounds__(256, 2) void bf_select_u8_kernel(BfArgsU8 a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.x;
    const int xcd = b & 7, rest = b >> 3;
    const int qt = rest % a.nqt;
    const int split = (rest / a.nqt) * 8 + xcd;
    if (split >= a.nsplit) return;
    if (a.tile_fail && a.tile_fail[qt / a.fail_group] == 0) return;

    constexpr int kRing = 6, kAuxRing = 8, kTileBytes = BF_BN * 128;
    constexpr int kThr0 = -(1 << 29);     // below every real score (|score| < 2^25); pad rows score ~ -2^30
    char* ring = smem;                                                      // [kRing][BN][128] swizzled
    int* auxr = reinterpret_cast<int*>(ring + kRing * kTileBytes);          // [kAuxRing][BN]
    u64* scratch = reinterpret_cast<u64*>(auxr + kAuxRing * BF_BN) + (size_t)wave * a.cap;

    const int qidx = qt * BF_TQ + wave * 32 + l31;
    const int half = a.cap >> 1;
    u64* candq = a.cand + ((size_t)qidx * a.nsplit + split) * a.cap + (size_t)h * half;
    int mycnt = 0;
    u64* wave_cand = a.cand + ((size_t)(qt * BF_TQ + wave * 32) * a.nsplit + split) * a.cap;
    const size_t qstride = (size_t)a.nsplit * a.cap;

    // stage j of this split = sample tile (split * tps + j) = rows [(split * tps + j) * tile_stride * 64, +64)
    const int tstr = a.tile_stride;
    const int tps = a.rows_per_split / BF_BN;
    const int tiles_all = (a.n + BF_BN - 1) / BF_BN;
    const int stiles_all = (tiles_all + tstr - 1) / tstr;
    const int nstages = max(0, min(tps, stiles_all - split * tps));
    const int r_begin = split * tps * tstr * BF_BN;
    const int stage_rows = tstr * BF_BN;  // row distance between consecutive stages

    // one tile = 8 DMA pieces of 1 KiB (8 rows) + 64 aux words: each wave issues 2 pieces + 16 aux words
    const int dma_row = lane >> 3;                    // row inside a piece
    auto issue_tile = [&](int stage) __attribute__((always_inline)) {
        const int slot = stage % kRing;
        const int row0 = r_begin + stage * stage_rows;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * wave + jj;
            const int row = 8 * j + dma_row;
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            const uint8_t* src = a.base_i8 + (size_t)(row0 + row) * 128 + c * 16;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ring + slot * kTileBytes + j * 1024), 16, 0, 0);
        }
        if (lane < 16)
            __builtin_amdgcn_global_load_lds((gptr_t)(a.aux + row0 + 16 * wave + lane),
                                             (lptr_t)(auxr + (stage % kAuxRing) * BF_BN + 16 * wave), 4, 0, 0);
    };

    // query fragments: 4 K-steps of 32 bytes; lane (l31,h) holds bytes 32*ks + 16*h + {0..15}
    i32x4 bq[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        i32x4 v = *reinterpret_cast<const i32x4*>(a.queries + (size_t)qidx * 128 + 32 * ks + 16 * h);
        bq[ks] = v ^ (int)0x80808080;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): retire the query loads before any DMA is counted

    const bool use_top8 = a.kprime <= 16;
    const bool track8 = use_top8 || a.xj > 0;
    int t8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t8[i] = INT32_MIN;
    int thr = kThr0;  // pass <=> score > thr

    // Appended keys are stored straight away: with LDS-DMA staging a store younger than the tile requests only
    // makes the counted vmcnt wait at the end of the stage more conservative (DMA older than two stages has
    // landed anyway), so the register-pending scheme of the f32 kernel buys nothing here.
    auto consider = [&](int s, int pos) __attribute__((always_inline)) {
        if (s > thr) {
            const u64 key = make_sel_key(i32_ord(s), (uint32_t)pos);
            if (mycnt < half) candq[mycnt] = key;
            mycnt++;
            if (track8 && s > t8[7]) {
                int v = s;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int hi = max(t8[i], v);
                    v = min(t8[i], v);
                    t8[i] = hi;
                }
            }
        }
    };
    auto finish_stage = [&](bool lastst) __attribute__((always_inline)) {
        const bool need = lastst || (mycnt > half - BF_BN / 2);
        if (lastst) {
            // final pass, every lane on its own half-buffer: drop what the final threshold rules out
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const uint32_t t_ord = i32_ord(thr);
            const int n = mycnt < half ? mycnt : half;
            int w = 0;
            for (int i = 0; i < n; ++i) {
                const u64 k0 = candq[i];
                if ((uint32_t)(k0 >> 32) >= t_ord) candq[w++] = k0;  // ties stay: the compaction decides
            }
            mycnt = w;
        }
        const uint32_t t_ord = compact_queries(wave_cand, qstride, a.cap, a.kprime, scratch, need, mycnt, lane);
        if (t_ord != 0u) thr = max(thr, ord_i32(t_ord));
    };
    auto check_block = [&](const i32x16& acc, int blk, int stage) __attribute__((always_inline)) {
        const int* ax = auxr + (stage % kAuxRing) * BF_BN + blk * 32 + 4 * h;
        const int row0 = r_begin + stage * stage_rows + blk * 32;
        int sc[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // rows 8g+4h+{0..3} of this block: one 16-byte read of their aux values
            const i32x4 av = *reinterpret_cast<const i32x4*>(ax + 8 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) sc[4 * g + j] = 2 * acc[4 * g + j] + av[j];
        }
        int m4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) m4[g] = max(max(sc[4 * g], sc[4 * g + 1]), max(sc[4 * g + 2], sc[4 * g + 3]));
        const int mx = max(max(m4[0], m4[1]), max(m4[2], m4[3]));
        if (__any(mx > thr) && !(a.dbg & 32)) {
            // hits are sparse: look only into the 4-row groups that hold one
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (__any(m4[g] > thr)) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) consider(sc[4 * g + j], row0 + acc_row(4 * g + j, h));
                }
            }
            // 16 rows of this split score >= min of the two lanes' 8th best; later rows tie-break behind them
            if (use_top8) thr = max(thr, min(t8[7], __shfl_xor(t8[7], 32, 64)));
            if (__any(mycnt > half - BF_BN / 2)) finish_stage(false);
        }
    };
    auto exchange_counted = [&]() __attribute__((always_inline)) {
        int mine = t8[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) mine = (i == a.xj - 1) ? t8[i] : mine;
        const int pv = min(mine, __shfl_xor(mine, 32, 64));
        const uint32_t pu = i32_ord(pv);  // INT32_MIN -> 0 = nothing to report yet
        const uint32_t t = counted_bound(a.gq + (size_t)qidx * a.nsplit, a.nsplit, split, h, pu, a.xm);
        // rows of OTHER splits that tie with the bound may precede ours: keep score >= bound
        if (t > 1u) thr = max(thr, ord_i32(t) - 1);
    };

    for (int t = 0; t < kRing - 1 && t < nstages; ++t) issue_tile(t);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    // Software pipeline over the stages: the fragment reads of tile t are issued first, their LDS latency
    // is covered by the selection epilogue of tile t-1 (whose scores sit in acc0/acc1), then the MFMAs of
    // tile t overwrite the accumulators; their latency is covered by the barrier and the next reads.
    const int sw = (l31 >> 1) & 7;  // fragment reads undo the source-side swizzle
    i32x16 acc0, acc1;
    for (int t = 0; t <= nstages; ++t) {
        const bool compute = t < nstages;
        i32x4 fa[4], fb[4];
        if (compute) {
            const char* tp = ring + (t % kRing) * kTileBytes + l31 * 128;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = ((2 * ks + h) ^ sw) * 16;
                fa[ks] = *reinterpret_cast<const i32x4*>(tp + off);
                fb[ks] = *reinterpret_cast<const i32x4*>(tp + 32 * 128 + off);
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the reads above the epilogue
        if (t > 0 && !(a.dbg & 1)) {
            check_block(acc0, 0, t - 1);
            check_block(acc1, 1, t - 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!compute) break;
        const bool more = t + kRing - 1 < nstages;
        // the slot of tile t+5 held tile t-1: every wave finished reading its rows before the last barrier
        // (its aux words live in a deeper ring: other waves may still be selecting on tile t-1)
        if (more && !(a.dbg & 2)) issue_tile(t + kRing - 1);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc0[i] = 0;
            acc1[i] = 0;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[ks], bq[ks], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[ks], bq[ks], acc1, 0, 0, 0);
        }
        if (a.dbg & 1) asm volatile("" ::"v"(acc0[0]), "v"(acc1[0]), "v"(acc0[15]), "v"(acc1[15]));
        if (a.xj > 0 && !(a.dbg & 256) && t > 0 && (((t - 1) & t) == 0 || ((t - 1) & 31) == 31)) exchange_counted();
        // tile t+1 must have landed; tiles t+2 .. t+5 (3 DMA instructions each) stay in flight.
        // Younger stores/loads of the epilogue only make this wait more conservative (in-order return).
        if (a.dbg & 8) continue;
        if (more) asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (nstages > 0 && !(a.dbg & 1)) {
        if (a.xj > 0 && !(a.dbg & 256)) exchange_counted();
        finish_stage(true);
    }
    if (h == 0) a.cand_cnt[(size_t)qidx * a.nsplit + split] = mycnt;
}

// ---- hit entries of the streaming scans ------------------------------------------------------------------------
// A scan lane that finds rows at or above its threshold in a 32-row block appends ONE word per block to its list:
// (block index inside the split) << 16 | 16-bit mask of its rows -- a dozen instructions.  The workgroup barrier at the
// end of every stage makes all eight waves wait for the slowest, and with ~1 block in 10 holding a hit some wave is on
// this path in most stages: decoding the mask into row positions there (a divergent loop, or 16 x if-chains) cost
// 0.09 ms of 0.72 ms at C2.  The re-rank kernels expand the entries (scan_gather_entries).
// mask bit (15 - i) <=> accumulator register i.  Built without the scalar registers: a compare writes VCC and the
// instruction that consumes it (v_addc / v_cndmask) waits ~30 clocks for it -- 16 such pairs cost 0.1 ms of 0.65 at C2;
// the sign of (score - threshold) shifted in with v_alignbit is two plain VALU instructions per score.
// (score >= t  <=>  score - t is not negative, for all finite values and for infinite scores against finite thresholds;
//  inf - inf = NaN may add a spurious candidate, never lose one.)
__device__ __forceinline__ uint32_t hit_mask_f32(const f32x16& c, float t) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    uint32_t m = 0u;
    const f32x2 tt = {t, t};
    // (one v_pk_add_f32 for two scores -- the accumulators are register pairs --, four of them ahead of their eight
    //  v_alignbit: a packed result read by the very next instruction costs a wait state)
#pragma unroll
    for (int i0 = 0; i0 < 16; i0 += 8) {
        f32x2 d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = f32x2{c[i0 + 2 * j], c[i0 + 2 * j + 1]} - tt;
        asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            m = __builtin_amdgcn_alignbit(m, __float_as_uint(d[j][0]), 31);   // (m << 1) | sign
            m = __builtin_amdgcn_alignbit(m, __float_as_uint(d[j][1]), 31);
        }
    }
    return ~m & 0xffffu;
}
__device__ __forceinline__ uint32_t hit_mask_i32(const i32x16& c, int t) {
    uint32_t m = 0u;
#pragma unroll
    for (int i = 0; i < 16; ++i) m = __builtin_amdgcn_alignbit(m, (uint32_t)(c[i] - t), 31);   // scores and thresholds < 2^30
    return ~m & 0xffffu;
}
// list s = (split, half) of query q: entries -> row positions at keys[offs[s] ..]; one thread per list (a handful of
// entries each).  offs[] holds the prefix sums of the lists' POSITION counts (list_cnt), none above caph here.
// Entry i of the query's lists is one contiguous plane [list] (round 3; a list per cache line before: the re-rank
// fetched 2 * nsplit lines per query for one or two entries each -- 33 MB at C2, 12 us): the threads of a wave read
// neighbouring words, and the first four planes are requested together (an entry lists at least one row, so entry
// i < len exists or is never looked at).
__device__ __forceinline__ void scan_gather_entries(u64* keys, const int* offs, const uint32_t* list, int q, int nl, int caph,
                                                    int tps, int tid, int nthreads) {
    for (int s = tid; s < nl; s += nthreads) {
        const int len = offs[s + 1] - offs[s];
        if (len == 0) continue;
        const uint32_t* e = list + (size_t)q * caph * nl + s;
        const uint32_t row_base = (uint32_t)(s >> 1) * (uint32_t)tps * BF_BN + 4u * (s & 1);
        uint32_t pre[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pre[i] = (i < len && i < caph) ? e[(size_t)i * nl] : 0u;
        int j = 0;
        auto expand = [&](const uint32_t ent) __attribute__((always_inline)) {
            uint32_t m = ent & 0xffffu;
            const uint32_t r0 = row_base + (ent >> 16) * 32u;
            while (m && j < len) {
                const int bit = 31 - __builtin_clz(m);   // highest bit first = lowest register = lowest row
                m &= ~(1u << bit);
                const int reg = 15 - bit;
                keys[offs[s] + j++] = (u64)(r0 + (reg & 3) + 8 * (reg >> 2));
            }
        };
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < caph && j < len) expand(pre[i]);
        for (int i = 4; i < caph && j < len; ++i) expand(e[(size_t)i * nl]);
    }
}

// ---------------------------------------------------------------------------------------
// uint8 fast path (large batches): the selection threshold of every query is FIXED before the rows are streamed.
//   1. sample pass   - bf_select_u8_kernel over every 8th 64-row tile with k' = r: the r-th best score of the sample
//                      (bf_u8_threshold_kernel) is, with overwhelming probability, below the k-th best score of the
//                      whole base while only ~r*8 rows reach it;
//   2. scan          - bf_scan_u8_kernel: 128*QG queries per workgroup (QG x 32 per wave: the L2 -> LDS stream of the
//                      rows, which bounds the adaptive kernel at 128 queries per workgroup, shrinks QG-fold), the
//                      accumulators start from aux >> 1 (read from LDS straight into the MFMA registers) so that
//                      acc = (score - (aux & 1)) / 2 needs no arithmetic; epilogue = one max tree + one compare per
//                      32 x 64 block; rows that reach the threshold append their POSITION to the lane's own list;
//   3. re-rank       - bf_rerank_u8_list_kernel: exact integer distances of the listed rows, (distance, position)
//                      order, top k.  It also VERIFIES the bet: a query whose lists overflowed or hold fewer than k
//                      rows flags its query-tile group;
//   4. fallback      - flagged groups are redone by the adaptive kernel + bf_rerank_kernel (launched always; their
//                      workgroups leave at once when the flag is clear).  The result is exact either way.
// ---------------------------------------------------------------------------------------
struct BfScanArgs {
    const uint8_t* base_i8;   // [n_pad][128]
    const int32_t* auxh;      // [n_pad] aux >> 1 (pad rows: -2^29)
    const uint8_t* queries;   // [qpad][128]
    const int* thr;           // [qpad] pass <=> dot' + auxh >= thr
    uint32_t* list;           // [qpad][caph][nsplit][2] hit entries: block << 16 | row mask (see hit_mask_i32)
    int* list_cnt;            // [qpad][nsplit][2] rows listed
    int n, nqt, nsplit, tps, caph;
    int tile_stride;          // SAMPLE: every tile_stride-th tile
    int* top8;                // SAMPLE: [qpad][nsplit][2][8] best (score >> 1) values each lane saw, descending
};

// SAMPLE = the sample pass: no thresholds yet; every lane keeps the 8 best values it sees (sorted insertion network in
// registers, entered only when a value beats the lane's 8th best) and writes them out at the end.
template <int QG, bool SAMPLE>
__global__ __launch_bounds__(256, 2) void bf_scan_u8_kernel(BfScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.x;
    const int xcd = b & 7, rest = b >> 3;
    const int qt = rest % a.nqt;
    const int split = (rest / a.nqt) * 8 + xcd;
    if (split >= a.nsplit) return;

    constexpr int kGroup = 4, kRing = 2 * kGroup, kAuxRing = 8, kTileBytes = BF_BN * 128;
    char* ring = smem;
    int* auxr = reinterpret_cast<int*>(ring + kRing * kTileBytes);

    const int tstr = SAMPLE ? a.tile_stride : 1;
    const int tiles_all = (a.n + BF_BN - 1) / BF_BN;
    const int stiles_all = (tiles_all + tstr - 1) / tstr;
    const int nstages = max(0, min(a.tps, stiles_all - split * a.tps));
    const int r_begin = split * a.tps * tstr * BF_BN;
    const int stage_rows = tstr * BF_BN;

    // LDS-DMA in the scalar-base form (see bf_scan_bf16_kernel's issue_tile: the lane's constant byte offset in a VGPR, the
    // stage's base in an SGPR pair, M0 from scalars; every wave requests all 64 aux values, one instruction, no exec mask)
    const int dma_row = lane >> 3;
    uint32_t dma_voff[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int row = 8 * (2 * wave + jj) + dma_row;
        dma_voff[jj] = (uint32_t)(row * 128 + ((lane & 7) ^ ((row >> 1) & 7)) * 16);
    }
    const uint32_t dma_lds0 = (uint32_t)(uintptr_t)(lptr_t)ring + (uint32_t)(2 * __builtin_amdgcn_readfirstlane(wave) * 1024);
    const uint32_t aux_lds0 = (uint32_t)(uintptr_t)(lptr_t)auxr;
    const uint32_t aux_voff = (uint32_t)lane * 4u;
    auto issue_tile = [&](int stage) __attribute__((always_inline)) {
        const int slot = stage % kRing;
        const int row0 = r_begin + stage * stage_rows;
        const unsigned long long sbase = (unsigned long long)(uintptr_t)a.base_i8 + (unsigned long long)row0 * 128ull;
        const uint32_t m0v = dma_lds0 + (uint32_t)slot * kTileBytes;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                         :: "v"(dma_voff[jj]), "s"(sbase), "s"(m0v + (uint32_t)jj * 1024u) : "memory", "m0");
        const unsigned long long abase = (unsigned long long)(uintptr_t)a.auxh + (unsigned long long)row0 * 4ull;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1"
                     :: "v"(aux_voff), "s"(abase), "s"(aux_lds0 + (uint32_t)((stage % kAuxRing) * BF_BN * 4)) : "memory", "m0");
#pragma clang diagnostic pop
    };

    // this lane's QG queries, their thresholds, their lists (each lane owns the rows of its half h: no atomics)
    i32x4 bq[QG][4];
    int thr[QG], cnt[QG], ecnt[QG];   // cnt: rows listed, ecnt: list words written (one per block with a hit)
    uint32_t* lp[QG];
    int t8[SAMPLE ? QG : 1][8];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const int qidx = (qt * 4 + wave) * (32 * QG) + g * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            i32x4 v = *reinterpret_cast<const i32x4*>(a.queries + (size_t)qidx * 128 + 32 * ks + 16 * h);
            bq[g][ks] = v ^ (int)0x80808080;
        }
        cnt[g] = 0;
        ecnt[g] = 0;
        if constexpr (SAMPLE) {
            thr[g] = -(1 << 28);  // above the pad rows' -2^29: they never enter the lists
            lp[g] = nullptr;
#pragma unroll
            for (int i = 0; i < 8; ++i) t8[g][i] = -(1 << 28);
        } else {
            thr[g] = a.thr[qidx];
            lp[g] = a.list + (size_t)qidx * a.caph * (2 * a.nsplit) + split * 2 + h;   // entry i of a list: [q][i][list]
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): retire the query loads before any DMA is counted

    // Tiles are consumed in GROUPS of four (256 rows): the waves meet once per group -- a wave that met a hit arrives late,
    // the fewer meetings the less of that lateness everyone waits for -- with the group's tiles landed (requested one
    // group earlier) and, behind the barrier, the previous group's slots free for the next requests.  Ring = 2 groups.
    for (int t = 0; t < kGroup && t < nstages; ++t) issue_tile(t);

    const int sw = (l31 >> 1) & 7;
    // rows of one finished 32 x 64 block pair that reach the threshold: position appended to the lane's list
    auto examine = [&](const i32x16& c0, const i32x16& c1, int g, int row0) __attribute__((always_inline)) {
        int m0 = max(max(c0[0], c0[1]), c0[2]);
#pragma unroll
        for (int i = 3; i + 1 < 16; i += 2) m0 = max(max(m0, c0[i]), c0[i + 1]);
        m0 = max(m0, c0[15]);
        int m1 = max(max(c1[0], c1[1]), c1[2]);
#pragma unroll
        for (int i = 3; i + 1 < 16; i += 2) m1 = max(max(m1, c1[i]), c1[i + 1]);
        m1 = max(m1, c1[15]);
        const int tg = thr[g];
        if constexpr (SAMPLE) {
            // only the best value of each 32-row block enters the lane's top-8: any threshold is admissible (the
            // re-rank verifies the outcome), a second value of the same block among a lane's eight best is rare, and
            // examining all 32 values whenever ANY of the 64 lanes has a hit would cost 16x the instructions
            if (__builtin_expect(__any(max(m0, m1) > tg), 0)) {
                auto ins = [&](int v) __attribute__((always_inline)) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int hi = max(t8[g][i], v);
                        v = min(t8[g][i], v);
                        t8[g][i] = hi;
                    }
                };
                ins(m0);
                ins(m1);
                thr[g] = t8[g][7];
            }
        } else {
            // one entry per 32-row block with a hit (see hit_mask_i32): the re-rank expands them.  Each block on its own
            // trigger: at k = 100 a check meets a hit 4 times in 10, and a mask costs 32 VALU instructions
            const uint32_t blk = (uint32_t)(row0 - r_begin) >> 5;
            if (__builtin_expect(__any(m0 >= tg), 0)) {
                const uint32_t k0 = hit_mask_i32(c0, tg);
                if (k0) {
                    if (ecnt[g] < a.caph) lp[g][(size_t)ecnt[g] * (2 * a.nsplit)] = (blk << 16) | k0;
                    ecnt[g]++;
                    cnt[g] += __builtin_popcount(k0);
                }
            }
            if (__builtin_expect(__any(m1 >= tg), 0)) {
                const uint32_t k1 = hit_mask_i32(c1, tg);
                if (k1) {
                    if (ecnt[g] < a.caph) lp[g][(size_t)ecnt[g] * (2 * a.nsplit)] = ((blk + 1) << 16) | k1;
                    ecnt[g]++;
                    cnt[g] += __builtin_popcount(k1);
                }
            }
        }
    };

    auto process_tile = [&](int t) __attribute__((always_inline)) {
        const char* tp = ring + (t % kRing) * kTileBytes + l31 * 128;
        const int* ax = auxr + (t % kAuxRing) * BF_BN + 4 * h;
        const int row0 = r_begin + t * stage_rows;
        i32x4 fa[4], fb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int off = ((2 * ks + h) ^ sw) * 16;
            fa[ks] = *reinterpret_cast<const i32x4*>(tp + off);
            fb[ks] = *reinterpret_cast<const i32x4*>(tp + 32 * 128 + off);
        }
        // accumulators start from aux >> 1 of their rows: acc register 4j+i of half h = row 8j + 4h + i of the block
        i32x16 init0, init1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const i32x4 v0 = *reinterpret_cast<const i32x4*>(ax + 8 * j);
            const i32x4 v1 = *reinterpret_cast<const i32x4*>(ax + 32 + 8 * j);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                init0[4 * j + i] = v0[i];
                init1[4 * j + i] = v1[i];
            }
        }
        i32x16 p0, p1;  // scores of the previous query group, examined under the MFMAs of the current one
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            i32x16 c0 = init0, c1 = init1;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[ks], bq[g][ks], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[ks], bq[g][ks], c1, 0, 0, 0);
            }
            if (g > 0) examine(p0, p1, g - 1, row0);
            p0 = c0;
            p1 = c1;
        }
        examine(p0, p1, QG - 1, row0);
    };
    for (int t = 0; t < nstages; t += kGroup) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        for (int u = t + kGroup; u < t + 2 * kGroup && u < nstages; ++u) issue_tile(u);
#pragma unroll
        for (int u = 0; u < kGroup; ++u)
            if (t + u < nstages) process_tile(t + u);
    }
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const int qidx = (qt * 4 + wave) * (32 * QG) + g * 32 + l31;
        if constexpr (SAMPLE) {
            int* o = a.top8 + (((size_t)qidx * a.nsplit + split) * 2 + h) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = t8[g][i];
        } else {
            a.list_cnt[((size_t)qidx * a.nsplit + split) * 2 + h] = cnt[g];
        }
    }
}

// r-th largest of the 64 * NPER ordered keys a wave holds (key[i] of every lane; 0 = "no value", below every real key):
// bit by bit from the top, the largest T that at least r keys reach.  The counts are scalar (ballot + popcount): no LDS,
// no barrier, no sort -- a threshold needs ONE order statistic of the sample, not its order.  (Round 3; the workgroup-wide
// bitonic sort of the 512 / 1024 keys took 18 us at C2 and 40 us at C4.)  Fewer than r real keys: 0.
template <int NPER>
__device__ __forceinline__ uint32_t wave_rth_largest_u32(const uint32_t (&key)[NPER], int r) {
    uint32_t T = 0;
    for (int b = 31; b >= 0; --b) {
        const uint32_t cand = T | (1u << b);
        int c = 0;
#pragma unroll
        for (int i = 0; i < NPER; ++i) c += __popcll(__builtin_amdgcn_ballot_w64(key[i] >= cand));
        if (c >= r) T = cand;
    }
    return T;
}

// r-th best value of the sample (union of the lanes' top-8 lists) -> scan threshold of each query; one wave per query
// (four queries per workgroup), the sample's values in registers.
// Any value works as a threshold -- the re-rank verifies the outcome -- this one makes ~r * stride rows pass.
template <int NPER>
__global__ __launch_bounds__(256) void bf_u8_threshold_kernel(const int* top8, int nlists, int r, int nq, int qpad, int* thr) {
    const int lane = threadIdx.x & 63, q = blockIdx.x * 4 + (threadIdx.x >> 6);
    constexpr int kPassAll = -(1 << 28);  // below every real value, above the pad rows' -2^29
    if (q >= qpad) return;
    if (q >= nq) {                        // padding queries: nothing may pass
        if (lane == 0) thr[q] = 0x7FFFFFFF;
        return;
    }
    const int total = nlists * 8;
    uint32_t key[NPER];
#pragma unroll
    for (int i = 0; i < NPER; ++i) {
        const int idx = i * 64 + lane;
        key[i] = idx < total ? i32_ord(top8[(size_t)q * total + idx]) : 0u;
    }
    int t = kPassAll;
    if (total >= r) t = max(kPassAll, ord_i32(wave_rth_largest_u32<NPER>(key, r)));
    if (lane == 0) thr[q] = t;
}

// exact distances of the listed rows, canonical (distance, position) order, top k; verification of the threshold bet
struct RerankListArgs {
    const uint8_t* base;      // original bytes [n][128]
    const uint8_t* queries;   // [qpad][128]
    const uint32_t* list;
    const int* list_cnt;
    const int32_t* ext_ids;
    int32_t* out_ids;
    float* out_dists;
    int32_t* out_cnt;
    int* tile_fail;           // [ceil(nq / fail_queries)]
    int n, k, nsplit, caph, p2max, fail_queries;
    int tps;                  // tiles per split of the scan (entries hold block indices inside their split)
};

__global__ __launch_bounds__(256) void bf_rerank_u8_list_kernel(RerankListArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);              // [p2max]
    int* offs = reinterpret_cast<int*>(keys + a.p2max);    // [2 * nsplit + 1]
    __shared__ int s_over;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = 2 * a.nsplit;
    // exclusive prefix sum of the (clipped) list lengths: wave 0, each lane a run of consecutive lists
    if (tid == 0) s_over = 0;
    __syncthreads();
    if (tid < 64) {
        const int per = (nl + 63) / 64;
        int sum = 0, over = 0;
        for (int i = 0; i < per; ++i) {
            const int s = tid * per + i;
            if (s < nl) {
                const int c = a.list_cnt[(size_t)q * nl + s];
                over |= c > a.caph;
                sum += c < a.caph ? c : a.caph;
            }
        }
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (tid >= o) incl += v;
        }
        int run = incl - sum;
        for (int i = 0; i < per; ++i) {
            const int s = tid * per + i;
            if (s < nl) {
                offs[s] = run;
                const int c = a.list_cnt[(size_t)q * nl + s];
                run += c < a.caph ? c : a.caph;
            }
        }
        if (tid == 63) offs[nl] = incl;
        if (__any(over != 0) && tid == 0) s_over = 1;
    }
    __syncthreads();
    const int total = offs[nl];
    const int need = a.k < a.n ? a.k : a.n;
    if (s_over || total < need || total > a.p2max) {
        // lists overflowed (ties, or an unlucky sample), or fewer than k rows reached the threshold: the adaptive
        // kernel redoes this query's tile group
        if (tid == 0) atomicOr(&a.tile_fail[q / a.fail_queries], 1);
        return;
    }
    scan_gather_entries(keys, offs, a.list, q, nl, a.caph, a.tps, tid, blockDim.x);
    __syncthreads();
    const int P = next_pow2(total < 2 ? 2 : total);
    // sum (a-b)^2 = a.a + b.b - 2 a.b on packed bytes (exact in int32; distcomp_l2sqr_sift.cc:41-50 gives the same
    // integer).  8 lanes x 16 bytes per row, 8 rows per wave and pass, the row loads of a pass issued together.
    const uint8_t* qq = a.queries + (size_t)q * 128;
    const int sub = lane & 7, g8 = lane >> 3;
    const i32x4 qv = *reinterpret_cast<const i32x4*>(qq + sub * 16);
    int qsq = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) qsq = __builtin_amdgcn_udot4(qv[j], qv[j], qsq, false);
    // (four passes of 8 rows per wave requested together: 128 rows of the query in flight per workgroup round)
    for (int j0 = wave * 32; j0 < total; j0 += 128) {
        uint32_t pos[4];
        i32x4 bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 8 * u + g8;
            pos[u] = (uint32_t)keys[j < total ? j : total - 1];
            bv[u] = *reinterpret_cast<const i32x4*>(a.base + (size_t)pos[u] * 128 + sub * 16);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int bsq = 0, dot = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bsq = __builtin_amdgcn_udot4(bv[u][t], bv[u][t], bsq, false);
                dot = __builtin_amdgcn_udot4(bv[u][t], qv[t], dot, false);
            }
            int d = qsq + bsq - 2 * dot;
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            d += __shfl_xor(d, 4, 64);
            const int j = j0 + 8 * u + g8;
            if (sub == 0 && j < total) keys[j] = ((u64)i32_ord(d) << 32) | pos[u];
        }
    }
    for (int i = total + tid; i < P; i += blockDim.x) keys[i] = ~0ull;
    __syncthreads();
    block_bitonic_u64_asc(keys, P, tid, blockDim.x);
    const int found = total < a.k ? total : a.k;
    for (int i = tid; i < a.k; i += blockDim.x) {
        int32_t id = -1;
        float d = INFINITY;
        if (i < found) {
            const u64 key = keys[i];
            const uint32_t pos = (uint32_t)key;
            id = a.ext_ids ? a.ext_ids[pos] : (int32_t)pos;
            d = (float)ord_i32((uint32_t)(key >> 32));
        }
        a.out_ids[(size_t)q * a.k + i] = id;
        a.out_dists[(size_t)q * a.k + i] = d;
    }
    if (tid == 0 && a.out_cnt) a.out_cnt[q] = found;
}

// ---------------------------------------------------------------------------------------
// f32 fast path (large batches, D <= 128; l2 / negdotprod / cosinesimil / angulardist): the uint8 fast path's
// structure (sample pass -> fixed thresholds -> streaming scan -> list re-rank with verification -> adaptive
// fallback) with the SELECTION contraction on the bf16 matrix cores.  Every row and query is split into two bf16
// numbers, x = hi + lo + O(2^-16 |x|), and q.b is taken as qh.bh + qh.bl + ql.bh: three v_mfma_f32_32x32x16_bf16 per 16
// elements = 3/16 of the matrix-pipe time of the f32 MFMA, f32 accumulation, relative error ~5e-5 per product (the
// dropped ql.bl term and the split residues).  That error only has to be small against the gap between the k-th and
// the ~8r-th neighbour (the threshold lets ~70-100 rows through per query, k' = 14 for the adaptive kernel); the
// re-rank computes the reference formula in f32 on the original rows, and the verification + fallback make the result
// independent of the approximation: exact like the adaptive path.
// Workgroup = 8 waves = 256 queries (32 per wave, both bf16 halves of the query fragments resident: 64 VGPRs); rows
// stream by LDS-DMA into a ring of 32 KB stages (64 rows x 128 x {hi, lo}), 16-byte chunks swizzled on the source side
// (chunk ^ (row & 15)) so that ds_read_b128 of 16 rows is conflict-free; the L2 norm term enters as the accumulators'
// initial value, read from LDS straight into the MFMA registers.
// ---------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

enum ScanMode : int { SC_L2 = 0, SC_DOT = 1, SC_COS = 2 };

struct BfScanF32Args {
    const __bf16* base_hi;    // [n_pad][128]
    const __bf16* base_lo;    // [n_pad][128]
    const float* auxp;        // [n_pad]: l2 -0.5|b|^2, cosine 1/|b|, dot 0; rows >= n: -inf (l2) / 0
    const __bf16* q_hi;       // [qpad][128]
    const __bf16* q_lo;
    // one-product scan (round 3): fp16 tiles of the rows and queries times a power of two `scale` (3 more significant bits
    // than bf16: an eighth of the rounding error, so the one-product bound E1 leaves room on far more data and the threshold
    // sits closer to the k'-th score); scores, start values and thresholds of that scan are in units of scale_rows * scale_queries
    const void* base_h16;     // [n_pad][dp] _Float16
    const void* q_h16;        // [qpad][dp] _Float16
    const float* auxp16;      // [n_pad] start values of the fp16 scan (l2: scale^2 * aux; cosine: aux)
    int prio_half;            // one-product scan, eight waves: waves 4..7 run at priority 1
    const float* thr;         // [qpad] pass <=> score >= thr
    uint32_t* list;           // [qpad][caph][nsplit][2] hit entries: block << 16 | row mask (see hit_mask_f32)
    int* list_cnt;            // [qpad][nsplit][2] rows listed
    int n, nqt, nsplit, tps, caph;
    int tile_stride;          // SAMPLE
    float* top8;              // SAMPLE: [qpad][nsplit][2][8]
    const int* group_flag;    // [nqt] or null: a workgroup runs only if group_flag[its query tile] == group_want
    int group_want;
};

// Workgroup = 4 waves, ONE per SIMD, each with the SIMD's whole 512-entry register file; a wave serves QG groups of 32
// queries (QG = 4: 512 queries per workgroup, 256 registers of query fragments; QG = 2: 256 queries), so that every
// fragment read from LDS feeds 3 * QG MFMAs.  (The first shape of this kernel -- 8 waves x 32 queries, two waves per
// SIMD -- read 20 ds_read_b128 per wave and block: the LDS port was busy 1280 of the 1536 clocks the block's MFMAs
// take, and two query groups per wave spilled at 256 registers.)
// (SAMPLE = true, a sample pass with near-exact scores, is not instantiated any more: the sample pass runs
//  bf_scan_bf16_kernel<MODE, true, 2> and the threshold kernel accounts for its error.)
// Fragment registers of the scan kernels are PINNED to physical VGPRs.  A fragment is the target of an asynchronous
// ds_read (inline asm) and becomes valid at a counted s_waitcnt (inline asm naming the same register, so that the MFMAs
// behind it cannot move above it).  With ordinary "=v"/"+v" operands the register allocator may give the wait's
// operand another register than the read's and insert the copy IN FRONT of the wait -- a copy of bytes that have not
// arrived (seen in bf_scan_bf16_kernel<.., 1, 8>'s last block: rows listed at random, a neighbour lost in ~8 % of the
// batches at 70k rows x 600 queries).  One fixed register tuple per slot leaves nothing to copy.
#define BF_FRAG_RD(REGS, var, addr) asm volatile("ds_read_b128 %0, %1" : "={" REGS "}"(var) : "v"(addr) : "memory")
#define BF_FRAG_RD_OFF(REGS, var, addr, off) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "={" REGS "}"(var) : "v"(addr), "n"(off) : "memory")

template <int MODE, bool SAMPLE, int QG>
__global__ __launch_bounds__(256) void bf_scan_f32_kernel(BfScanF32Args a) {
    constexpr int NW = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.x;
    const int xcd = b & 7, rest = b >> 3;
    const int qt = rest % a.nqt;
    const int split = (rest / a.nqt) * 8 + xcd;
    if (split >= a.nsplit) return;
    if (a.group_flag && (a.group_flag[qt] != 0) != (a.group_want != 0)) return;   // the other scan kernel serves this query tile

    constexpr int kRing = 4, kAuxRing = 8, kHalfBytes = BF_BN * 256, kStageBytes = 2 * kHalfBytes;  // hi tile | lo tile
    char* ring = smem;
    float* auxr = reinterpret_cast<float*>(ring + kRing * kStageBytes);  // [kAuxRing][BN]

    const int tstr = SAMPLE ? a.tile_stride : 1;
    const int tiles_all = (a.n + BF_BN - 1) / BF_BN;
    const int stiles_all = (tiles_all + tstr - 1) / tstr;
    const int nstages = max(0, min(a.tps, stiles_all - split * a.tps));
    const int r_begin = split * a.tps * tstr * BF_BN;
    const int stage_rows = tstr * BF_BN;

    // one stage = 32 DMA pieces of 1 KiB (4 rows x 256 B; 0..15 hi tile, 16..31 lo tile): wave w issues pieces
    // kPieces * w .. + kPieces - 1, and its share of the stage's 64 aux values
    // (scalar-base LDS-DMA, see bf_scan_bf16_kernel's issue_tile; NW = 4: wave w issues pieces 8w .. 8w+7, all of one
    //  half -- hi tile for waves 0 and 1, lo tile for 2 and 3)
    constexpr int kPieces = 32 / NW;
    static_assert(kPieces == 8, "a wave's pieces lie in one half of the stage");
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    uint32_t dma_voff[kPieces];
#pragma unroll
    for (int jj = 0; jj < kPieces; ++jj) {
        const int pj = (kPieces * wave + jj) & 15;     // piece pj of its tile = rows 4*pj .. 4*pj+3
        const int row = 4 * pj + (lane >> 4);
        dma_voff[jj] = (uint32_t)((row * 128 + ((lane & 15) ^ (row & 15)) * 8) * 2);
    }
    const int dma_half = (kPieces * wave_u) >> 4;
    const uint32_t dma_lds0 = (uint32_t)(uintptr_t)(lptr_t)ring + (uint32_t)(dma_half * kHalfBytes + ((kPieces * wave_u) & 15) * 1024);
    const unsigned long long dma_src0 = (unsigned long long)(uintptr_t)(dma_half ? a.base_lo : a.base_hi);
    const uint32_t aux_lds0 = (uint32_t)(uintptr_t)(lptr_t)auxr;
    const uint32_t aux_voff = (uint32_t)lane * 4u;
    auto issue_tile = [&](int stage) __attribute__((always_inline)) {
        const int slot = stage % kRing;
        const int row0 = r_begin + stage * stage_rows;
        const unsigned long long sbase = dma_src0 + (unsigned long long)row0 * 256ull;
        const uint32_t m0v = dma_lds0 + (uint32_t)slot * kStageBytes;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#pragma unroll
        for (int jj = 0; jj < kPieces; ++jj)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                         :: "v"(dma_voff[jj]), "s"(sbase), "s"(m0v + (uint32_t)jj * 1024u) : "memory", "m0");
        const unsigned long long abase = (unsigned long long)(uintptr_t)a.auxp + (unsigned long long)row0 * 4ull;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1"
                     :: "v"(aux_voff), "s"(abase), "s"(aux_lds0 + (uint32_t)((stage % kAuxRing) * BF_BN * 4)) : "memory", "m0");
#pragma clang diagnostic pop
    };

    // this lane's QG queries: fragments of both halves; lane (l31, h) holds elements 16*kc + 8*h + {0..7}
    bf16x8 qh[QG][8], ql[QG][8];
    float thr[QG];
    int cnt[QG], ecnt[QG];   // cnt: rows listed, ecnt: list words written (one per block with a hit)
    uint32_t* lp[QG];
    float t8[SAMPLE ? QG : 1][8];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const int qidx = (qt * NW + wave) * (32 * QG) + g * 32 + l31;
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            // straight into AGPRs (the MFMA reads its B operand from there: the 256 registers of query fragments leave
            // the VGPR file to the accumulators, and no v_accvgpr_read copies in the loop)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(qh[g][kc]) : "v"(a.q_hi + (size_t)qidx * 128 + 16 * kc + 8 * h) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(ql[g][kc]) : "v"(a.q_lo + (size_t)qidx * 128 + 16 * kc + 8 * h) : "memory");
        }
        cnt[g] = 0;
        ecnt[g] = 0;
        if constexpr (SAMPLE) {
            thr[g] = -INFINITY;
            lp[g] = nullptr;
#pragma unroll
            for (int i = 0; i < 8; ++i) t8[g][i] = -INFINITY;
        } else {
            thr[g] = a.thr[qidx];
            lp[g] = a.list + (size_t)qidx * a.caph * (2 * a.nsplit) + split * 2 + h;   // entry i of a list: [q][i][list]
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // retire the query loads before any DMA is counted

    for (int t = 0; t < kRing - 1 && t < nstages; ++t) issue_tile(t);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    // byte offset of this lane's fragment of K-step kc inside its row (row & 15 is the same for both blocks: 32 = 0 mod 16)
    int foff[8];
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) foff[kc] = l31 * 256 + (((kc * 2 + h) ^ (l31 & 15)) * 16);

    // Score check of one finished 32-row block (16 scores per lane and query group), spread over the MFMA stream of the
    // NEXT block: two elements of the running maximum per K-step (a clump of ~40 VALU instructions between two MFMAs
    // holds the matrix pipe up; two per gap ride in its shadow), then one compare; the element-wise path only on a hit.
    auto score_of = [&](const f32x16& c, int i, const float* ax) __attribute__((always_inline)) -> float {
        if constexpr (MODE == SC_COS) return c[i] * ax[(i & 3) + 8 * (i >> 2)];
        else return c[i];
    };
    auto finish_check = [&](float m, const f32x16& c, int g, const float* ax, int row0) __attribute__((always_inline)) {
        if constexpr (SAMPLE) {
            // only the block's best value enters the lane's top-8 (see bf_scan_u8_kernel); pad rows of the dot / cosine
            // modes score 0: the tile that holds them keeps its real maximum out of the estimate only if that is negative
            if (__builtin_expect(__any(m > thr[g]), 0)) {
                float v = (row0 + 32 <= a.n || MODE == SC_L2) ? m : -INFINITY;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float hi = fmaxf(t8[g][j], v);
                    v = fminf(t8[g][j], v);
                    t8[g][j] = hi;
                }
                thr[g] = t8[g][7];
            }
        } else if (__builtin_expect(__any(m >= thr[g]), 0)) {   // (cold: the common path falls through)
            // one entry for the block if this lane has a hit (see hit_mask_f32): the re-rank expands them
            uint32_t km;
            if constexpr (MODE == SC_COS) {
                f32x16 sc;
#pragma unroll
                for (int i = 0; i < 16; ++i) sc[i] = score_of(c, i, ax);
                km = hit_mask_f32(sc, thr[g]);
            } else {
                km = hit_mask_f32(c, thr[g]);
            }
            if (row0 + 32 > a.n) {   // pad rows of the dot / cosine modes score 0: never listed
                // (behind an opaque copy of the row count: hoisted out of the branch, these 60 instructions would run
                //  on every check)
                int nvalid = a.n - row0;
                asm volatile("" : "+s"(nvalid));
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (acc_row(i, h) >= nvalid) km &= ~(0x8000u >> i);
            }
            if (km) {
                if (ecnt[g] < a.caph) lp[g][(size_t)ecnt[g] * (2 * a.nsplit)] = ((uint32_t)(row0 - r_begin) >> 5 << 16) | km;
                ecnt[g]++;
                cnt[g] += __builtin_popcount(km);
            }
        }
    };

    // The 32-row blocks form ONE software pipeline across block and stage boundaries: K-step s of the stream reads the
    // fragments of K-step s+2 (the last two steps of a block fetch the first two of the next one, and the next block's
    // accumulator start values), so no block begins with an exposed LDS round trip.  The stage's counted wait + barrier
    // sits in its MIDDLE (after block 0): it proves stage t+1 landed before block 1 prefetches from it, and that every
    // wave is done with stage t-1, whose slot then takes stage t+3.
    // The LDS reads and their waits are written by hand: the compiler's own s_waitcnt placement uses lgkmcnt(0)
    // throughout this loop, i.e. every wait also waits for the reads just issued for two steps ahead.  LDS reads return
    // in order, so "at most 4 outstanding" = the reads of steps kc+1 and kc+2 may still fly while step kc's MFMAs
    // start.  Each wait names the registers it guards, so that the MFMAs that read them cannot be moved above it.
    // The MFMAs run group-major (the three products of a group back to back on its accumulator); in a block's last
    // K-step the scores of group g are checked in the shadow of group g+1's MFMAs, the last group's under the first
    // MFMAs of the next block: no second set of accumulators.
    bf16x8 fh[4], fl[4];      // fragment slots: K-step kc of any block uses slot kc % 4
    f32x16 iv;                // start values of the next block (l2: -0.5|b|^2 of its rows; else 0)
#pragma unroll
    for (int i = 0; i < 16; ++i) iv[i] = 0.f;
    f32x16 acc[QG];
    const float* pv_ax = auxr;   // aux values / first row of the block whose last group is still unchecked
    int pv_row0 = 0;
    bool have_pv = false;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    auto load_frag = [&](uint32_t rp, int kc) __attribute__((always_inline)) {
        const uint32_t ad = rp + foff[kc];
        switch (kc % 4) {   // (kc is a constant after unrolling)
            case 0: BF_FRAG_RD("v[208:211]", fh[0], ad); BF_FRAG_RD_OFF("v[224:227]", fl[0], ad, kHalfBytes); break;
            case 1: BF_FRAG_RD("v[212:215]", fh[1], ad); BF_FRAG_RD_OFF("v[228:231]", fl[1], ad, kHalfBytes); break;
            case 2: BF_FRAG_RD("v[216:219]", fh[2], ad); BF_FRAG_RD_OFF("v[232:235]", fl[2], ad, kHalfBytes); break;
            default: BF_FRAG_RD("v[220:223]", fh[3], ad); BF_FRAG_RD_OFF("v[236:239]", fl[3], ad, kHalfBytes); break;
        }
    };
    // the counted wait that makes slot kc % 4 (and, with INIT, the start values) valid
    auto wait_frag = [&](auto cnt_tag, int kc, bool with_init) __attribute__((always_inline)) {
        constexpr int N = decltype(cnt_tag)::value;
#define BF_W3(STR)                                                                                                       \
        switch (kc % 4) {                                                                                                \
            case 0:                                                                                                      \
                if (with_init) asm volatile(STR : "+{v[208:211]}"(fh[0]), "+{v[224:227]}"(fl[0]), "+{v[240:255]}"(iv));  \
                else asm volatile(STR : "+{v[208:211]}"(fh[0]), "+{v[224:227]}"(fl[0]));                                 \
                break;                                                                                                   \
            case 1: asm volatile(STR : "+{v[212:215]}"(fh[1]), "+{v[228:231]}"(fl[1])); break;                           \
            case 2: asm volatile(STR : "+{v[216:219]}"(fh[2]), "+{v[232:235]}"(fl[2])); break;                           \
            default: asm volatile(STR : "+{v[220:223]}"(fh[3]), "+{v[236:239]}"(fl[3])); break;                          \
        }
        if constexpr (N == 0) { BF_W3("s_waitcnt lgkmcnt(0)") }
        else if constexpr (N == 4) { BF_W3("s_waitcnt lgkmcnt(4)") }
        else { BF_W3("s_waitcnt lgkmcnt(8)") }
#undef BF_W3
    };
    auto load_init = [&](uint32_t ax) __attribute__((always_inline)) {
        // register 4j+i of half h = row 8j + 4h + i of the block.  The four reads fill ONE 16-register tuple (fixed
        // registers: inline asm cannot address parts of an operand), which the block's first MFMAs take as C operand
        // as it stands -- no copies
        if constexpr (MODE == SC_L2) {
            asm volatile("ds_read_b128 v[240:243], %1\n\tds_read_b128 v[244:247], %1 offset:32\n\t"
                         "ds_read_b128 v[248:251], %1 offset:64\n\tds_read_b128 v[252:255], %1 offset:96"
                         : "={v[240:255]}"(iv) : "v"(ax) : "memory");
        }
    };
    // K-step kc of group g; first = the block's first step: the accumulator starts from the block's start values (as the
    // MFMA's C operand: no copies)
    auto mfma3 = [&](int g, int kc, bool first, const f32x16& iv) __attribute__((always_inline)) {
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[kc % 4], qh[g][kc], first ? iv : acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[kc % 4], qh[g][kc], acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[kc % 4], ql[g][kc], acc[g], 0, 0, 0);
    };
    // best score of a lane's 16 (a tree: three dependent steps)
    auto block_max = [&](int g, const float* ax) __attribute__((always_inline)) -> float {
        float m[5];
#pragma unroll
        for (int j = 0; j < 5; ++j)
            m[j] = fmaxf(fmaxf(score_of(acc[g], 3 * j, ax), score_of(acc[g], 3 * j + 1, ax)), score_of(acc[g], 3 * j + 2, ax));
        return fmaxf(fmaxf(fmaxf(m[0], m[1]), m[2]), fmaxf(fmaxf(m[3], m[4]), score_of(acc[g], 15, ax)));
    };
    // the three MFMAs of (g, kc) with the VALU work that precedes this call in the same block spread into their shadow
    // (an MFMA holds the matrix pipe 32 clocks; the wave issues ~4 other instructions meanwhile.  A check placed BEHIND
    // its MFMAs -- a dozen dependent VALU instructions, a compare and a branch -- left the pipe idle ~150 clocks)
    auto mfma3_over = [&](int g, int kc, bool first) __attribute__((always_inline)) {
        mfma3(g, kc, first, iv);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
    };
    constexpr int kInitReads = MODE == SC_L2 ? 4 : 0;
    const uint32_t ring_a = lds0, aux_a = lds0 + kRing * kStageBytes + 16 * h;   // LDS byte addresses
    // (unconditional, also for a split without tiles: a branch here makes `iv` a merge of two values, and the compiler
    //  keeps such merges alive with copies -- of registers whose read is still in flight)
    load_init(aux_a);
    load_frag(ring_a, 0);
    load_frag(ring_a, 1);
    for (int t = 0; t < nstages; ++t) {
        const uint32_t th_ = ring_a + (t % kRing) * kStageBytes;   // hi tile; lo tile kHalfBytes behind
        const float* axs = auxr + (t % kAuxRing) * BF_BN + 4 * h;
        const int row0 = r_begin + t * stage_rows;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const uint32_t rp = th_ + blk * 32 * 256;
            // the block behind this one: block 1 of this stage, or block 0 of the next stage
            const uint32_t nrp = blk == 0 ? th_ + 32 * 256 : ring_a + ((t + 1) % kRing) * kStageBytes;
            const uint32_t nax = blk == 0 ? aux_a + ((t % kAuxRing) * BF_BN + 32) * 4 : aux_a + (((t + 1) % kAuxRing) * BF_BN) * 4;
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                // reads of step kc+2, then the wait for step kc's fragments (+ the start values at kc = 0)
                if (kc + 2 < 8) {
                    load_frag(rp, kc + 2);
                    wait_frag(std::integral_constant<int, 4>{}, kc, kc == 0 && MODE == SC_L2);
                } else {
                    // (also in the very last block, where the "next block" is a ring slot nobody filled: the same
                    //  straight-line code everywhere.  A branch around these asm statements makes the compiler split
                    //  the fragments' live ranges with copies on either side of the waits -- copies of bytes in flight.)
                    if (kc == 6) load_init(nax);
                    load_frag(nrp, kc + 2 - 8);
                    // in flight behind step kc's fragments: step 7's (kc = 6 only), the start values, the next block's
                    if (kInitReads) wait_frag(std::integral_constant<int, 8>{}, kc, false);
                    else wait_frag(std::integral_constant<int, 4>{}, kc, false);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (kc == 0) {
                    // groups 0 .. QG-2 were checked at the end of the previous block; the last one is checked here,
                    // under the MFMAs of group 0
                    if (have_pv) {
                        const float m = block_max(QG - 1, pv_ax);
                        mfma3_over(0, 0, true);
                        __builtin_amdgcn_sched_barrier(0);
                        finish_check(m, acc[QG - 1], QG - 1, pv_ax, pv_row0);
                    } else {
                        mfma3(0, 0, true, iv);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 1; g < QG; ++g) mfma3(g, 0, true, iv);
                } else if (kc == 7) {
                    mfma3(0, 7, false, acc[0]);
#pragma unroll
                    for (int g = 1; g < QG; ++g) {
                        __builtin_amdgcn_sched_barrier(0);
                        const float m = block_max(g - 1, axs + blk * 32);
                        mfma3_over(g, 7, false);
                        __builtin_amdgcn_sched_barrier(0);
                        finish_check(m, acc[g - 1], g - 1, axs + blk * 32, row0 + blk * 32);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < QG; ++g) mfma3(g, kc, false, acc[0]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            pv_ax = axs + blk * 32;
            pv_row0 = row0 + blk * 32;
            have_pv = true;
            if (blk == 0) {
                // stage t+1 must have landed (stage t+2, kPieces + 1 = 9 DMA instructions per wave, may stay in flight);
                // behind the barrier no wave reads stage t-1 any more: its slot takes stage t+3
                if (t + 2 < nstages) asm volatile("s_waitcnt vmcnt(9)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                if (t + kRing - 1 < nstages) issue_tile(t + kRing - 1);
            }
        }
    }
    // (the prefetches the last block issued for a block that does not exist: their registers are free only now)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (have_pv) finish_check(block_max(QG - 1, pv_ax), acc[QG - 1], QG - 1, pv_ax, pv_row0);  // the last block's last group
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const int qidx = (qt * NW + wave) * (32 * QG) + g * 32 + l31;
        if constexpr (SAMPLE) {
            float* o = a.top8 + (((size_t)qidx * a.nsplit + split) * 2 + h) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = t8[g][i];
        } else {
            a.list_cnt[((size_t)qidx * a.nsplit + split) * 2 + h] = cnt[g];
        }
    }
}

// ---------------------------------------------------------------------------------------
// The same scan with ONE bf16 product per score (q_hi . b_hi): a third of the MFMAs, half the tile stream.  Its scores
// carry an error of up to E1 = 2^-7 |q||b| (both operands rounded to 8 significant bits); the threshold kernel hands a
// query tile to this kernel only when the sample shows that much room -- twice over -- between the score every top-k
// row must reach and a threshold that still lists few rows (bf_f32_threshold_kernel), and the re-rank proves the
// outcome with E1 in place of the split product's 2^-14 (bf_rerank_f32_list_kernel).  Tiles without that room (low
// dimensions, tightly packed neighbours) go through bf_scan_f32_kernel.
// Four waves, one per SIMD, QG groups of 32 queries each; the 32-row blocks alternate between two sets of
// accumulators: the scores of block n are checked -- one group every other K-step, its dozen VALU instructions spread
// under that step's MFMAs -- while block n+1 accumulates into the other set.
// ---------------------------------------------------------------------------------------
// SAMPLE = the sample pass (every tile_stride-th tile; per-lane top-8 of the blocks' best scores instead of lists): the
// SAME arithmetic as the scan, so a sampled row scores the same bits in both.
// KCH > 1 (round 3): rows longer than 128 -- KCH chunks of 128 dimensions, one stage per (64-row tile, chunk); the two
// blocks of a tile keep accumulating through the tile's KCH stages and are checked after its last one; the query
// fragments of all chunks live in AGPRs (32 * KCH per group: QG = 1, NW = 4 -- 128 queries per workgroup).
template <int MODE, bool SAMPLE, int QG, int NW, int KCH = 1>
__global__ __launch_bounds__(NW * 64) void bf_scan_bf16_kernel(BfScanF32Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.x;
    const int xcd = b & 7, rest = b >> 3;
    const int qt = rest % a.nqt;
    const int split = (rest / a.nqt) * 8 + xcd;
    if (split >= a.nsplit) return;
    if (a.group_flag && (a.group_flag[qt] != 0) != (a.group_want != 0)) return;

    // hi tile only: 16 KB stages.  The waves meet every kSync stages (a wave that met a hit runs ~200 clocks late; the
    // fewer meetings, the less of that lateness everyone waits for): ring = the stage being read + kSync landed +
    // kSync in flight
    constexpr int kSync = 4, kRing = 2 * kSync + 1, kAuxRing = 16, kStageBytes = BF_BN * 256;
    char* ring = smem;
    float* auxr = reinterpret_cast<float*>(ring + kRing * kStageBytes);  // [kAuxRing][BN]
    const int tstr = SAMPLE ? a.tile_stride : 1;
    const int tiles_all = (a.n + BF_BN - 1) / BF_BN;
    const int stiles_all = (tiles_all + tstr - 1) / tstr;
    const int ntiles = max(0, min(a.tps, stiles_all - split * a.tps));
    const int nstages = ntiles * KCH;   // one stage per (tile, chunk of 128 dimensions)
    constexpr int kDp = 128 * KCH;       // row length of the bf16 tiles
    const int r_begin = split * a.tps * tstr * BF_BN;
    const int stage_rows = tstr * BF_BN;

    // one stage = 16 DMA pieces of 1 KiB (4 rows x 256 B): wave w issues pieces (16/NW) w .. and the stage's 64 aux values.
    // Written out in the scalar-base form (round 3): `global_load_lds_dwordx4 voff, s[base]` with the lane's constant byte
    // offset in a VGPR and the stage's base in an SGPR pair -- through the builtin every piece cost a 64-bit address sum per
    // lane, a v_readfirstlane for M0 and an exec-masked tail for the aux values, ~45 instructions per stage issued by all
    // eight waves in one burst behind each barrier.  Every wave requests all 64 aux values (one instruction, no exec mask;
    // the waves write the same bytes).
    uint32_t dma_voff[16 / NW];
#pragma unroll
    for (int jj = 0; jj < 16 / NW; ++jj) {
        const int pj = (16 / NW) * wave + jj;
        const int row = 4 * pj + (lane >> 4);
        const int c = (lane & 15) ^ (row & 15);
        dma_voff[jj] = (uint32_t)((row * kDp + c * 8) * 2);
    }
    const uint32_t dma_lds0 = (uint32_t)(uintptr_t)(lptr_t)ring + (uint32_t)((16 / NW) * __builtin_amdgcn_readfirstlane(wave) * 1024);   // (a scalar: it goes to M0)
    const uint32_t aux_lds0 = (uint32_t)(uintptr_t)(lptr_t)auxr;
    const uint32_t aux_voff = (uint32_t)lane * 4u;
    auto issue_tile = [&](int stage) __attribute__((always_inline)) {
        const int slot = stage % kRing;
        const int row0 = r_begin + (stage / KCH) * stage_rows;
        const unsigned long long sbase = (unsigned long long)(uintptr_t)a.base_h16 + ((unsigned long long)row0 * kDp + (stage % KCH) * 128) * 2ull;
        const uint32_t m0v = dma_lds0 + (uint32_t)slot * kStageBytes;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#pragma unroll
        for (int jj = 0; jj < 16 / NW; ++jj)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                         :: "v"(dma_voff[jj]), "s"(sbase), "s"(m0v + (uint32_t)jj * 1024u) : "memory", "m0");
        const unsigned long long abase = (unsigned long long)(uintptr_t)a.auxp16 + (unsigned long long)row0 * 4ull;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1"
                     :: "v"(aux_voff), "s"(abase), "s"(aux_lds0 + (uint32_t)((stage % kAuxRing) * BF_BN * 4)) : "memory", "m0");
#pragma clang diagnostic pop
    };

    f16x8 qh[QG][8 * KCH];
    float thr[QG];
    int cnt[QG], ecnt[QG];
    uint32_t* lp[QG];
    float t8[SAMPLE ? QG : 1][8];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const int qidx = (qt * NW + wave) * (32 * QG) + g * 32 + l31;
#pragma unroll
        for (int kc = 0; kc < 8 * KCH; ++kc)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(qh[g][kc]) : "v"(static_cast<const _Float16*>(a.q_h16) + (size_t)qidx * kDp + 16 * kc + 8 * h) : "memory");
        cnt[g] = 0;
        ecnt[g] = 0;
        if constexpr (SAMPLE) {
            thr[g] = -INFINITY;
            lp[g] = nullptr;
#pragma unroll
            for (int i = 0; i < 8; ++i) t8[g][i] = -INFINITY;
        } else {
            thr[g] = a.thr[qidx];
            lp[g] = a.list + (size_t)qidx * a.caph * (2 * a.nsplit) + split * 2 + h;   // entry i of a list: [q][i][list]
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int t = 0; t <= kSync && t < nstages; ++t) issue_tile(t);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    // LDS address of this lane's fragment of K-step kc in the stage being read (its first block; the second sits 8 KB
    // behind: an immediate offset of the read).  Moved to the next slot of the ring once per stage -- one VALU per K-step and
    // stage instead of one address addition per read (round 3: 8 of the ~100 non-MFMA instructions of a stage's block pair).
    uint32_t vad[8];
#pragma unroll
    for (int kc = 0; kc < 8; ++kc)
        vad[kc] = (uint32_t)(uintptr_t)(lptr_t)smem + (uint32_t)(l31 * 256 + (((kc * 2 + h) ^ (l31 & 15)) * 16));

    auto score_of = [&](const f32x16& c, int i, const float* ax) __attribute__((always_inline)) -> float {
        if constexpr (MODE == SC_COS) return c[i] * ax[(i & 3) + 8 * (i >> 2)];
        else return c[i];
    };
    // maximum of a lane's 16 scores: seven v_max3 and a v_max.  (The leaf triples are written out: from nested fmaxf the
    // compiler folds one leaf into the root and builds its max(c0, c1) with a canonicalising v_max x, x in front of each --
    // ten instructions where eight do, per check.)
    auto block_max_leaves = [&](const f32x16& c, const float* ax, float (&m)[5]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if constexpr (MODE == SC_COS)
                m[j] = fmaxf(fmaxf(score_of(c, 3 * j, ax), score_of(c, 3 * j + 1, ax)), score_of(c, 3 * j + 2, ax));
            else
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m[j]) : "v"(c[3 * j]), "v"(c[3 * j + 1]), "v"(c[3 * j + 2]));
        }
    };
    auto block_max_root = [&](const f32x16& c, const float* ax, const float (&m)[5]) __attribute__((always_inline)) -> float {
        if constexpr (MODE == SC_COS) {
            return fmaxf(fmaxf(fmaxf(m[0], m[1]), m[2]), fmaxf(fmaxf(m[3], m[4]), score_of(c, 15, ax)));
        } else {
            float r0, r1, r;
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(m[0]), "v"(m[1]), "v"(m[2]));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r1) : "v"(m[3]), "v"(m[4]), "v"(c[15]));
            asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(r0), "v"(r1));
            return r;
        }
    };
    auto block_max = [&](const f32x16& c, const float* ax) __attribute__((always_inline)) -> float {
        float m[5];
        block_max_leaves(c, ax, m);
        return block_max_root(c, ax, m);
    };
    // (see bf_scan_f32_kernel)
    // does any lane's block maximum reach its threshold?  (a wave-uniform value in a scalar register)
    auto trigger_of = [&](float m, int g) __attribute__((always_inline)) -> unsigned long long {
        return SAMPLE ? __builtin_amdgcn_ballot_w64(m > thr[g]) : __builtin_amdgcn_ballot_w64(m >= thr[g]);
    };
    auto finish_check = [&](unsigned long long trig, float m, const f32x16& c, int g, const float* ax, int row0) __attribute__((always_inline)) {
        if constexpr (SAMPLE) {
            // only the block's best value enters the lane's top-8; pad rows of the dot / cosine modes score 0: the tile
            // that holds them stays out of the estimate
            if (__builtin_expect(trig != 0ull, 0)) {
                float v = (row0 + 32 <= a.n || MODE == SC_L2) ? m : -INFINITY;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float hi = fmaxf(t8[g][j], v);
                    v = fminf(t8[g][j], v);
                    t8[g][j] = hi;
                }
                thr[g] = t8[g][7];
            }
        } else if (__builtin_expect(trig != 0ull, 0)) {   // (cold: the common path falls through)
            uint32_t km;
            if constexpr (MODE == SC_COS) {
                f32x16 sc;
#pragma unroll
                for (int i = 0; i < 16; ++i) sc[i] = score_of(c, i, ax);
                km = hit_mask_f32(sc, thr[g]);
            } else {
                km = hit_mask_f32(c, thr[g]);
            }
            if (row0 + 32 > a.n) {
                int nvalid = a.n - row0;
                asm volatile("" : "+s"(nvalid));
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (acc_row(i, h) >= nvalid) km &= ~(0x8000u >> i);
            }
            if (km) {
                if (ecnt[g] < a.caph) lp[g][(size_t)ecnt[g] * (2 * a.nsplit)] = ((uint32_t)(row0 - r_begin) >> 5 << 16) | km;
                ecnt[g]++;
                cnt[g] += __builtin_popcount(km);
            }
        }
    };

    f16x8 fh[4];
    f32x16 iv;
#pragma unroll
    for (int i = 0; i < 16; ++i) iv[i] = 0.f;
    f32x16 acc[2][QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[0][g][i] = 0.f;
            acc[1][g][i] = -INFINITY;   // "the block before the first": reaches no threshold (no have-a-previous-block test
        }                               //  in the loop: a uniform flag costs a VALU -> SALU round trip per branch)
    }
    const float* pv_ax = auxr;
    int pv_row0 = 0;
    bool have_pv = false;
    float m_pend = 0.f;       // maximum (and trigger) of the group whose branch comes in the next K-step
    unsigned long long trig_pend = 0ull;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    // fragment of K-step kc of the current stage's block 0 (OFF = 0) or block 1 (OFF = 8192)
    auto load_frag = [&](auto off_tag, int kc) __attribute__((always_inline)) {
        constexpr int OFF = decltype(off_tag)::value;
        const uint32_t ad = vad[kc];
        if constexpr (NW == 4) {
            switch (kc % 4) {   // (kc is a constant after unrolling)
                case 0: BF_FRAG_RD_OFF("v[224:227]", fh[0], ad, OFF); break;
                case 1: BF_FRAG_RD_OFF("v[228:231]", fh[1], ad, OFF); break;
                case 2: BF_FRAG_RD_OFF("v[232:235]", fh[2], ad, OFF); break;
                default: BF_FRAG_RD_OFF("v[236:239]", fh[3], ad, OFF); break;
            }
        } else {   // (two waves per SIMD share the register file: stay low)
            switch (kc % 4) {
                case 0: BF_FRAG_RD_OFF("v[96:99]", fh[0], ad, OFF); break;
                case 1: BF_FRAG_RD_OFF("v[100:103]", fh[1], ad, OFF); break;
                case 2: BF_FRAG_RD_OFF("v[104:107]", fh[2], ad, OFF); break;
                default: BF_FRAG_RD_OFF("v[108:111]", fh[3], ad, OFF); break;
            }
        }
    };
    using Off0 = std::integral_constant<int, 0>;
    using Off1 = std::integral_constant<int, 32 * 256>;
    // the counted wait that makes slot kc % 4 (and, with with_init, the start values) valid
    auto wait_frag = [&](auto cnt_tag, int kc, bool with_init) __attribute__((always_inline)) {
        constexpr int N = decltype(cnt_tag)::value;
#define BF_W1(STR)                                                                                              \
        if constexpr (NW == 4) {                                                                                \
            switch (kc % 4) {                                                                                   \
                case 0:                                                                                         \
                    if (with_init) asm volatile(STR : "+{v[224:227]}"(fh[0]), "+{v[240:255]}"(iv));             \
                    else asm volatile(STR : "+{v[224:227]}"(fh[0]));                                            \
                    break;                                                                                      \
                case 1: asm volatile(STR : "+{v[228:231]}"(fh[1])); break;                                      \
                case 2: asm volatile(STR : "+{v[232:235]}"(fh[2])); break;                                      \
                default: asm volatile(STR : "+{v[236:239]}"(fh[3])); break;                                     \
            }                                                                                                   \
        } else {                                                                                                \
            switch (kc % 4) {                                                                                   \
                case 0:                                                                                         \
                    if (with_init) asm volatile(STR : "+{v[96:99]}"(fh[0]), "+{v[112:127]}"(iv));               \
                    else asm volatile(STR : "+{v[96:99]}"(fh[0]));                                              \
                    break;                                                                                      \
                case 1: asm volatile(STR : "+{v[100:103]}"(fh[1])); break;                                      \
                case 2: asm volatile(STR : "+{v[104:107]}"(fh[2])); break;                                      \
                default: asm volatile(STR : "+{v[108:111]}"(fh[3])); break;                                     \
            }                                                                                                   \
        }
        if constexpr (N == 0) { BF_W1("s_waitcnt lgkmcnt(0)") }
        else if constexpr (N == 2) { BF_W1("s_waitcnt lgkmcnt(2)") }
        else if constexpr (N == 3) { BF_W1("s_waitcnt lgkmcnt(3)") }
        else if constexpr (N == 6) { BF_W1("s_waitcnt lgkmcnt(6)") }
        else { static_assert(N == 7, "wait count"); BF_W1("s_waitcnt lgkmcnt(7)") }
#undef BF_W1
    };
    // one MFMA of K-step kc.  A block's first MFMAs take the start values (-|b|^2 / 2, read from LDS into the pinned tuple
    // `iv`) as their C operand.  Through the builtin the compiler first COPIES the tuple into the accumulator's registers
    // (8 v_mov_b64 + an s_nop per block); the instruction written out names the pinned registers as C (round 3).
    auto mfma_step = [&](f32x16& c, const f16x8& q, int kc, bool first_ch) __attribute__((always_inline)) {
        if constexpr (MODE == SC_L2) {
            if (kc == 0 && first_ch) {
                if constexpr (NW == 4)
                    asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(c) : "{v[224:227]}"(fh[0]), "a"(q), "{v[240:255]}"(iv));
                else
                    asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(c) : "{v[96:99]}"(fh[0]), "a"(q), "{v[112:127]}"(iv));
                return;
            }
        }
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[kc % 4], q, (kc == 0 && first_ch) ? iv : c, 0, 0, 0);
    };
    auto load_init = [&](uint32_t ax) __attribute__((always_inline)) {
        if constexpr (MODE == SC_L2) {
            if constexpr (NW == 4)
                asm volatile("ds_read_b128 v[240:243], %1\n\tds_read_b128 v[244:247], %1 offset:32\n\t"
                             "ds_read_b128 v[248:251], %1 offset:64\n\tds_read_b128 v[252:255], %1 offset:96"
                             : "={v[240:255]}"(iv) : "v"(ax) : "memory");
            else   // (two waves per SIMD share the register file: stay low)
                asm volatile("ds_read_b128 v[112:115], %1\n\tds_read_b128 v[116:119], %1 offset:32\n\t"
                             "ds_read_b128 v[120:123], %1 offset:64\n\tds_read_b128 v[124:127], %1 offset:96"
                             : "={v[112:127]}"(iv) : "v"(ax) : "memory");
        }
    };
    constexpr int kInitReads = MODE == SC_L2 ? 4 : 0;
    const uint32_t aux_a = lds0 + kRing * kStageBytes + 16 * h;
    // (unconditional, also for a split without tiles: a branch here makes `iv` a merge of two values, and the compiler
    //  keeps such merges alive with copies -- of registers whose read is still in flight)
    // fragment reads run kPre K-steps ahead of their MFMAs (round 3: 3, was 2 -- the four fragment slots allow it: the slot
    // a read lands in was consumed one step earlier)
    constexpr int kPre = 3;
    // The two waves of a SIMD (w and w + 4) run the same program and meet at every barrier: in lockstep they reach their
    // MFMA pairs and their check's VALU work together.  Waves 4..7 check the previous block two K-steps later (stagger:
    // MI355X_MICROARCH.md, "Two waves per SIMD", item 9) -- same instructions, same results, other K-steps.
    // (Each copy of the loop carries its own prologue reads and its own final wait: a pinned fragment register alive across
    //  the branch between the copies is merged with COPIES of registers whose reads are in flight -- tests/test_isa_audit.py.)
    auto main_loop = [&](auto stag_tag) __attribute__((always_inline)) {
    constexpr int kStag = decltype(stag_tag)::value;
    // (static priority for the second-dispatched half, the arbitration loser of every SIMD pair: same guide, item 4;
    //  NMSLIB_GPU_BF16_PRIO=0 in the launcher's environment switches it off for experiments)
    if (kStag != 0 && a.prio_half) __builtin_amdgcn_s_setprio(1);
    load_init(aux_a);
#pragma unroll
    for (int kc = 0; kc < kPre; ++kc) load_frag(Off0{}, kc);
    for (int tt = 0; tt < ntiles; ++tt) {
#pragma unroll
      for (int ch = 0; ch < KCH; ++ch) {
        const int t = tt * KCH + ch;
        constexpr bool kOne = KCH == 1;
        const bool first_ch = kOne || ch == 0, last_ch = kOne || ch == KCH - 1;
        const float* axs = auxr + (t % kAuxRing) * BF_BN + 4 * h;
        const int row0 = r_begin + tt * stage_rows;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            // the block before this one holds a tile's final scores iff it ran the tile's last chunk; this block
            // starts from the start values iff it runs the tile's first chunk; the block behind it likewise
            const bool prev_final = blk == 1 ? last_ch : first_ch;
            const bool init_next = blk == 0 ? first_ch : last_ch;
            // (block 1's last two K-steps prefetch from the NEXT stage's slot: the addresses move there one by one)
            const int dslot = ((t + 1) % kRing == 0) ? -(kRing - 1) * kStageBytes : kStageBytes;
            const uint32_t nax = blk == 0 ? aux_a + ((t % kAuxRing) * BF_BN + 32) * 4 : aux_a + (((t + 1) % kAuxRing) * BF_BN) * 4;
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                // the read of step kc+kPre, then the wait for step kc's fragment: at most the reads of steps kc+1 .. kc+kPre
                // (and the next block's start values) stay in flight
                if (kc + kPre < 8) {
                    if (blk == 0) load_frag(Off0{}, kc + kPre);
                    else load_frag(Off1{}, kc + kPre);
                    wait_frag(std::integral_constant<int, kPre>{}, kc, kc == 0 && MODE == SC_L2 && first_ch);
                } else {
                    // (also in the very last block: see bf_scan_f32_kernel)
                    if (kc == 6 && init_next) load_init(nax);
                    if (blk == 0) {
                        load_frag(Off1{}, kc + kPre - 8);
                    } else {
                        vad[kc + kPre - 8] += (uint32_t)dslot;
                        load_frag(Off0{}, kc + kPre - 8);
                    }
                    // (from K-step 6 on the next block's four start-value reads are in flight too, older than the last read)
                    if (kInitReads && init_next && kc >= 6) wait_frag(std::integral_constant<int, kPre + 4>{}, kc, false);
                    else wait_frag(std::integral_constant<int, kPre>{}, kc, false);
                }
                __builtin_amdgcn_sched_barrier(0);
                // The previous block's scores are checked one group per kEvery K-steps, in two halves: step gc * kEvery
                // computes the group's maximum, its dozen VALU instructions spread under that step's MFMAs; the NEXT
                // step compares and branches right behind its first MFMA, when the maximum is long there -- a compare +
                // branch at the end of the step that computes it costs ~125 clocks (the wave issues in order: the branch
                // waits for the VALU chain that waited for the step's last MFMA to issue; measured 252 vs 129 clocks
                // per K-step).
                constexpr int kEvery = 8 / QG;
                const int gc = kc >= kStag ? (kc - kStag) / kEvery : 0;
                const bool chk = prev_final && kc >= kStag && ((kc - kStag) % kEvery) == 0;
                const bool br = prev_final && kc >= kStag && ((kc - kStag) % kEvery) == 1;
                float m = 0.f;
                // (the written-out MFMAs of a block's first K-step are invisible to sched_group_barrier: there the check's
                //  VALU work is placed by hand -- leaves under the first MFMA, root under the second)
                const bool by_hand = MODE == SC_L2 && kc == 0 && first_ch && chk && QG == 2;
                float m5[5];
                if (chk && !by_hand) m = block_max(acc[blk ^ 1][gc], pv_ax);   // (also before the first block: no branch between the MFMAs)
                mfma_step(acc[blk][0], qh[0][8 * ch + kc], kc, first_ch);
                if (by_hand) {
                    if constexpr (QG == 2) {
                        __builtin_amdgcn_sched_barrier(0);
                        block_max_leaves(acc[blk ^ 1][gc], pv_ax, m5);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_step(acc[blk][1], qh[1][8 * ch + kc], kc, first_ch);
                        __builtin_amdgcn_sched_barrier(0);
                        m = block_max_root(acc[blk ^ 1][gc], pv_ax, m5);
                    }
                } else if (br) {
                    __builtin_amdgcn_sched_barrier(0);
                    finish_check(trig_pend, m_pend, acc[blk ^ 1][gc], gc, pv_ax, pv_row0);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 12 / QG, 0);
                }
#pragma unroll
                for (int g = 1; g < QG; ++g) {
                    if (by_hand) break;
                    mfma_step(acc[blk][g], qh[g][8 * ch + kc], kc, first_ch);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 12 / QG, 0);
                }
                if (chk) {
                    // the compare too belongs to this step: the branch of the next step then tests a scalar register that
                    // has been there for a hundred clocks, not a VALU result.  (Pinned: the compiler would otherwise sink
                    // the maximum and the compare into the next step's branch.)
                    asm volatile("" : "+v"(m));
                    unsigned long long trig = trigger_of(m, gc);
                    asm volatile("" : "+s"(trig));
                    m_pend = m;
                    trig_pend = trig;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (last_ch) {
                pv_ax = axs + blk * 32;
                pv_row0 = row0 + blk * 32;
                have_pv = true;
            }
            if (blk == 1) {
#pragma unroll
                for (int kc = kPre; kc < 8; ++kc) vad[kc] += (uint32_t)dslot;
            }
            if (blk == 0) {
                // every kSync stages: the next kSync stages (requested at the last meeting) must have landed; behind the
                // barrier no wave reads the kSync stages before this one any more: their slots take the next requests
                if (t % kSync == 0) {
                    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                    for (int u = t + kSync + 1; u <= t + 2 * kSync && u < nstages; ++u) issue_tile(u);
                }
            }
        }
      }
    }
    // (the prefetches the last block issued for a block that does not exist: their registers are free only now)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    if constexpr (NW == 8) {
        if (__builtin_amdgcn_readfirstlane(wave) >= 4) main_loop(std::integral_constant<int, 2>{});
        else main_loop(std::integral_constant<int, 0>{});
    } else {
        main_loop(std::integral_constant<int, 0>{});
    }
    if (have_pv) {   // the last block (nstages > 0: it was block 1 of its stage)
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            const float m = block_max(acc[1][g], pv_ax);
            finish_check(trigger_of(m, g), m, acc[1][g], g, pv_ax, pv_row0);
        }
    }
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const int qidx = (qt * NW + wave) * (32 * QG) + g * 32 + l31;
        if constexpr (SAMPLE) {
            float* o = a.top8 + (((size_t)qidx * a.nsplit + split) * 2 + h) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = t8[g][i];
        } else {
            a.list_cnt[((size_t)qidx * a.nsplit + split) * 2 + h] = cnt[g];
        }
    }
}

// rows -> two bf16 tiles (hi = bf16(x), lo = bf16(x - hi)), padded to 128 columns and n_pad rows; auxp = aux with the
// pad rows' value
// h16 (nullable) = fp16(scale * x) for the one-product scan, auxp16 its start values (aux * aux16_mul)
__global__ void split_bf16_kernel(const float* src, int rows, int rows_pad, int ld, int dim, __bf16* hi, __bf16* lo,
                                  const float* aux, float aux_pad, float* auxp, int dp, _Float16* h16, float scale,
                                  float* auxp16, float aux16_mul) {
    const size_t total = (size_t)rows_pad * dp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / (size_t)dp;
        const int c = (int)(i - r * dp);
        const float v = (r < (size_t)rows && c < dim) ? src[r * ld + c] : 0.f;
        const __bf16 hh = (__bf16)v;
        if (hi) hi[i] = hh;
        if (lo) lo[i] = (__bf16)(v - (float)hh);
        if (h16) h16[i] = (_Float16)(scale * v);
        if (c == 0 && auxp) auxp[r] = r < (size_t)rows ? (aux ? aux[r] : 0.f) : aux_pad;
        if (c == 0 && auxp16) auxp16[r] = r < (size_t)rows ? (aux ? aux[r] * aux16_mul : 0.f) : aux_pad;
    }
}

// Query preparation of the float fast path in ONE launch (round 3; was pad_rows + split_bf16 + four fillBuffers):
//   raw != null: queries [nq][dim] -> padded f32 copy pad_out [qpad][ld] (zeros beyond nq / dim) and its bf16 split;
//   raw == null: `sel` [qpad][ld] (already padded, centred) -> bf16 split only;
//   and the words every later kernel of the batch expects to find cleared: the fast path's tile flags, the verified
//   adaptive path's flags, both shared-threshold regions of the fallback selections.
__global__ void bf_f32_prep_kernel(const float* raw, int nq, int dim, const float* sel, int qpad, int ld, float* pad_out,
                                   __bf16* hi, __bf16* lo, int* clr0, int n0, int* clr1, int n1, uint32_t* clr2, size_t n2,
                                   int dp, _Float16* h16, float scale) {
    const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (size_t)gridDim.x * blockDim.x;
    const size_t total = (size_t)qpad * dp;
    for (size_t i = gtid; i < total; i += gsz) {
        const size_t r = i / (size_t)dp;
        const int c = (int)(i - r * dp);
        float v;
        if (raw) {
            v = (r < (size_t)nq && c < dim) ? raw[r * dim + c] : 0.f;
            if (c < ld) pad_out[r * ld + c] = v;
        } else {
            v = c < dim ? sel[r * ld + c] : 0.f;
        }
        const __bf16 hh = (__bf16)v;
        if (hi) hi[i] = hh;
        if (lo) lo[i] = (__bf16)(v - (float)hh);
        h16[i] = (_Float16)(scale * v);
    }
    for (size_t i = gtid; i < (size_t)n0; i += gsz) clr0[i] = 0;
    for (size_t i = gtid; i < (size_t)n1; i += gsz) clr1[i] = 0;
    for (size_t i = gtid; i < n2; i += gsz) clr2[i] = 0u;
}

// Error bound of the one-product score q_hi . b_hi of one query against any row (first wave of the workgroup; result
// in every lane):  |q.b - q^.b^| <= |q - q^||b| + |q^||b - b^|  with the ACTUAL rounding residual of this query and the
// largest residual of the rows (bres; relative to |b| for the cosine score, where bscale = 1), plus the f32 accumulation
// of the MFMAs.  Typically ~0.4 of the worst case 2^-7 |q||b|.
// Round 3: the scan's operands are fp16(scale * x); everything here is in the scan's units (queries and rows times `scale`:
// bscale = scale * |b|max -- or scale for the cosine score --, bres the rows' largest residual |scale b - fp16(scale b)|).
// A query beyond fp16's range (|scale q_i| > 65504 -> inf) gets an infinite bound: its tile leaves the one-product scan.
__device__ __forceinline__ float one_product_error(const float* qs, int dim, int lane, float bscale, float bres, float scale) {
    float ss = 0.f, rr = 0.f;
    bool over = false;
    for (int d = lane; d < dim; d += 64) {
        const float v = scale * qs[d], w = v - (float)(_Float16)v;
        over |= !(fabsf(v) <= 65504.f);
        ss = fmaf(v, v, ss);
        rr = fmaf(w, w, rr);
    }
    const float qn = sqrtf(wave_sum(ss)), qr = sqrtf(wave_sum(rr));
    if (__any(over)) return INFINITY;
    return 1.01f * (qr * bscale + 1.004f * qn * bres) + 2e-5f * qn * bscale;
}

// Thresholds of one query from the sample (the union of the lanes' top-8 lists, sorted); one workgroup per query.
// The sample pass scores rows with ONE bf16 product: every sample score s^ is within E1 (one_product_error) of the
// row's exact score s.  Let s^_r be the r-th best sample score: at least k' rows of the whole base score s^ >= s^_r
// (see bf_f32_fast_plan), hence exact s >= s^_r - E1: the exact score S_k of the k-th neighbour is >= s^_r - E1.
//   thr3 = s^_r - 1.02 E1 for the split-product scan (whose scores are exact to E3 << E1): it lists those k' rows.
//   thr1 = threshold of the one-product scan (the sample's own arithmetic: a sampled row scores the same bits): the
//          first sample score at least 2.1 E1 below s^_r.  Unlisted rows have s^ < thr1, so s < thr1 + E1
//          <= s^_r - 1.1 E1 < S_k: the re-rank's proof holds.  If the sample has no such score among its best rcap --
//          the scores near the top are packed more tightly than the one-product error -- the query's tile is flagged
//          `precise` and goes through the split-product scan.
// Only the `depth` best of each lane's eight take part (a lane holds 1/nlists of the sample: more than four of the
// best rcap in one lane is rare, and a dropped value only lowers a threshold).
// One wave per query, the sample's values in registers (round 3): s^_r is ONE order statistic (wave_rth_largest_u32), and
// thr1 -- the first value at least 2.1 E1 below it in descending order -- is the LARGEST such value (the float subtraction is
// monotone), a masked maximum; its place in the order, which must not exceed rcap, is a count.  Same thresholds, bit for
// bit, as the sorted array gave.
template <int NPER>
__global__ __launch_bounds__(256) void bf_f32_threshold_kernel(const float* top8, int nlists, int depth, int r, int rcap, int nq,
                                                               int qpad, const float* queries_sel, int ldb, int dim, float bscale,
                                                               float bres, int group_q, int force_precise, float* thr3,
                                                               float* thr1, int* precise, float scale_q, float unit) {
    const int lane = threadIdx.x & 63, q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= qpad) return;
    if (q >= nq) {  // padding queries: nothing passes
        if (lane == 0) {
            thr3[q] = INFINITY;
            thr1[q] = INFINITY;
        }
        return;
    }
    const int total = nlists * depth;
    uint32_t key[NPER];
#pragma unroll
    for (int i = 0; i < NPER; ++i) {
        const int idx = i * 64 + lane;
        const int l = depth == 4 ? idx >> 2 : idx >> 3, j = idx - l * depth;   // (depth is 4 or 8)
        key[i] = idx < total ? f32_ord(top8[((size_t)q * nlists + l) * 8 + j]) : 0u;
    }
    const float e1 = one_product_error(queries_sel + (size_t)q * ldb, dim, lane, bscale, bres, scale_q);   // (the scan's units)
    const uint32_t T = total >= r ? wave_rth_largest_u32<NPER>(key, r) : 0u;
    const float t3 = total >= r ? ord_f32(T) : -INFINITY;   // s^_r
    float t1 = t3;
    bool ok = force_precise != 1;
    if (ok && t3 > -INFINITY) {
        const float need = 2.1f * e1;
        uint32_t best = 0u;     // largest key at or below s^_r whose value is at least `need` below it (0: none)
#pragma unroll
        for (int i = 0; i < NPER; ++i) {
            const float v = ord_f32(key[i]);
            const bool c = key[i] != 0u && key[i] <= T && v > -INFINITY && t3 - v >= need;
            best = c && key[i] > best ? key[i] : best;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t other = (uint32_t)__shfl_xor((int)best, o, 64);
            best = other > best ? other : best;
        }
        ok = false;
        if (best != 0u) {
            int above = 0;   // values in front of it in descending order
#pragma unroll
            for (int i = 0; i < NPER; ++i) above += __popcll(__builtin_amdgcn_ballot_w64(key[i] > best));
            const int jmax = rcap < total ? rcap : total;
            const int j = best == T ? r : above + 1;   // (the sorted array's search started at place r)
            if (j <= jmax) {
                t1 = ord_f32(best);
                ok = true;
            }
        }
    }
    if (lane == 0) {
        thr3[q] = (t3 - 1.02f * e1) / unit;   // (the split-product scan runs on the unscaled bf16 tiles)
        thr1[q] = force_precise == 2 ? INFINITY : t1;   // (2: NMSLIB_GPU_DEBUG & 8192 -- a scan without hits, timing experiments)
        if (!ok) atomicOr(&precise[q / group_q], 1);
    }
}

// exact distances (the reference formula on the ORIGINAL rows) of the listed rows, (distance, position) order, top k;
// verification of the threshold bet like bf_rerank_u8_list_kernel
struct RerankListF32Args {
    const float* base;        // original rows [n][ldb]
    const float* queries;     // original padded queries [qpad][ldb]
    const uint32_t* list;
    const int* list_cnt;
    const int32_t* ext_ids;
    int32_t* out_ids;
    float* out_dists;
    int32_t* out_cnt;
    int* tile_fail;
    int n, k, nsplit, caph, p2max, fail_queries, space, dim, ldb;
    int tps;                   // tiles per split of the scan (entries hold block indices inside their split)
    const float* queries_sel;  // the queries the selection saw (centred for l2 on un-centred data)
    const float* thr;          // [qpad] thresholds of the split-product scan (score units)
    const float* thr1;         // [qpad] thresholds of the one-product scan
    const int* precise;        // [query tiles] which scan served the tile (1: split product)
    int no_split;              // rows longer than 128: no split-product scan -- a `precise` tile goes to the adaptive kernel
    float bmax;                // largest row norm of the selection rows
    float bres;                // largest bf16 rounding residual of the selection rows (see row_maxnorm_kernel)
    float scale, scale_q, bres16;   // one-product tiles: the fp16 scan's scales of rows / queries, the rows' largest fp16 residual (scaled)
    int sel_dim, sel_ld;       // columns / row stride of queries_sel (dim / ldb, or the augmented queries' when qaux is set)
    const float* qaux;         // centred cosine / angular: [qpad][4] = |q'|^2, |q| - |mu|, |q|, flag (0: zero-norm query); else null
    unsigned long long* prof;  // NMSLIB_GPU_DEBUG & 4096: phase clocks (100 MHz), summed over the workgroups
};

__global__ __launch_bounds__(256) void bf_rerank_f32_list_kernel(RerankListF32Args a) {
    const unsigned long long t0 = a.prof ? wall_clock64() : 0;
    auto lap = [&](int i) __attribute__((always_inline)) {
        if (a.prof && threadIdx.x == 0) atomicAdd(&a.prof[i], wall_clock64() - t0);
    };
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);              // [p2max]
    int* offs = reinterpret_cast<int*>(keys + a.p2max);    // [2 * nsplit + 1]
    __shared__ int s_over, s_split;
    __shared__ float s_qn2, s_e1, s_thr;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = 2 * a.nsplit;
    // exclusive prefix sum of the (clipped) list lengths: wave 0, each lane a run of consecutive lists
    if (tid == 0) s_over = 0;
    __syncthreads();
    if (tid < 64) {
        const int per = (nl + 63) / 64;
        int sum = 0, over = 0;
        for (int i = 0; i < per; ++i) {
            const int s = tid * per + i;
            if (s < nl) {
                const int c = a.list_cnt[(size_t)q * nl + s];
                over |= c > a.caph;
                sum += c < a.caph ? c : a.caph;
            }
        }
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (tid >= o) incl += v;
        }
        int run = incl - sum;
        for (int i = 0; i < per; ++i) {
            const int s = tid * per + i;
            if (s < nl) {
                offs[s] = run;
                const int c = a.list_cnt[(size_t)q * nl + s];
                run += c < a.caph ? c : a.caph;
            }
        }
        if (tid == 63) offs[nl] = incl;
        if (__any(over != 0) && tid == 0) s_over = 1;
    } else if (wave == 1) {
        // what the proof at the end needs of the query alone (its norm, the error bound of its tile's scan, its threshold):
        // requested here, beside the list lengths, instead of as three more memory round trips behind the sort
        const bool split_product = a.precise[q / a.fail_queries] != 0;
        const float* qs = a.queries_sel + (size_t)q * a.sel_ld;
        float ss = 0.f;
        for (int d = lane; d < a.sel_dim; d += 64) ss = fmaf(qs[d], qs[d], ss);
        ss = wave_sum(ss);
        const float e1 = split_product ? 0.f : one_product_error(qs, a.sel_dim, lane, a.scale * (a.space == SP_L2 || a.space == SP_NEGDOT || a.qaux ? a.bmax : 1.0f), a.bres16, a.scale_q);
        if (lane == 0) {
            s_qn2 = ss;
            s_e1 = e1;
            s_split = split_product ? 1 : 0;
            s_thr = split_product ? a.thr[q] : a.thr1[q];
        }
    }
    __syncthreads();
    const int total = offs[nl];
    if ((a.no_split && a.precise[q / a.fail_queries] != 0) ||      // (no scan served this tile: its lists are stale)
        (a.qaux && a.qaux[(size_t)q * 4 + 3] == 0.f)) {            // (zero-norm query, centred cosine: every distance is 1)
        if (tid == 0) atomicOr(&a.tile_fail[q / a.fail_queries], 1);
        return;
    }
    const int need = a.k < a.n ? a.k : a.n;
    // the adaptive kernel keeps k' = k + max(4, k/8) per split: ask for the same slack over the whole base here
    const int want = need + (a.k / 8 > 4 ? a.k / 8 : 4) < a.n ? need + (a.k / 8 > 4 ? a.k / 8 : 4) : a.n;
    if (s_over || total < want || total > a.p2max) {
        if (tid == 0) atomicOr(&a.tile_fail[q / a.fail_queries], 1);
        return;
    }
    lap(0);
    scan_gather_entries(keys, offs, a.list, q, nl, a.caph, a.tps, tid, blockDim.x);
    __syncthreads();
    lap(1);
    const int P = next_pow2(total < 2 ? 2 : total);
    if (a.prof && tid == 0) {
        atomicAdd(&a.prof[8], (unsigned long long)total);
        atomicAdd(&a.prof[9], (unsigned long long)P);
        atomicAdd(&a.prof[10], 1ull);
    }
    // exact distances: 16 lanes per row, 4 rows per wave and pass, 4 passes requested together -- 64 rows of the query
    // in flight per workgroup round (the kernel is a chain of memory round trips).  Bit-identical to
    // wave_exact_distance_f32 (the adaptive path's re-rank; a query must get the same floats whichever path served it):
    // lane `sub` plays that function's lanes sub, sub+16, sub+32, sub+48 (dimensions v and v + 64 each), adds them in
    // the order of its xor-32 and xor-16 steps, and the xor 8 / 4 / 2 / 1 steps run across the 16 lanes.
    const float* qq = a.queries + (size_t)q * a.ldb;
    const int sub = lane & 15, rg = lane >> 4;
    if (a.dim <= 128) {
        // (rows of up to 128 dimensions: every load of a round issued up front -- the shape this kernel was tuned in)
        float qv[8];
        bool ok[8];
    #pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int d = sub + 16 * c;          // c = 0..3: dimension of virtual lane sub + 16c; c = 4..7: the same + 64
            ok[c] = d < a.dim;
            qv[c] = ok[c] ? qq[d] : 0.f;
        }
        // (four passes of 4 rows per wave requested together: 64 rows of the query in flight per workgroup round; eight
        //  passes -- two round trips instead of three at ~150 listed rows -- measured no faster: 35 vs 31 us for this phase)
        constexpr int kU = 4;
        for (int j0 = wave * (4 * kU); j0 < total; j0 += 16 * kU) {
            uint32_t pos[kU];
            float xv[kU][8];
    #pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int j = j0 + 4 * u + rg;
                pos[u] = (uint32_t)keys[j < total ? j : total - 1];
                const float* row = a.base + (size_t)pos[u] * a.ldb + sub;
    #pragma unroll
                for (int c = 0; c < 8; ++c) xv[u][c] = ok[c] ? row[16 * c] : 0.f;
            }
            __builtin_amdgcn_wave_barrier();
    #pragma unroll
            for (int u = 0; u < kU; ++u) {
                float p0[4], p1[4], p2[4];   // per virtual lane: the fma chain over dimensions v, v + 64
    #pragma unroll
                for (int v = 0; v < 4; ++v) {
                    p0[v] = p1[v] = p2[v] = 0.f;
    #pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        if (64 * c >= a.dim) continue;    // (wave_exact_distance_f32 stops at the last 64-chunk)
                        const float x = xv[u][v + 4 * c], y = qv[v + 4 * c];
                        if (a.space == SP_L2) {
                            const float t = x - y;
                            p0[v] = fmaf(t, t, p0[v]);
                        } else {
                            p0[v] = fmaf(x, y, p0[v]);
                            if (a.space != SP_NEGDOT) {
                                p1[v] = fmaf(x, x, p1[v]);
                                p2[v] = fmaf(y, y, p2[v]);
                            }
                        }
                    }
                }
                auto tree = [&](const float* p) __attribute__((always_inline)) -> float {
                    float s = (p[0] + p[2]) + (p[1] + p[3]);     // xor 32, then xor 16
    #pragma unroll
                    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                    return s;
                };
                const float s0 = tree(p0);
                float d;
                if (a.space == SP_L2) d = sqrtf(s0);
                else if (a.space == SP_NEGDOT) d = -s0;
                else {
                    const float sim = normdot_finish(s0, tree(p1), tree(p2));
                    d = a.space == SP_ANGULAR ? acosf(sim) : fmaxf(0.0f, 1.0f - sim);
                }
                const int j = j0 + 4 * u + rg;
                if (sub == 0 && j < total) keys[j] = ((u64)f32_ord(d) << 32) | pos[u];
            }
        }
    } else {
        // lane `sub` of a row's 16 lanes plays the virtual lanes v = sub, sub + 16, sub + 32, sub + 48 of
        // wave_exact_distance_f32: virtual lane v runs ONE fma chain over the dimensions v, v + 64, v + 128, ... (any row
        // length since round 3: two 64-dimension steps per pass, the chains carried across the passes)
        const int nstep = (a.dim + 63) >> 6;
        for (int j0 = wave * 16; j0 < total; j0 += 64) {
            uint32_t pos[4];
            const float* rowp[4];
    #pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + 4 * u + rg;
                pos[u] = (uint32_t)keys[j < total ? j : total - 1];
                rowp[u] = a.base + (size_t)pos[u] * a.ldb + sub;
            }
            float p0[4][4], p1[4][4], p2[4][4];   // [row][virtual lane]
    #pragma unroll
            for (int u = 0; u < 4; ++u)
    #pragma unroll
                for (int v = 0; v < 4; ++v) p0[u][v] = p1[u][v] = p2[u][v] = 0.f;
            for (int c0 = 0; c0 < nstep; c0 += 2) {
                float xv[4][8], qv[8];
                bool ok[8];
    #pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int d = sub + 16 * (c & 3) + 64 * (c0 + (c >> 2));   // c = 0..3: step c0, c = 4..7: step c0 + 1
                    ok[c] = d < a.dim;
                    qv[c] = ok[c] ? qq[d] : 0.f;
    #pragma unroll
                    for (int u = 0; u < 4; ++u) xv[u][c] = ok[c] ? rowp[u][16 * (c & 3) + 64 * (c0 + (c >> 2))] : 0.f;
                }
    #pragma unroll
                for (int u = 0; u < 4; ++u) {
    #pragma unroll
                    for (int v = 0; v < 4; ++v) {
    #pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            if (64 * (c0 + c) >= a.dim) continue;    // (wave_exact_distance_f32 stops at the last 64-chunk)
                            const float x = xv[u][v + 4 * c], y = qv[v + 4 * c];
                            if (a.space == SP_L2) {
                                const float t = x - y;
                                p0[u][v] = fmaf(t, t, p0[u][v]);
                            } else {
                                p0[u][v] = fmaf(x, y, p0[u][v]);
                                if (a.space != SP_NEGDOT) {
                                    p1[u][v] = fmaf(x, x, p1[u][v]);
                                    p2[u][v] = fmaf(y, y, p2[u][v]);
                                }
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
    #pragma unroll
            for (int u = 0; u < 4; ++u) {
                auto tree = [&](const float* p) __attribute__((always_inline)) -> float {
                    float s = (p[0] + p[2]) + (p[1] + p[3]);     // xor 32, then xor 16
    #pragma unroll
                    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                    return s;
                };
                const float s0 = tree(p0[u]);
                float d;
                if (a.space == SP_L2) d = sqrtf(s0);
                else if (a.space == SP_NEGDOT) d = -s0;
                else {
                    const float sim = normdot_finish(s0, tree(p1[u]), tree(p2[u]));
                    d = a.space == SP_ANGULAR ? acosf(sim) : fmaxf(0.0f, 1.0f - sim);
                }
                const int j = j0 + 4 * u + rg;
                if (sub == 0 && j < total) keys[j] = ((u64)f32_ord(d) << 32) | pos[u];
            }
        }
    }
    __syncthreads();
    lap(2);
    if (total <= 512 && a.k <= 32) {
        // the first k of the order without sorting everything: every wave sorts a quarter (128 keys, two per lane, in
        // registers), the four heads of 32 meet in LDS and wave 0 sorts those -- two barriers instead of the 45 of the
        // workgroup-wide bitonic sort of 512 keys
        __shared__ u64 s_head[128];
        const int e0 = wave * 128 + 2 * lane;
        u64 k0 = e0 < total ? keys[e0] : ~0ull, k1 = e0 + 1 < total ? keys[e0 + 1] : ~0ull;
        wave_sort128_u64(k0, k1, lane);
        if (lane < 16) {
            s_head[wave * 32 + 2 * lane] = k0;
            s_head[wave * 32 + 2 * lane + 1] = k1;
        }
        __syncthreads();
        if (wave == 0) {
            k0 = s_head[2 * lane];
            k1 = s_head[2 * lane + 1];
            wave_sort128_u64(k0, k1, lane);
            if (lane < 16) {
                keys[2 * lane] = k0;
                keys[2 * lane + 1] = k1;
            }
        }
        __syncthreads();
    } else {
        for (int i = total + tid; i < P; i += blockDim.x) keys[i] = ~0ull;
        __syncthreads();
        block_bitonic_u64_asc(keys, P, tid, blockDim.x);
    }
    lap(3);
    const int found = total < a.k ? total : a.k;
    // PROOF that no unlisted row belongs to the top k.  The selection score of every row carries an error of at most
    // E = 2^-14 |q||b|: bf16 rounds to 8 significant bits (relative error <= 2^-8), so the dropped lo.lo product and the
    // residues of the two splits (x - hi - lo) are each <= 2^-16 |q||b| (Cauchy-Schwarz over the row), their sum
    // <= 3 * 2^-16, plus the f32 accumulation of the MFMAs (~2^-19).  Unlisted rows scored below the threshold T, so
    // their exact score is below T + E; if the exact score S_k of the k-th result is at least that, every unlisted row
    // is strictly farther than the k-th result.  Otherwise the adaptive kernel redoes the query's tile group.
    if (total < a.n) {
        const bool split_product = s_split != 0;
        if (tid == 0) {
            const float qn2 = s_qn2, qn = sqrtf(qn2);
            const float dk = ord_f32((uint32_t)(keys[found - 1] >> 32));
            float sk, e, extra = 0.f;
            if (a.qaux) {
                // centred cosine / angular: the score is -(1 - cos)|q|, the inner product of the augmented vectors (qn, bmax:
                // theirs).  Beyond the scan's error: the f32 rounding of the augmented columns (2^-22 of the sum of the
                // products' magnitudes) and of the reference formula itself -- its similarity carries a few ulp of 1.0, and the
                // proof is about the distances the reference computes: 2e-6 |q| in score units.
                const float sh = sinf(0.5f * dk);
                const float omc = a.space == SP_ANGULAR ? 2.0f * sh * sh : dk;
                const float qnorm = a.qaux[(size_t)q * 4 + 2];
                sk = -omc * qnorm;
                e = 6.1036e-5f * qn * a.bmax;
                extra = 1e-6f * qn * a.bmax + 2e-6f * qnorm;
            } else if (a.space == SP_L2) {
                sk = 0.5f * (qn2 - dk * dk);
                e = 6.1036e-5f * qn * a.bmax;
            } else if (a.space == SP_NEGDOT) {
                sk = -dk;
                e = 6.1036e-5f * qn * a.bmax;
            } else {  // cosine / angular: score = q.b / |b| = similarity * |q|
                sk = (a.space == SP_ANGULAR ? cosf(dk) : 1.0f - dk) * qn;
                e = 6.1036e-5f * qn;
            }
            const float t = s_thr;
            if (!split_product) {   // one fp16 product: threshold, bound and score in the scan's units (scale^2)
                const float unit = a.scale * a.scale_q;
                sk *= unit;
                extra *= unit;
                e = s_e1;
            }
            e += extra + 1e-6f * (fabsf(sk) + fabsf(t));  // rounding of sk itself
            if (!(sk - t >= e)) atomicOr(&a.tile_fail[q / a.fail_queries], 1);
        }
    }
    lap(4);
    for (int i = tid; i < a.k; i += blockDim.x) {
        int32_t id = -1;
        float d = INFINITY;
        if (i < found) {
            const u64 key = keys[i];
            const uint32_t pos = (uint32_t)key;
            id = a.ext_ids ? a.ext_ids[pos] : (int32_t)pos;
            d = ord_f32((uint32_t)(key >> 32));
        }
        a.out_ids[(size_t)q * a.k + i] = id;
        a.out_dists[(size_t)q * a.k + i] = d;
    }
    if (tid == 0 && a.out_cnt) a.out_cnt[q] = found;
    lap(5);
}

// largest row norm -> out[0]; largest bf16 rounding residual |b - bf16(b)| -> out[1] (relative to |b| if `relative`: the
// cosine score divides by it); atomicMax on the bits of non-negative floats
// out[2] = largest |element| (the fp16 scale is chosen from it); out[3] = largest fp16 residual |scale b - fp16(scale b)| of
// the rows times `scale16` (0: not wanted) -- relative to |b| (i.e. scale16 * the relative residual) if `relative`
__global__ void row_maxnorm_kernel(const float* rows, int n, int ld, int dim, int relative, unsigned* out_bits, float scale16) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    float mb = 0.f, mr = 0.f, ma = 0.f, mh = 0.f;   // this wave's maxima over its rows: one set of atomics per wave
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < n; row += gridDim.x * wpb) {
        const float* p = rows + (size_t)row * ld;
        float s = 0.f, r = 0.f, rh = 0.f;
        for (int d = lane; d < dim; d += 64) {
            const float v = p[d], w = v - (float)(__bf16)v;
            const float vs = scale16 * v, wh = vs - (float)(_Float16)vs;
            s = fmaf(v, v, s);
            r = fmaf(w, w, r);
            rh = fmaf(wh, wh, rh);
            ma = fmaxf(ma, fabsf(v));
        }
        s = wave_sum(s);
        r = wave_sum(r);
        rh = wave_sum(rh);
        const float nb = sqrtf(s), nr = sqrtf(r), nh = sqrtf(rh);
        mb = fmaxf(mb, nb);
        mr = fmaxf(mr, relative ? (nb > 0.f ? nr / nb : 0.f) : nr);
        mh = fmaxf(mh, relative ? (nb > 0.f ? nh / nb : 0.f) : nh);
    }
    ma = wave_max(ma);
    if (lane == 0) {
        atomicMax(out_bits, __float_as_uint(mb));
        atomicMax(out_bits + 1, __float_as_uint(mr));
        atomicMax(out_bits + 2, __float_as_uint(ma));
        atomicMax(out_bits + 3, __float_as_uint(mh));
    }
}

// ---------------------------------------------------------------------------------------
// Re-rank: one workgroup per query.  Exact reference-formula distance of every survivor,
// 64-bit keys (distance, position) sorted ascending, first k emitted.
// ---------------------------------------------------------------------------------------
struct RerankArgs {
    const void* base;
    const void* queries;
    const u64* cand;
    const int* cand_cnt;
    const int32_t* ext_ids;
    int32_t* out_ids;
    float* out_dists;
    int32_t* out_cnt;
    int space, dim, ldb, k, nsplit, cap, kprime, p2max;
    const int* tile_fail;  // u8 fast path fallback: only queries of flagged groups are redone
    int fail_queries;
    // l2 verification (bf_select_f32_kernel<BF_L2> scores rows by -0.5|b|^2 + q.b: cancellation): see the proof below
    int* verify_flags;         // [ceil(nq / verify_queries)] or null
    int verify_queries;
    const float* queries_sel;  // the queries the selection saw
    float bmax;                // largest norm of the selection rows
    float eps_rel;             // error of a selection score relative to |q||b|max + |b|max^2 / 2
};

__global__ __launch_bounds__(256) void bf_rerank_kernel(RerankArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);                      // [p2max]
    int* offs = reinterpret_cast<int*>(keys + a.p2max);            // [nsplit + 1]
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.tile_fail && a.tile_fail[q / a.fail_queries] == 0) return;

    if (tid == 0) {
        int o = 0;
        for (int s = 0; s < a.nsplit; ++s) {
            offs[s] = o;
            int c = a.cand_cnt[(size_t)q * a.nsplit + s];
            o += c < a.kprime ? c : a.kprime;
        }
        offs[a.nsplit] = o;
    }
    __syncthreads();
    const int total = offs[a.nsplit];
    // gather survivor positions (low word of keys[] used as a temporary list)
    __shared__ uint32_t s_kth;   // verification: the k'-th best SELECTION score among the survivors (ordered bits)
    const bool verify = a.verify_flags != nullptr && a.space == SP_L2 && total >= a.kprime;
    if (tid == 0) s_kth = 0xffffffffu;
    for (int idx = tid; idx < a.nsplit * a.kprime; idx += blockDim.x) {
        const int s = idx / a.kprime, i = idx - s * a.kprime;
        const int c = offs[s + 1] - offs[s];
        if (i < c) keys[offs[s] + i] = a.cand[((size_t)q * a.nsplit + s) * a.cap + i];
    }
    __syncthreads();
    if (verify) {
        // every unlisted row scored at most K' = the k'-th best selection score of the survivors: a split that is full
        // holds k' survivors at or above its own cut, and the shared threshold never exceeds the k'-th best overall
        for (int i = tid; i < total; i += blockDim.x) {
            const uint32_t si = (uint32_t)(keys[i] >> 32);
            int above = 0;
            for (int j = 0; j < total; ++j) above += (uint32_t)(keys[j] >> 32) > si;
            if (above < a.kprime) atomicMin(&s_kth, si);
        }
    }
    __syncthreads();
    for (int i = tid; i < total; i += blockDim.x) keys[i] = (u64)sel_key_pos(keys[i]);
    __syncthreads();
    const int P = next_pow2(total < 2 ? 2 : total);
    if (a.space != SP_L2SQR_SIFT && a.dim <= 256) {
        // four survivors per wave and step, their rows requested together (see wave_exact_distance_f32_x4)
        const float* qq = reinterpret_cast<const float*>(a.queries) + (size_t)q * a.ldb;
        for (int j0 = wave * 4; j0 < total; j0 += 16) {
            uint32_t pos[4];
            const float* rows[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0 + r < total ? j0 + r : total - 1;
                pos[r] = (uint32_t)keys[j];
                rows[r] = reinterpret_cast<const float*>(a.base) + (size_t)pos[r] * a.ldb;
            }
            float d[4];
            wave_exact_distance_f32_x4(a.space, rows, qq, a.dim, lane, d);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (lane == 0 && j0 + r < total) keys[j0 + r] = ((u64)f32_ord(d[r]) << 32) | pos[r];
        }
    } else {
        for (int j = wave; j < total; j += 4) {
            const uint32_t pos = (uint32_t)keys[j];
            u64 key;
            if (a.space == SP_L2SQR_SIFT) {
                const uint8_t* row = reinterpret_cast<const uint8_t*>(a.base) + (size_t)pos * 128;
                const uint8_t* qq = reinterpret_cast<const uint8_t*>(a.queries) + (size_t)q * 128;
                const int d = wave_exact_distance_u8(row, qq, lane);
                key = ((u64)i32_ord(d) << 32) | pos;
            } else {
                const float* row = reinterpret_cast<const float*>(a.base) + (size_t)pos * a.ldb;
                const float* qq = reinterpret_cast<const float*>(a.queries) + (size_t)q * a.ldb;
                const float d = wave_exact_distance_f32(a.space, row, qq, a.dim, lane);
                key = ((u64)f32_ord(d) << 32) | pos;
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) keys[j] = key;
        }
    }
    for (int i = total + tid; i < P; i += blockDim.x) keys[i] = ~0ull;
    __syncthreads();
    block_bitonic_u64_asc(keys, P, tid, blockDim.x);
    const int found = total < a.k ? total : a.k;
    for (int i = tid; i < a.k; i += blockDim.x) {
        int32_t id = -1;
        float d = INFINITY;
        if (i < found) {
            const u64 key = keys[i];
            const uint32_t pos = (uint32_t)key;
            id = a.ext_ids ? a.ext_ids[pos] : (int32_t)pos;
            d = (a.space == SP_L2SQR_SIFT) ? (float)ord_i32((uint32_t)(key >> 32))
                                           : ord_f32((uint32_t)(key >> 32));
        }
        a.out_ids[(size_t)q * a.k + i] = id;
        a.out_dists[(size_t)q * a.k + i] = d;
    }
    if (tid == 0 && a.out_cnt) a.out_cnt[q] = found;
    if (verify && found > 0) {
        // PROOF that no unlisted row belongs to the top k.  The selection scored rows by s^ = -0.5|b|^2 + q.b in f32 (the
        // MFMA's accumulation): |s^ - s| <= E = eps_rel (|q||b|max + |b|max^2 / 2).  Unlisted rows: s^ <= K', so their
        // exact squared distance |q|^2 - 2s is at least |q|^2 - 2K' - 2E.  If the k-th result's exact squared distance
        // is below that, it is closer than every unlisted row.  Otherwise -- more than k' - k rows within 2E of the k-th,
        // near-duplicates -- the tile is redone with the reference's own form, sum (a-b)^2 (BF_L2D).
        __shared__ float s_qn2v;
        if (tid < 64) {
            const float* qs = a.queries_sel + (size_t)q * a.ldb;
            float ss = 0.f;
            for (int d = tid; d < a.dim; d += 64) ss = fmaf(qs[d], qs[d], ss);
            ss = wave_sum(ss);
            if (tid == 0) s_qn2v = ss;
        }
        __syncthreads();
        if (tid == 0) {
            const float qn2 = s_qn2v, qn = sqrtf(qn2);
            const float kth = ord_f32(s_kth);
            const float dk = ord_f32((uint32_t)(keys[found - 1] >> 32));
            const float e = a.eps_rel * (qn * a.bmax + 0.5f * a.bmax * a.bmax);
            const float floor_d2 = qn2 - 2.0f * kth - 2.0f * e - 2e-6f * (qn2 + fabsf(kth));
            if (!(dk * dk <= floor_d2)) atomicOr(&a.verify_flags[q / a.verify_queries], 1);
        }
    }
}

// ---------------------------------------------------------------------------------------
// Small preparation kernels
// ---------------------------------------------------------------------------------------
__global__ void row_aux_f32_kernel(const float* base, int n, int ldb, int dim, int space, float* aux) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* p = base + (size_t)row * ldb;
    float s = 0.f;
    for (int d = lane; d < dim; d += 64) s = fmaf(p[d], p[d], s);
    s = wave_sum(s);
    if (lane == 0) {
        float v = 0.f;
        if (space == SP_L2) v = -0.5f * s;
        else if (space == SP_COSINE || space == SP_ANGULAR)
            v = (s < 1.17549435e-38f * 2.0f) ? 0.f : 1.0f / sqrtf(s);
        aux[row] = v;
    }
}

// aux[row] = 256*sum(a) - sum(a^2), rows_i8[row] = a ^ 0x80; rows n..n_pad-1: zero bytes, aux = kPadAux
__global__ void prepare_u8_kernel(const uint8_t* base, int n, int n_pad, uint8_t* rows_i8, int32_t* aux, int32_t* auxh) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n_pad) return;
    uint16_t* dst = reinterpret_cast<uint16_t*>(rows_i8 + (size_t)row * 128);
    if (row >= n) {
        dst[lane] = 0;
        if (lane == 0) {
            aux[row] = -(1 << 30);
            if (auxh) auxh[row] = -(1 << 29);
        }
        return;
    }
    const uint8_t* p = base + (size_t)row * 128;
    const int x0 = p[2 * lane], x1 = p[2 * lane + 1];
    dst[lane] = (uint16_t)((x0 | (x1 << 8)) ^ 0x8080);
    const int sum = wave_sum_i(x0 + x1);
    const int sq = wave_sum_i(x0 * x0 + x1 * x1);
    if (lane == 0) {
        aux[row] = 256 * sum - sq;
        if (auxh) auxh[row] = (256 * sum - sq) >> 1;
    }
}

__global__ void pad_rows_kernel(const uint8_t* src, int rows, int row_bytes, uint8_t* dst, int rows_pad,
                                int ld_bytes) {
    const size_t total = (size_t)rows_pad * ld_bytes;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / ld_bytes, c = i - r * ld_bytes;
        dst[i] = (r < (size_t)rows && c < (size_t)row_bytes) ? src[r * row_bytes + c] : (uint8_t)0;
    }
}

// Query preparation of the uint8 fast path in ONE launch (round 3; was pad_rows + two fillBuffers): the padded queries and
// the words later kernels of the batch expect cleared (tile flags, the fallback selection's shared thresholds)
__global__ void bf_u8_prep_kernel(const uint8_t* raw, int nq, uint8_t* padded, int qpad, int* clr0, int n0, uint32_t* clr1,
                                  size_t n1) {
    const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (size_t)gridDim.x * blockDim.x;
    if ((reinterpret_cast<uintptr_t>(raw) & 3) == 0) {   // 128-byte rows as 32-bit words
        const size_t words = (size_t)qpad * 32;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(raw);
        uint32_t* dst = reinterpret_cast<uint32_t*>(padded);
        for (size_t i = gtid; i < words; i += gsz) dst[i] = i < (size_t)nq * 32 ? src[i] : 0u;
    } else {                                             // (a caller's buffer at an odd address)
        const size_t bytes = (size_t)qpad * 128;
        for (size_t i = gtid; i < bytes; i += gsz) padded[i] = i < (size_t)nq * 128 ? raw[i] : (uint8_t)0;
    }
    for (size_t i = gtid; i < (size_t)n0; i += gsz) clr0[i] = 0;
    for (size_t i = gtid; i < n1; i += gsz) clr1[i] = 0u;
}

// hnsw.h:486-497 NormalizeVect: v *= 1/sqrt(sum v^2) unless the sum is exactly 0
__global__ void normalize_rows_kernel(float* rows, int n, int ld, int dim) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    float* p = rows + (size_t)row * ld;
    float s = 0.f;
    for (int d = lane; d < dim; d += 64) s = fmaf(p[d], p[d], s);
    s = wave_sum(s);
    if (s != 0.0f) {
        const float inv = 1.0f / sqrtf(s);
        for (int d = lane; d < dim; d += 64) p[d] *= inv;
    }
}

// Column sums of the stored rows (f64 accumulators): stats[c] = sum_r base[r][c], stats[ldb] = sum of squares of everything.
// One-off at finalize: decides whether the L2 selection runs on a centred copy (see launch_center_rows).
__global__ void col_stats_kernel(const float* base, int n, int ldb, int dim, int rows_per_block, double* stats) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(n, r0 + rows_per_block);
    double sq = 0.0;
    for (int c = tx; c < dim; c += 64) {
        double s = 0.0;
        for (int r = r0 + ty; r < r1; r += 4) {
            const double v = (double)base[(size_t)r * ldb + c];
            s += v;
            sq += v * v;
        }
        atomicAdd(&stats[c], s);
    }
    atomicAdd(&stats[ldb], sq);
}

// dst[r][c] = src[r][c] - mean[c] for c < dim (pad columns stay 0); rows >= rows_valid are copied unchanged
__global__ void center_rows_kernel(const float* src, const float* mean, int rows, int rows_valid, int ld, int dim, float* dst) {
    const size_t total = (size_t)rows * ld;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / ld;
        const int c = (int)(i - r * ld);
        const float v = src[i];
        dst[i] = (c < dim && r < (size_t)rows_valid) ? v - mean[c] : v;
    }
}

// BF_COSC preparation.  Rows: aux[0][r] = -|b'|^2, aux[1][r] = |b| - |mu|, aux[2][r] = 1/|b| (0 for a zero-norm row: the
// reference's rule norm^2 < 2*FLT_MIN -> similarity 0).  Norms are accumulated in f64: |b| - |mu| is a small
// difference of large numbers.
__global__ void row_aux_cosc_kernel(const float* orig, const float* centred, int n, int ldb, int dim, double mu_norm,
                                    float* aux) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* p = orig + (size_t)row * ldb;
    const float* c = centred + (size_t)row * ldb;
    double nb2 = 0.0;
    float a = 0.f, nf = 0.f;
    for (int d = lane; d < dim; d += 64) {
        const float x = p[d], y = c[d];
        nb2 += (double)x * (double)x;
        nf = fmaf(x, x, nf);
        a = fmaf(y, y, a);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nb2 += __shfl_xor(nb2, o, 64);
    a = wave_sum(a);
    nf = wave_sum(nf);
    if (lane == 0) {
        const double nb = sqrt(nb2);
        aux[row] = -a;
        aux[(size_t)n + row] = (float)(nb - mu_norm);
        aux[2 * (size_t)n + row] = (nf < 1.17549435e-38f * 2.0f) ? 0.f : (float)(1.0 / nb);
    }
}
// Queries: qaux[q] = {|q'|^2, |q| - |mu|, |q|, 1 or 0 (zero-norm query: every distance is 1, positions decide)}
__global__ void query_aux_cosc_kernel(const float* orig, const float* centred, int nq, int ldb, int dim, double mu_norm,
                                      float* qaux) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (q >= nq) return;
    const float* p = orig + (size_t)q * ldb;
    const float* c = centred + (size_t)q * ldb;
    double n2 = 0.0;
    float a = 0.f, nf = 0.f;
    for (int d = lane; d < dim; d += 64) {
        const float x = p[d], y = c[d];
        n2 += (double)x * (double)x;
        nf = fmaf(x, x, nf);
        a = fmaf(y, y, a);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n2 += __shfl_xor(n2, o, 64);
    a = wave_sum(a);
    nf = wave_sum(nf);
    if (lane == 0) {
        const double nn = sqrt(n2);
        const bool zero = nf < 1.17549435e-38f * 2.0f;
        qaux[(size_t)q * 4 + 0] = a;
        qaux[(size_t)q * 4 + 1] = (float)(nn - mu_norm);
        qaux[(size_t)q * 4 + 2] = (float)nn;
        qaux[(size_t)q * 4 + 3] = zero ? 0.f : 1.f;
    }
}

// Centred cosine / angular on the bf16 fast path (round 3).  The BF_COSC score -(1 - cos)|q| (see bf_select_f32_kernel)
// is an INNER PRODUCT of augmented vectors -- three more columns behind the centred ones:
//     -(1 - cos(q,b)) |q| = [ q'.b' - dq db - (|b'|^2 - db^2)/2 - (|q'|^2 - dq^2)/2 ] / |b|,   dq = |q| - |mu|, db = |b| - |mu|
//     q+ = ( q',  -dq,  lambda,          -cq / lambda ),      cq = (|q'|^2 - dq^2) / 2
//     b+ = ( b',   db,  -cb / lambda,     lambda      ) / |b|, cb = (|b'|^2 - db^2) / 2
// so the one-product / split-product scans run it in their inner-product mode (no start values) over rows of dim + 3
// columns.  lambda ~ sqrt(cb) balances the two constant columns: the error bounds are Cauchy-Schwarz bounds over the whole
// vector, and a column pair (1, c) with c ~ 10^4 would inflate them by that factor.  Norms in f64 as in row_aux_cosc_kernel.
__global__ void row_aug_cosc_kernel(const float* orig, const float* centred, int n, int ldb, int dim, double mu_norm,
                                    float lambda, float* out, int ldo, int* zero_rows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* p = orig + (size_t)row * ldb;
    const float* c = centred + (size_t)row * ldb;
    double nb2 = 0.0, a = 0.0;
    float nf = 0.f;
    for (int d = lane; d < dim; d += 64) {
        const float x = p[d], y = c[d];
        nb2 += (double)x * (double)x;
        a += (double)y * (double)y;
        nf = fmaf(x, x, nf);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        nb2 += __shfl_xor(nb2, o, 64);
        a += __shfl_xor(a, o, 64);
    }
    nf = wave_sum(nf);
    const bool zero = nf < 1.17549435e-38f * 2.0f;   // (the test of row_aux_cosc_kernel)
    const double nb = sqrt(nb2), db = nb - mu_norm, cb = 0.5 * (a - db * db), inv = zero ? 0.0 : 1.0 / nb;
    float* o = out + (size_t)row * ldo;
    for (int d = lane; d < ldo; d += 64) {
        double v = 0.0;
        if (d < dim) v = (double)c[d] * inv;
        else if (d == dim) v = db * inv;
        else if (d == dim + 1) v = -cb / (double)lambda * inv;
        else if (d == dim + 2) v = (double)lambda * inv;
        o[d] = (float)v;
    }
    if (zero && lane == 0) atomicOr(zero_rows, 1);   // (no score of this form for a zero row: the caller keeps the adaptive path)
}
// q+ of every (padded) query from its centred copy and qaux = {|q'|^2, |q| - |mu|, |q|, flag} (query_aux_cosc_kernel)
__global__ void query_aug_cosc_kernel(const float* centred, const float* qaux, int nq, int qpad, int ldb, int dim, float lambda,
                                      float* out, int ldo) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (q >= qpad) return;
    float* o = out + (size_t)q * ldo;
    const bool real = q < nq;
    const double a = real ? (double)qaux[(size_t)q * 4 + 0] : 0.0, dq = real ? (double)qaux[(size_t)q * 4 + 1] : 0.0;
    const double cq = 0.5 * (a - dq * dq);
    for (int d = lane; d < ldo; d += 64) {
        float v = 0.f;
        if (real) {
            if (d < dim) v = centred[(size_t)q * ldb + d];
            else if (d == dim) v = (float)-dq;
            else if (d == dim + 1) v = lambda;
            else if (d == dim + 2) v = (float)(-cq / (double)lambda);
        }
        o[d] = v;
    }
}

__global__ void pair_distance_kernel(int space, const void* a, const void* b, int dim, float* out) {
    const int lane = threadIdx.x & 63;
    float d;
    if (space == SP_L2SQR_SIFT)
        d = (float)wave_exact_distance_u8((const uint8_t*)a, (const uint8_t*)b, lane);
    else
        d = wave_exact_distance_f32(space, (const float*)a, (const float*)b, dim, lane);
    if (lane == 0) *out = d;
}

// per-shard top-k lists -> global top-k by (distance, id); one wave per query
__global__ void merge_topk_kernel(const float* dists_in, const int32_t* ids_in, size_t shard_stride, int nshards,
                                  int nq, int k, float* dists_out, int32_t* ids_out, int32_t* cnt_out,
                                  const int32_t* ext_ids) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);
    const int q = blockIdx.x, tid = threadIdx.x;
    const int total = nshards * k;
    const int P = next_pow2(total < 2 ? 2 : total);
    for (int i = tid; i < P; i += blockDim.x) {
        u64 key = ~0ull;
        if (i < total) {
            const int s = i / k, j = i - s * k;
            const size_t off = (size_t)s * shard_stride + (size_t)q * k + j;
            const int32_t id = ids_in[off];
            if (id >= 0) key = ((u64)f32_ord(dists_in[off]) << 32) | (uint32_t)id;
        }
        keys[i] = key;
    }
    __syncthreads();
    block_bitonic_u64_asc(keys, P, tid, blockDim.x);
    for (int i = tid; i < k; i += blockDim.x) {
        const u64 key = keys[i];
        const bool ok = key != ~0ull;
        const int32_t id = (int32_t)(uint32_t)key;
        ids_out[(size_t)q * k + i] = ok ? (ext_ids ? ext_ids[id] : id) : -1;
        dists_out[(size_t)q * k + i] = ok ? ord_f32((uint32_t)(key >> 32)) : INFINITY;
    }
    if (cnt_out && tid == 0) {
        int c = 0;
        while (c < k && keys[c] != ~0ull) ++c;
        cnt_out[q] = c;
    }
}

// The same merge when nshards * k keys do not fit LDS (k in the thousands): every per-shard list is already ascending
// in (distance, id), so an item's place in the merged order is its own index plus, for every other list, the number
// of keys below it there (bisection; keys are unique because ids are).  Items ranked below k write themselves.
__global__ void merge_topk_big_kernel(const float* dists_in, const int32_t* ids_in, size_t shard_stride, int nshards,
                                      int nq, int k, float* dists_out, int32_t* ids_out, int32_t* cnt_out,
                                      const int32_t* ext_ids) {
    const int q = blockIdx.x;
    auto key_of = [&](int s, int j) -> u64 {
        const size_t off = (size_t)s * shard_stride + (size_t)q * k + j;
        const int32_t id = ids_in[off];
        return id >= 0 ? (((u64)f32_ord(dists_in[off]) << 32) | (uint32_t)id) : ~0ull;
    };
    for (int i = threadIdx.x; i < k; i += blockDim.x) {  // (slots no list reaches stay "no result")
        ids_out[(size_t)q * k + i] = -1;
        dists_out[(size_t)q * k + i] = INFINITY;
    }
    __syncthreads();
    int valid = 0;
    for (int i = threadIdx.x; i < nshards * k; i += blockDim.x) {
        const int s = i / k, j = i - s * k;
        const u64 key = key_of(s, j);
        if (key == ~0ull) continue;
        valid++;
        int rank = j;
        for (int t = 0; t < nshards && rank < k; ++t) {
            if (t == s) continue;
            int lo = 0, hi = k;
            while (lo < hi) {  // number of keys of list t below `key` (missing entries compare as +inf)
                const int mid = (lo + hi) >> 1;
                if (key_of(t, mid) < key) lo = mid + 1;
                else hi = mid;
            }
            rank += lo;
        }
        if (rank < k) {
            const int32_t id = (int32_t)(uint32_t)key;
            ids_out[(size_t)q * k + rank] = ext_ids ? ext_ids[id] : id;
            dists_out[(size_t)q * k + rank] = ord_f32((uint32_t)(key >> 32));
        }
    }
    if (cnt_out) {
        __shared__ int total;
        if (threadIdx.x == 0) total = 0;
        __syncthreads();
        atomicAdd(&total, valid);
        __syncthreads();
        if (threadIdx.x == 0) cnt_out[q] = total < k ? total : k;
    }
}

// ---------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------
static int host_next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

BfPlan bf_make_plan(int n, int dim, int nq, int k, bool is_u8, int qpad_multiple) {
    BfPlan p{};
    p.nq = nq;
    p.qpad = (nq + qpad_multiple - 1) / qpad_multiple * qpad_multiple;
    p.nqt = p.qpad / BF_TQ;
    p.n = n;
    p.ldb = is_u8 ? 128 : f32_row_stride(dim);
    // float scores are ranked in the Q.B^T form, whose rounding differs from the reference's
    // direct formula: keep a few more than k per split and let the exact re-rank decide.
    p.kprime = is_u8 ? k : k + (k / 8 > 4 ? k / 8 : 4);
    p.cap = host_next_pow2(2 * p.kprime + BF_BN);
    if (p.cap < 256) p.cap = 256;
    // splits: fill the chip (>= 512 workgroups of 2/CU), keep >= 2 stages per split, and bound
    // the re-rank sort (nsplit * kprime keys in LDS)
    int want = (512 + p.nqt - 1) / p.nqt;
    int by_rows = (n + 2 * BF_BN - 1) / (2 * BF_BN);
    int by_sort = 4096 / p.kprime;
    int ns = want < by_rows ? want : by_rows;
    if (ns > by_sort) ns = by_sort;
    // one or two query tiles (single-query calls of nmslib_knn_query_fill): latency is the row stream of one
    // workgroup, so cut the rows finer and put a workgroup on every CU
    const int ns_cap = p.nqt <= 2 ? 256 : 64;
    if (ns > ns_cap) ns = ns_cap;
    if (const char* e = getenv("NMSLIB_GPU_SPLITS")) ns = atoi(e);  // tuning experiments
    ns = (ns + 7) / 8 * 8;
    if (ns < 8) ns = 8;
    p.nsplit = ns;
    int rps = (n + ns - 1) / ns;
    p.rows_per_split = (rps + BF_BN - 1) / BF_BN * BF_BN;
    if (p.rows_per_split < BF_BN) p.rows_per_split = BF_BN;
    p.p2max = host_next_pow2(p.nsplit * p.kprime);
    // counted bound: xm splits x 2 lanes x xj rows >= k'; leave a quarter of the splits as slack for laggards
    {
        int m = (p.kprime + 1) / 2;
        if (m > 8) m = 8;
        if (m > p.nsplit * 3 / 4) m = p.nsplit * 3 / 4;
        if (m < 1) m = 1;
        const int jj = (p.kprime + 2 * m - 1) / (2 * m);
        p.xm = m;
        p.xj = jj <= 8 ? jj : 0;
    }
    const int kcs = p.ldb < BF_KC ? p.ldb : BF_KC;
    if (is_u8)
        p.lds_select = 6 * BF_BN * 128 + 8 * BF_BN * 4 + 4 * (size_t)p.cap * 8;  // DMA ring + aux ring + scratch
    else
        p.lds_select = (size_t)2 * BF_BN * (kcs + 4) * 4 + 4 * BF_BN * 3 * 4 + 4 * (size_t)p.cap * 8;  // tiles + aux planes + scratch
    p.lds_rerank = (size_t)p.p2max * 8 + (p.nsplit + 1) * 4 + 16;
    return p;
}

template <int MODE, bool FULL, bool ONE>
static hipError_t launch_select_kern(const BfPlan& p, const BfArgs& a, hipStream_t s) {
    auto kern = bf_select_f32_kernel<MODE, FULL, ONE>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_select);
    if (e != hipSuccess) return e;
    const int grid = 8 * p.nqt * (p.nsplit / 8);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), p.lds_select, s, a);
    if (a.dbg & 1024) {
        long long h[2] = {0, 0};
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bf_clk), sizeof(h));
        fprintf(stderr, "[bf_select] block 0: %lld shader cycles, %.1f us, %.0f MHz\n", h[0], h[1] / 100.0,
                h[1] ? (double)h[0] / (h[1] / 100.0) : 0.0);
        static long long tr[2048][3];
        (void)hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_bf_trace), sizeof(tr));
        const int nb = grid < 2048 ? grid : 2048;
        long long t0 = tr[0][0];
        for (int i = 0; i < nb; ++i) t0 = tr[i][0] < t0 ? tr[i][0] : t0;
        if (getenv("NMSLIB_GPU_TRACE") && grid >= 512) {
            FILE* f = fopen(getenv("NMSLIB_GPU_TRACE"), "w");
            if (f) {
                for (int i = 0; i < nb; ++i)
                    fprintf(f, "%d %.2f %.2f %llx\n", i, (tr[i][0] - t0) / 100.0, (tr[i][1] - t0) / 100.0,
                            (unsigned long long)tr[i][2]);
                fclose(f);
            }
        }
    }
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_select_mode(const BfPlan& p, const BfArgs& a, hipStream_t s) {
#ifdef BF_FAST_COMPILE  // kernel experiments: only the headline instantiation
    if (MODE == BF_L2 && p.ldb == BF_KC) return launch_select_kern<BF_L2, true, true>(p, a, s);
    return hipErrorInvalidValue;
#else
    if constexpr (MODE == BF_L1 || MODE == BF_LINF || MODE == BF_L2D) {
        // VALU-bound modes: one generic instantiation is enough (and keeps the build short)
        return launch_select_kern<MODE, false, false>(p, a, s);
    } else {
        if (p.ldb == BF_KC) return launch_select_kern<MODE, true, true>(p, a, s);
        if (p.ldb % BF_KC == 0) return launch_select_kern<MODE, true, false>(p, a, s);
        return launch_select_kern<MODE, false, false>(p, a, s);
    }
#endif
}

static BfArgs make_args(const BfPlan& p, const float* base, const float* aux, const float* q, u64* cand,
                        int* cnt, uint32_t* gthr) {
    BfArgs a{};
    a.base = base;
    a.aux = aux;
    a.queries = q;
    a.cand = cand;
    a.cand_cnt = cnt;
    a.gthr = gthr;
    a.gq = gthr + p.qpad;
    a.xj = p.xj;
    a.xm = p.xm;
    a.n = p.n;
    a.ldb = p.ldb;
    a.nqt = p.nqt;
    a.nsplit = p.nsplit;
    a.rows_per_split = p.rows_per_split;
    a.kprime = p.kprime;
    a.cap = p.cap;
    a.kcs = p.ldb < BF_KC ? p.ldb : BF_KC;
    a.nchunks = (p.ldb + BF_KC - 1) / BF_KC;
    static const int dbg = getenv("NMSLIB_GPU_DEBUG") ? atoi(getenv("NMSLIB_GPU_DEBUG")) : 0;
    a.dbg = dbg;
    return a;
}

hipError_t launch_bf_select_f32(const BfPlan& p, int space, const float* base, const float* aux,
                                const float* queries_padded, const float* qaux_cosc, unsigned long long* cand,
                                int* cand_cnt, hipStream_t s) {
    return launch_bf_select_f32_ex(p, space, base, aux, queries_padded, qaux_cosc, cand, cand_cnt, nullptr, 1, s);
}

hipError_t launch_bf_select_f32_ex(const BfPlan& p, int space, const float* base, const float* aux,
                                   const float* queries_padded, const float* qaux_cosc, unsigned long long* cand,
                                   int* cand_cnt, const int* tile_fail, int fail_group, hipStream_t s, bool cleared) {
    // per-query shared thresholds live behind the survivor counts; cleared for every batch (by the caller's prep
    // kernel when `cleared`)
    uint32_t* gthr = reinterpret_cast<uint32_t*>(cand_cnt + (size_t)p.qpad * p.nsplit);
    if (!cleared) {
        hipError_t me = hipMemsetAsync(gthr, 0, ((size_t)p.qpad + (size_t)p.qpad * p.nsplit) * 4, s);
        if (me != hipSuccess) return me;
    }
    BfArgs a = make_args(p, base, aux, queries_padded, cand, cand_cnt, gthr);
    a.tile_fail = tile_fail;
    a.fail_group = fail_group;
    switch (space) {
        case SP_L2: return launch_select_mode<BF_L2>(p, a, s);
        case SP_NEGDOT: return launch_select_mode<BF_DOT>(p, a, s);
        case SP_COSINE:
        case SP_ANGULAR:
            if (qaux_cosc) {  // centred rows + three aux planes (engine: centred_)
                a.qaux = qaux_cosc;
                a.aux_stride = p.n;
                return launch_select_mode<BF_COSC>(p, a, s);
            }
            return launch_select_mode<BF_COS>(p, a, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_bf_select_direct_f32(const BfPlan& p, int space, const float* base,
                                       const float* queries_padded, unsigned long long* cand,
                                       int* cand_cnt, hipStream_t s) {
    return launch_bf_select_direct_f32_ex(p, space, base, queries_padded, cand, cand_cnt, nullptr, 1, s);
}

// SP_L2 here = squared differences summed by the VALU on the ORIGINAL rows (the exact tail of the verified l2 path)
hipError_t launch_bf_select_direct_f32_ex(const BfPlan& p, int space, const float* base, const float* queries_padded,
                                          unsigned long long* cand, int* cand_cnt, const int* tile_fail, int fail_group,
                                          hipStream_t s, bool cleared_second_region) {
    const size_t gwords = (size_t)p.qpad + (size_t)p.qpad * p.nsplit;
    uint32_t* gthr = reinterpret_cast<uint32_t*>(cand_cnt + (size_t)p.qpad * p.nsplit);
    if (cleared_second_region) {
        gthr += gwords;   // its own region, cleared at the start of the batch together with the first (bf_f32_prep_kernel)
    } else {
        hipError_t me = hipMemsetAsync(gthr, 0, gwords * 4, s);
        if (me != hipSuccess) return me;
    }
    BfArgs a = make_args(p, base, nullptr, queries_padded, cand, cand_cnt, gthr);
    a.tile_fail = tile_fail;
    a.fail_group = fail_group;
    if (space == SP_L1) return launch_select_mode<BF_L1>(p, a, s);
    if (space == SP_LINF) return launch_select_mode<BF_LINF>(p, a, s);
    if (space == SP_L2) return launch_select_mode<BF_L2D>(p, a, s);
    return hipErrorInvalidValue;
}

hipError_t launch_bf_select_u8(const BfPlan& p, const uint8_t* base_i8, const int32_t* aux,
                               const uint8_t* queries_padded, unsigned long long* cand, int* cand_cnt,
                               hipStream_t s) {
    return launch_bf_select_u8_ex(p, base_i8, aux, queries_padded, cand, cand_cnt, 1, nullptr, 1, s, false);
}

hipError_t launch_bf_select_u8_ex(const BfPlan& p, const uint8_t* base_i8, const int32_t* aux,
                                  const uint8_t* queries_padded, unsigned long long* cand, int* cand_cnt,
                                  int tile_stride, const int* tile_fail, int fail_group, hipStream_t s, bool cleared) {
    BfArgsU8 a{};
    a.tile_stride = tile_stride;
    a.tile_fail = tile_fail;
    a.fail_group = fail_group;
    a.base_i8 = base_i8;
    a.aux = aux;
    a.queries = queries_padded;
    a.cand = cand;
    a.cand_cnt = cand_cnt;
    a.n = p.n;
    a.nqt = p.nqt;
    a.nsplit = p.nsplit;
    a.rows_per_split = p.rows_per_split;
    a.kprime = p.kprime;
    a.cap = p.cap;
    uint32_t* gthr = reinterpret_cast<uint32_t*>(cand_cnt + (size_t)p.qpad * p.nsplit);
    a.gq = gthr + p.qpad;
    a.xj = p.xj;
    a.xm = p.xm;
    static const int dbg = getenv("NMSLIB_GPU_DEBUG") ? atoi(getenv("NMSLIB_GPU_DEBUG")) : 0;
    a.dbg = dbg;
    if (!cleared) {   // (cleared: the caller's preparation kernel zeroed the shared thresholds at the start of the batch)
        hipError_t me = hipMemsetAsync(gthr, 0, ((size_t)p.qpad + (size_t)p.qpad * p.nsplit) * 4, s);
        if (me != hipSuccess) return me;
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bf_select_u8_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_select);
    if (e != hipSuccess) return e;
    const int grid = 8 * p.nqt * (p.nsplit / 8);
    hipLaunchKernelGGL(bf_select_u8_kernel, dim3(grid), dim3(256), p.lds_select, s, a);
    return hipGetLastError();
}

hipError_t launch_bf_rerank(const BfPlan& p, int space, int dim, int k, const void* base,
                            const void* queries_padded, const unsigned long long* cand,
                            const int* cand_cnt, const int32_t* ext_ids, int32_t* out_ids,
                            float* out_dists, int32_t* out_cnt, hipStream_t s) {
    return launch_bf_rerank_ex(p, space, dim, k, base, queries_padded, cand, cand_cnt, ext_ids, out_ids, out_dists,
                               out_cnt, nullptr, 1, s);
}

hipError_t launch_bf_rerank_ex(const BfPlan& p, int space, int dim, int k, const void* base,
                               const void* queries_padded, const unsigned long long* cand,
                               const int* cand_cnt, const int32_t* ext_ids, int32_t* out_ids,
                               float* out_dists, int32_t* out_cnt, const int* tile_fail, int fail_queries,
                               hipStream_t s) {
    return launch_bf_rerank_verify(p, space, dim, k, base, queries_padded, cand, cand_cnt, ext_ids, out_ids, out_dists,
                                   out_cnt, tile_fail, fail_queries, nullptr, nullptr, 0.f, s);
}

hipError_t launch_bf_rerank_verify(const BfPlan& p, int space, int dim, int k, const void* base,
                                   const void* queries_padded, const unsigned long long* cand,
                                   const int* cand_cnt, const int32_t* ext_ids, int32_t* out_ids,
                                   float* out_dists, int32_t* out_cnt, const int* tile_fail, int fail_queries,
                                   int* verify_flags, const float* queries_sel, float bmax, hipStream_t s) {
    RerankArgs a{};
    a.verify_flags = verify_flags;
    a.verify_queries = BF_TQ;
    a.queries_sel = queries_sel;
    a.bmax = bmax;
    a.eps_rel = 1.5f * (float)(dim + 2) * 5.9604645e-8f;   // (D + 2) roundings of 2^-24, half again for the aux term
    a.tile_fail = tile_fail;
    a.fail_queries = fail_queries;
    a.base = base;
    a.queries = queries_padded;
    a.cand = cand;
    a.cand_cnt = cand_cnt;
    a.ext_ids = ext_ids;
    a.out_ids = out_ids;
    a.out_dists = out_dists;
    a.out_cnt = out_cnt;
    a.space = space;
    a.dim = dim;
    a.ldb = p.ldb;
    a.k = k;
    a.nsplit = p.nsplit;
    a.cap = p.cap;
    a.kprime = p.kprime;
    a.p2max = p.p2max;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bf_rerank_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_rerank);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bf_rerank_kernel, dim3(p.nq), dim3(256), p.lds_rerank, s, a);
    return hipGetLastError();
}

// The adaptive f32 selection + re-rank, verified for l2 (see bf_rerank_kernel), with the exact tail: tiles whose proof
// fails are selected again by BF_L2D on the original rows.  `gate` / gate_tiles: run only the query-tile groups flagged
// by an earlier stage (the fast paths' fallback), or null.  flags: [p.nqt] ints of workspace.
hipError_t launch_bf_adaptive_f32(const BfPlan& p, int space, int dim, int k, const float* base_orig, const float* sel_rows,
                                  const float* aux, const float* queries_orig, const float* queries_sel,
                                  const float* qaux_cosc, float bmax, unsigned long long* cand, int* cand_cnt, int* flags,
                                  const int32_t* ext_ids, int32_t* out_ids, float* out_dists, int32_t* out_cnt,
                                  const int* gate, int gate_tiles, hipStream_t s, bool cleared) {
    // cleared: the caller's prep kernel zeroed `flags` and both shared-threshold regions at the start of the batch
    hipError_t e = launch_bf_select_f32_ex(p, space, sel_rows, aux, queries_sel, qaux_cosc, cand, cand_cnt, gate, gate_tiles, s,
                                           cleared);
    if (e != hipSuccess) return e;
    const bool verify = space == SP_L2 && flags != nullptr && bmax > 0.f;
    if (!verify)
        return launch_bf_rerank_ex(p, space, dim, k, base_orig, queries_orig, cand, cand_cnt, ext_ids, out_ids, out_dists,
                                   out_cnt, gate, gate_tiles * BF_TQ, s);
    if (!cleared) {
        e = hipMemsetAsync(flags, 0, (size_t)p.nqt * 4, s);
        if (e != hipSuccess) return e;
    }
    e = launch_bf_rerank_verify(p, space, dim, k, base_orig, queries_orig, cand, cand_cnt, ext_ids, out_ids, out_dists,
                                out_cnt, gate, gate_tiles * BF_TQ, flags, queries_sel, bmax, s);
    if (e != hipSuccess) return e;
    e = launch_bf_select_direct_f32_ex(p, SP_L2, base_orig, queries_orig, cand, cand_cnt, flags, 1, s, cleared);
    if (e != hipSuccess) return e;
    return launch_bf_rerank_ex(p, space, dim, k, base_orig, queries_orig, cand, cand_cnt, ext_ids, out_ids, out_dists,
                               out_cnt, flags, BF_TQ, s);
}

// ---- uint8 fast path (see bf_scan_u8_kernel) ----------------------------------------------------------------
BfU8Fast bf_u8_fast_plan(int n, int nq, int k) {
    BfU8Fast f{};
    f.use = (n >= 65536 && nq >= 512 && k <= 256);
    if (const char* e = getenv("NMSLIB_GPU_U8_FAST")) f.use = f.use && atoi(e) != 0;
    if (!f.use) return f;
    f.qg = nq >= 2048 ? 4 : 2;
    const int tq = BF_TQ * f.qg;
    f.qpad = (nq + tq - 1) / tq * tq;
    f.nqt = f.qpad / tq;
    f.stride = 8;
    if (const char* e = getenv("NMSLIB_GPU_U8_STRIDE")) f.stride = atoi(e) > 0 ? atoi(e) : 8;   // (experiments)
    // r-th best of a 1/stride sample: expected k/stride rows of the true top-k fall into the sample; r sits
    // ~3 sigma above that (+ slack), so that fewer than k rows reaching the threshold is a ~1e-6 event per query
    const double kf = (double)k / f.stride;
    f.r = (int)(1.6 * kf + 3.0 * sqrt(kf) + 3.0 + 0.999);
    const int tiles_all = (n + BF_BN - 1) / BF_BN;
    int ns = (512 + f.nqt - 1) / f.nqt;
    if (ns > tiles_all / 16) ns = tiles_all / 16;  // at least 16 stages per split
    if (ns > 256) ns = 256;
    ns = (ns + 7) / 8 * 8;
    if (ns < 8) ns = 8;
    while ((tiles_all + ns - 1) / ns > 32768) ns += 8;   // list entries hold the 32-row block index inside the split in 16 bits
    f.nsplit = ns;
    f.tps = (tiles_all + ns - 1) / ns;
    const double mean_half = (double)f.r * f.stride / (2.0 * ns);
    int caph = 8;
    while (caph < 4.0 * mean_half + 8.0) caph <<= 1;
    f.caph = caph;
    f.p2max = host_next_pow2(2 * ns * caph);
    f.lds_scan = 8 * BF_BN * 128 + 8 * BF_BN * 4 + 64;
    f.lds_rerank = (size_t)f.p2max * 8 + (2 * (size_t)ns + 1) * 4 + 16;
    // sample pass = the same streaming kernel over every stride-th tile, per-lane top-8 lists instead of thresholds
    // (in the scan's own shape -- 512 queries per workgroup at large batches: half the L2 -> LDS stream of the
    //  256-query shape it first ran in)
    const int stiles = (tiles_all + f.stride - 1) / f.stride;
    const int s_nqt = f.qpad / (BF_TQ * f.qg);   // (the scan's own shape: same workgroups per query tile)
    int nss = (256 + s_nqt - 1) / s_nqt;
    if (nss > stiles / 32) nss = stiles / 32;
    if (nss > 64) nss = 64;
    nss = (nss + 7) / 8 * 8;
    if (nss < 8) nss = 8;
    f.s_nsplit = nss;
    f.s_tps = (stiles + nss - 1) / nss;
    f.lds_thr = (size_t)host_next_pow2(nss * 2 * 8) * 8 + 16;
    // fallback = the adaptive kernel over everything, same query padding
    f.fallback = bf_make_plan(n, 128, nq, k, true, tq);
    return f;
}

hipError_t launch_bf_u8_fast(const BfU8Fast& f, int n, int nq, int k, const uint8_t* base_u8, const uint8_t* base_i8,
                             const int32_t* aux, const int32_t* auxh, const uint8_t* queries_padded,
                             int* top8, unsigned long long* cand_fb, int* cnt_fb,
                             int* thr, uint32_t* list, int* list_cnt, int* tile_fail, const int32_t* ext_ids,
                             int32_t* out_ids, float* out_dists, int32_t* out_cnt, hipEvent_t scan_begin,
                             hipEvent_t scan_end, hipStream_t s, const uint8_t* queries_raw) {
    hipError_t e;
    {   // pad the queries (queries_raw [nq][128] -> queries_padded [qpad][128]) + every clear of the batch
        const BfPlan& fb = f.fallback;
        uint32_t* gthr = reinterpret_cast<uint32_t*>(cnt_fb + (size_t)fb.qpad * fb.nsplit);
        const size_t gwords = (size_t)fb.qpad + (size_t)fb.qpad * fb.nsplit;
        const size_t work = (size_t)f.qpad * 32 > gwords ? (size_t)f.qpad * 32 : gwords;
        size_t grid = (work + 255) / 256;
        if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(bf_u8_prep_kernel, dim3((unsigned)grid), dim3(256), 0, s, queries_raw, nq,
                           const_cast<uint8_t*>(queries_padded), f.qpad, tile_fail, f.nqt, gthr, gwords);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    auto scan = [&](const BfScanArgs& sa, int grid) -> hipError_t {
        const void* fn = f.qg == 4 ? (const void*)bf_scan_u8_kernel<4, false> : (const void*)bf_scan_u8_kernel<2, false>;
        hipError_t le = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds_scan);
        if (le != hipSuccess) return le;
        if (f.qg == 4) hipLaunchKernelGGL((bf_scan_u8_kernel<4, false>), dim3(grid), dim3(256), f.lds_scan, s, sa);
        else hipLaunchKernelGGL((bf_scan_u8_kernel<2, false>), dim3(grid), dim3(256), f.lds_scan, s, sa);
        return hipGetLastError();
    };
    // 1. sample pass + thresholds
    BfScanArgs sa{};
    sa.base_i8 = base_i8;
    sa.auxh = auxh;
    sa.queries = queries_padded;
    sa.n = n;
    const int s_nqt = f.qpad / (BF_TQ * f.qg);
    sa.nqt = s_nqt;
    sa.nsplit = f.s_nsplit;
    sa.tps = f.s_tps;
    sa.tile_stride = f.stride;
    sa.top8 = top8;
    {
        const void* fn = f.qg == 4 ? (const void*)bf_scan_u8_kernel<4, true> : (const void*)bf_scan_u8_kernel<2, true>;
        hipError_t le = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds_scan);
        if (le != hipSuccess) return le;
        const dim3 grid(8 * s_nqt * (f.s_nsplit / 8));
        if (f.qg == 4) hipLaunchKernelGGL((bf_scan_u8_kernel<4, true>), grid, dim3(256), f.lds_scan, s, sa);
        else hipLaunchKernelGGL((bf_scan_u8_kernel<2, true>), grid, dim3(256), f.lds_scan, s, sa);
        e = hipGetLastError();
    }
    if (e != hipSuccess) return e;
    if (2 * f.s_nsplit * 8 <= 512)
        hipLaunchKernelGGL(bf_u8_threshold_kernel<8>, dim3((f.qpad + 3) / 4), dim3(256), 0, s, top8, 2 * f.s_nsplit, f.r, nq, f.qpad, thr);
    else   // (s_nsplit <= 64: at most 1024 sample values per query)
        hipLaunchKernelGGL(bf_u8_threshold_kernel<16>, dim3((f.qpad + 3) / 4), dim3(256), 0, s, top8, 2 * f.s_nsplit, f.r, nq, f.qpad, thr);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    // 2. scan with fixed thresholds (tile flags: cleared by the preparation kernel)
    BfScanArgs a{};
    a.base_i8 = base_i8;
    a.auxh = auxh;
    a.queries = queries_padded;
    a.thr = thr;
    a.list = list;
    a.list_cnt = list_cnt;
    a.n = n;
    a.nqt = f.nqt;
    a.nsplit = f.nsplit;
    a.tps = f.tps;
    a.caph = f.caph;
    a.tile_stride = 1;
    if (scan_begin) (void)hipEventRecord(scan_begin, s);
    e = scan(a, 8 * f.nqt * (f.nsplit / 8));
    if (scan_end) (void)hipEventRecord(scan_end, s);
    if (e != hipSuccess) return e;
    // 3. exact re-rank of the listed rows + verification
    RerankListArgs r{};
    r.base = base_u8;
    r.queries = queries_padded;
    r.list = list;
    r.list_cnt = list_cnt;
    r.ext_ids = ext_ids;
    r.out_ids = out_ids;
    r.out_dists = out_dists;
    r.out_cnt = out_cnt;
    r.tile_fail = tile_fail;
    r.n = n;
    r.k = k;
    r.nsplit = f.nsplit;
    r.caph = f.caph;
    r.tps = f.tps;
    r.p2max = f.p2max;
    r.fail_queries = BF_TQ * f.qg;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(bf_rerank_u8_list_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds_rerank);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bf_rerank_u8_list_kernel, dim3(nq), dim3(256), f.lds_rerank, s, r);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    // 4. fallback for flagged tile groups (workgroups of clear groups leave at once)
    e = launch_bf_select_u8_ex(f.fallback, base_i8, aux, queries_padded, cand_fb, cnt_fb, 1, tile_fail, f.qg, s, /*cleared=*/true);
    if (e != hipSuccess) return e;
    return launch_bf_rerank_ex(f.fallback, SP_L2SQR_SIFT, 128, k, base_u8, queries_padded, cand_fb, cnt_fb, ext_ids, out_ids,
                               out_dists, out_cnt, tile_fail, BF_TQ * f.qg, s);
}

// ---- f32 fast path (see bf_scan_f32_kernel) -----------------------------------------------------------------
BfF32Fast bf_f32_fast_plan(int n, int dim, int nq, int k, int space, bool cosine_centred) {
    BfF32Fast f{};
    const bool cosine = space == SP_COSINE || space == SP_ANGULAR;
    const bool space_ok = space == SP_L2 || space == SP_NEGDOT || cosine;
    // centred cosine / angular (round 3): the inner-product mode over rows of dim + 3 columns (row_aug_cosc_kernel)
    f.cosc = cosine && cosine_centred;
    const int dim_rows = dim;
    if (f.cosc) dim += 3;
    f.use = space_ok && dim <= 1024 && n >= 65536 && nq >= 256 && k <= 128;
    if (const char* e = getenv("NMSLIB_GPU_F32_FAST")) f.use = f.use && atoi(e) != 0;
    if (f.cosc)
        if (const char* e = getenv("NMSLIB_GPU_COSC_FAST")) f.use = f.use && atoi(e) != 0;
    if (!f.use) return f;
    f.mode = space == SP_L2 ? 0 : ((space == SP_NEGDOT || f.cosc) ? 1 : 2);
    f.sel_dim = dim;
    // rows longer than 128 (round 3): chunks of 128 dimensions (bf_scan_bf16_kernel<.., KCH>), instantiated for 2, 3, 4, 6, 8
    f.kch = dim <= 128 ? 1 : (dim <= 256 ? 2 : (dim <= 384 ? 3 : (dim <= 512 ? 4 : (dim <= 768 ? 6 : 8))));
    f.dp = 128 * f.kch;
    // queries per workgroup = 256 * qg: four waves x (2 or 4) groups of 32 (see bf_scan_f32_kernel); the larger shape
    // halves the L2 -> LDS stream and the LDS reads per MFMA once the batch fills the chip with it
    f.qg = nq >= 1024 ? 2 : 1;
    if (const char* e = getenv("NMSLIB_GPU_F32_QG")) f.qg = atoi(e) == 2 ? 2 : (atoi(e) == 1 ? 1 : f.qg);
    if (f.kch > 1) f.qg = 1;   // (the fragments of all chunks fill the AGPRs: one group of 32 queries per wave, four waves)
    const int tq = f.kch > 1 ? 128 : 256 * f.qg;
    f.tq = tq;
    f.qpad = (nq + tq - 1) / tq * tq;
    f.nqt = f.qpad / tq;
    f.stride = 16;  // (measured at C2 with the one-product sample pass, ms/step: 4 -> 0.457, 8 -> 0.427, 16 -> 0.419)
    if (const char* e = getenv("NMSLIB_GPU_SAMPLE_STRIDE")) f.stride = atoi(e) > 0 ? atoi(e) : 16;
    // float spaces keep a slack of k' - k rows for the re-rank (the selection score is not the reference formula):
    // aim the threshold at k' = k + max(4, k/8)
    const int kp = k + (k / 8 > 4 ? k / 8 : 4);
    const double kf = (double)kp / f.stride;
    f.r = (int)(1.6 * kf + 3.0 * sqrt(kf) + 3.0 + 0.999);
    // the one-product scan may lower its threshold to the rcap-th best sample score to gain room for its error
    f.rcap = 4 * f.r + 32;
    f.force_precise = false;
    if (const char* e = getenv("NMSLIB_GPU_F32_TERMS")) f.force_precise = atoi(e) == 3;
    const int tiles_all = (n + BF_BN - 1) / BF_BN;
    int ns = (256 + f.nqt - 1) / f.nqt;
    if (ns > tiles_all / 16) ns = tiles_all / 16;
    if (ns > 256) ns = 256;
    ns = (ns + 7) / 8 * 8;
    if (ns < 8) ns = 8;
    while ((tiles_all + ns - 1) / ns > 32768) ns += 8;   // list entries hold the 32-row block index inside the split in 16 bits
    f.nsplit = ns;
    f.tps = (tiles_all + ns - 1) / ns;
    const double mean_half = (double)f.rcap * f.stride / (2.0 * ns);   // (lists sized for the lower threshold)
    int caph = 8;
    while (caph < 3.0 * mean_half + 8.0) caph <<= 1;
    f.caph = caph;
    f.p2max = host_next_pow2((int)(2.5 * f.rcap * f.stride) + 64);      // rows of one query in the re-rank
    if (f.p2max > 2 * ns * caph) f.p2max = host_next_pow2(2 * ns * caph);
    f.lds_scan = 4 * 2 * BF_BN * 256 + 8 * BF_BN * 4 + 64;
    f.lds_scan1 = 9 * BF_BN * 256 + 16 * BF_BN * 4 + 64;
    f.lds_rerank = (size_t)f.p2max * 8 + (2 * (size_t)ns + 1) * 4 + 16;
    const int stiles = (tiles_all + f.stride - 1) / f.stride;
    const int s_nqt = f.qpad / (f.kch > 1 ? 128 : 256);
    int nss = (256 + s_nqt - 1) / s_nqt;
    if (nss > stiles / 16) nss = stiles / 16;
    if (nss > 64) nss = 64;
    nss = (nss + 7) / 8 * 8;
    if (nss < 8) nss = 8;
    f.s_nsplit = nss;
    f.s_tps = (stiles + nss - 1) / nss;
    f.lds_thr = (size_t)host_next_pow2(nss * 2 * 8) * 8 + 16;
    f.fallback = bf_make_plan(n, dim_rows, nq, k, false, tq);
    if (f.kch > 1) f.force_precise = false;   // (no split-product scan at these lengths: such tiles go to the adaptive kernel)
    return f;
}

hipError_t launch_row_maxnorm(const float* rows, int n, int ld, int dim, bool relative_residual, float* out, hipStream_t s,
                              float scale16) {
    hipError_t e = hipMemsetAsync(out, 0, 16, s);
    if (e != hipSuccess || n == 0) return e;
    int grid = (n + 3) / 4;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(row_maxnorm_kernel, dim3(grid), dim3(256), 0, s, rows, n, ld, dim, relative_residual ? 1 : 0,
                       reinterpret_cast<unsigned*>(out), scale16);
    return hipGetLastError();
}

hipError_t launch_split_bf16(const float* src, int rows, int rows_pad, int ld, int dim, void* hi, void* lo,
                             const float* aux, float aux_pad, float* auxp, hipStream_t s, int dp, void* h16, float scale,
                             float* auxp16, float aux16_mul) {
    const size_t total = (size_t)rows_pad * dp;
    if (total == 0) return hipSuccess;
    size_t grid = (total + 255) / 256;
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)grid), dim3(256), 0, s, src, rows, rows_pad, ld, dim,
                       static_cast<__bf16*>(hi), static_cast<__bf16*>(lo), aux, aux_pad, auxp, dp, static_cast<_Float16*>(h16),
                       scale, auxp16, aux16_mul);
    return hipGetLastError();
}

template <int MODE, bool SAMPLE, int QG>
static hipError_t launch_scan_f32_one(const BfScanF32Args& a, int grid, size_t lds, hipStream_t s) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bf_scan_f32_kernel<MODE, SAMPLE, QG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((bf_scan_f32_kernel<MODE, SAMPLE, QG>), dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}
template <int MODE, bool SAMPLE, int QG, int NW>
static hipError_t launch_scan_bf16_one(const BfScanF32Args& a, int grid, size_t lds, hipStream_t s) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bf_scan_bf16_kernel<MODE, SAMPLE, QG, NW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((bf_scan_bf16_kernel<MODE, SAMPLE, QG, NW>), dim3(grid), dim3(NW * 64), lds, s, a);
    return hipGetLastError();
}
// terms: 3 = the split product, 1 = one bf16 product (the sample pass: always, 256 queries per workgroup)
template <int MODE>
static hipError_t launch_scan_f32_mode(const BfScanF32Args& a, bool sample, int terms, int qg, int grid, size_t lds, hipStream_t s, int kch = 1) {
    // The one-product kernel runs with EIGHT waves per workgroup, two per SIMD, each serving half the query groups of the
    // four-wave shape (same workgroup, same tile stream): its K-step is short (two MFMAs), and what one wave cannot hide at
    // one wave per SIMD -- a branch, a VALU chain in front of it, the start-value copy -- the other wave's MFMAs cover.
    // Measured at C2, same box: scan 0.2675 -> 0.2545 ms, sample pass too (step 0.400 -> 0.3925 ms); 512-query batches:
    // scan 0.169 -> 0.155 ms.  NMSLIB_GPU_BF16_W8 (bits: 1 = 512-query tiles, 2 = 256-query tiles, 4 = sample pass) selects
    // the shape for experiments; the split-product kernel needs the whole register file of a SIMD and stays at four waves.
    if (kch > 1) {   // rows longer than 128: four waves x one group of 32 queries, the chunks' fragments in AGPRs
        if (terms != 1) return hipSuccess;   // (no split-product scan at these lengths)
        auto go = [&](auto kern) -> hipError_t {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
            return hipGetLastError();
        };
        switch (kch) {
            case 2: return sample ? go(bf_scan_bf16_kernel<MODE, true, 1, 4, 2>) : go(bf_scan_bf16_kernel<MODE, false, 1, 4, 2>);
            case 3: return sample ? go(bf_scan_bf16_kernel<MODE, true, 1, 4, 3>) : go(bf_scan_bf16_kernel<MODE, false, 1, 4, 3>);
            case 4: return sample ? go(bf_scan_bf16_kernel<MODE, true, 1, 4, 4>) : go(bf_scan_bf16_kernel<MODE, false, 1, 4, 4>);
            case 6: return sample ? go(bf_scan_bf16_kernel<MODE, true, 1, 4, 6>) : go(bf_scan_bf16_kernel<MODE, false, 1, 4, 6>);
            case 8: return sample ? go(bf_scan_bf16_kernel<MODE, true, 1, 4, 8>) : go(bf_scan_bf16_kernel<MODE, false, 1, 4, 8>);
            default: return hipErrorInvalidValue;
        }
    }
    const char* w8e = getenv("NMSLIB_GPU_BF16_W8");
    const int w8 = w8e ? atoi(w8e) : 7;
    if (sample) return (w8 & 4) ? launch_scan_bf16_one<MODE, true, 1, 8>(a, grid, lds, s) : launch_scan_bf16_one<MODE, true, 2, 4>(a, grid, lds, s);
    if (terms == 1) {
        if (qg == 2) return (w8 & 1) ? launch_scan_bf16_one<MODE, false, 2, 8>(a, grid, lds, s) : launch_scan_bf16_one<MODE, false, 4, 4>(a, grid, lds, s);
        return (w8 & 2) ? launch_scan_bf16_one<MODE, false, 1, 8>(a, grid, lds, s) : launch_scan_bf16_one<MODE, false, 2, 4>(a, grid, lds, s);
    }
    if (qg == 2) return launch_scan_f32_one<MODE, false, 4>(a, grid, lds, s);
    return launch_scan_f32_one<MODE, false, 2>(a, grid, lds, s);
}

hipError_t launch_bf_f32_fast(const BfF32Fast& f, int space, int n, int dim, int ldb, int nq, int k, const float* base_orig,
                              const float* sel_rows, const float* aux, const void* base_hi, const void* base_lo,
                              const float* auxp, float bmax, float bres, const float* queries_orig, const float* queries_sel,
                              void* q_hi, void* q_lo, float* top8, unsigned long long* cand_fb, int* cnt_fb, float* thr,
                              uint32_t* list, int* list_cnt, int* tile_fail, int* flags_fb, const int32_t* ext_ids,
                              int32_t* out_ids, float* out_dists, int32_t* out_cnt, hipEvent_t scan_begin,
                              hipEvent_t scan_end, hipStream_t s, const float* queries_raw, float* queries_pad_out,
                              const float* qaux_cosc, const float* queries_centred, int sel_ld, const BfF16Side& h16) {
    // centred cosine / angular (f.cosc): queries_sel holds the augmented queries q+ [qpad][sel_ld] of f.sel_dim columns and
    // base_hi / base_lo the augmented rows (bmax / bres: theirs); queries_centred + qaux_cosc serve the adaptive fallback
    const int sdim = f.cosc ? f.sel_dim : dim, sld = f.cosc ? sel_ld : ldb;
    float* thr1 = thr + f.qpad;
    int* precise = tile_fail + f.nqt;
    hipError_t e;
    {   // pad (when the caller passes the raw queries) + bf16 split + every clear of the batch
        const BfPlan& fb = f.fallback;
        uint32_t* gthr = reinterpret_cast<uint32_t*>(cnt_fb + (size_t)fb.qpad * fb.nsplit);
        const size_t gwords = 2 * ((size_t)fb.qpad + (size_t)fb.qpad * fb.nsplit);
        size_t work = (size_t)f.qpad * f.dp > gwords ? (size_t)f.qpad * f.dp : gwords;
        size_t grid = (work + 255) / 256;
        if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(bf_f32_prep_kernel, dim3((unsigned)grid), dim3(256), 0, s, queries_raw, nq, sdim, queries_sel, f.qpad,
                           sld, queries_pad_out, f.kch > 1 ? nullptr : static_cast<__bf16*>(q_hi),
                           f.kch > 1 ? nullptr : static_cast<__bf16*>(q_lo),
                           tile_fail, 2 * f.nqt, flags_fb, fb.nqt, gthr, gwords, f.dp, static_cast<_Float16*>(h16.q_h16), h16.scale_q);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    auto scan = [&](const BfScanF32Args& sa, bool sample, int terms, int grid) -> hipError_t {
        const size_t lds = terms == 1 ? f.lds_scan1 : f.lds_scan;
        if (f.mode == 0) return launch_scan_f32_mode<SC_L2>(sa, sample, terms, f.qg, grid, lds, s, f.kch);
        if (f.mode == 1) return launch_scan_f32_mode<SC_DOT>(sa, sample, terms, f.qg, grid, lds, s, f.kch);
        return launch_scan_f32_mode<SC_COS>(sa, sample, terms, f.qg, grid, lds, s, f.kch);
    };
    BfScanF32Args a{};
    a.base_hi = static_cast<const __bf16*>(base_hi);
    a.base_lo = static_cast<const __bf16*>(base_lo);
    a.auxp = auxp;
    a.q_hi = static_cast<const __bf16*>(q_hi);
    a.q_lo = static_cast<const __bf16*>(q_lo);
    a.base_h16 = h16.base_h16;
    {
        static const int prio = getenv("NMSLIB_GPU_BF16_PRIO") ? atoi(getenv("NMSLIB_GPU_BF16_PRIO")) : 0;
        a.prio_half = prio;
    }
    a.q_h16 = h16.q_h16;
    a.auxp16 = h16.auxp16;
    a.n = n;
    a.nqt = f.nqt;
    // 1. sample pass (one product: the scores carry the error E1) + thresholds + the choice of the scan per query tile
    BfScanF32Args sa = a;
    sa.nqt = f.qpad / (f.kch > 1 ? 128 : 256);
    sa.nsplit = f.s_nsplit;
    sa.tps = f.s_tps;
    sa.tile_stride = f.stride;
    sa.top8 = top8;
    e = scan(sa, true, 1, 8 * sa.nqt * (f.s_nsplit / 8));
    if (e != hipSuccess) return e;
    // (fallback flags + precise flags: cleared by the prep kernel)
    {
        static const bool dbg_nohit = getenv("NMSLIB_GPU_DEBUG") && (atoi(getenv("NMSLIB_GPU_DEBUG")) & 8192);
        const int depth = f.rcap <= 64 && f.s_nsplit >= 32 ? 4 : 8;
        const dim3 tgrid((f.qpad + 3) / 4);
#define BF_THR_ARGS                                                                                                               \
    top8, 2 * f.s_nsplit, depth, f.r, f.rcap, nq, f.qpad, queries_sel, sld, sdim, h16.scale * (f.mode == 2 ? 1.0f : bmax), h16.bres16, f.tq, \
        f.force_precise ? 1 : (dbg_nohit ? 2 : 0), thr, thr1, precise, h16.scale_q, h16.scale * h16.scale_q
        if (2 * f.s_nsplit * depth <= 512) hipLaunchKernelGGL(bf_f32_threshold_kernel<8>, tgrid, dim3(256), 0, s, BF_THR_ARGS);
        else hipLaunchKernelGGL(bf_f32_threshold_kernel<16>, tgrid, dim3(256), 0, s, BF_THR_ARGS);   // (s_nsplit <= 64)
#undef BF_THR_ARGS
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    // 2. scan with fixed thresholds: every query tile by one of the two kernels (the other's workgroups leave at once)
    a.list = list;
    a.list_cnt = list_cnt;
    a.nsplit = f.nsplit;
    a.tps = f.tps;
    a.caph = f.caph;
    a.tile_stride = 1;
    a.group_flag = precise;
    const int grid = 8 * f.nqt * (f.nsplit / 8);
    if (scan_begin) (void)hipEventRecord(scan_begin, s);
    if (!f.force_precise) {
        a.thr = thr1;
        a.group_want = 0;
        e = scan(a, false, 1, grid);
        if (e != hipSuccess) return e;
    }
    a.thr = thr;
    a.group_want = 1;
    e = scan(a, false, 3, grid);
    if (scan_end) (void)hipEventRecord(scan_end, s);
    if (e != hipSuccess) return e;
    // 3. exact re-rank (reference formula, original rows) + verification
    RerankListF32Args r{};
    r.base = base_orig;
    r.queries = queries_orig;
    r.list = list;
    r.list_cnt = list_cnt;
    r.ext_ids = ext_ids;
    r.out_ids = out_ids;
    r.out_dists = out_dists;
    r.out_cnt = out_cnt;
    r.tile_fail = tile_fail;
    r.n = n;
    r.k = k;
    r.nsplit = f.nsplit;
    r.caph = f.caph;
    r.tps = f.tps;
    r.p2max = f.p2max;
    r.fail_queries = f.tq;
    r.no_split = f.kch > 1 ? 1 : 0;
    r.space = space;
    r.dim = dim;
    r.ldb = ldb;
    r.queries_sel = queries_sel;
    r.sel_dim = sdim;
    r.sel_ld = sld;
    r.scale = h16.scale;
    r.scale_q = h16.scale_q;
    r.bres16 = h16.bres16;
    r.qaux = f.cosc ? qaux_cosc : nullptr;
    r.thr = thr;
    r.thr1 = thr1;
    r.precise = precise;
    r.bmax = bmax;
    r.bres = bres;
    static const int dbg_prof = getenv("NMSLIB_GPU_DEBUG") ? atoi(getenv("NMSLIB_GPU_DEBUG")) : 0;
    static unsigned long long* d_prof = nullptr;
    if (dbg_prof & 4096) {
        if (!d_prof) (void)hipMalloc(&d_prof, 16 * 8);
        (void)hipMemsetAsync(d_prof, 0, 16 * 8, s);
        r.prof = d_prof;
    }
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(bf_rerank_f32_list_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds_rerank);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bf_rerank_f32_list_kernel, dim3(nq), dim3(256), f.lds_rerank, s, r);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (dbg_prof & 4096) {
        unsigned long long h[16];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, d_prof, sizeof(h), hipMemcpyDeviceToHost);
        const double wg = h[10] ? (double)h[10] : 1.0;
        fprintf(stderr, "[rerank f32 list] workgroups %llu rows/query %.1f P %.1f | us since start: prefix %.2f gather %.2f distances %.2f "
                "sort %.2f proof %.2f out %.2f\n", h[10], h[8] / wg, h[9] / wg, h[0] / wg / 100, h[1] / wg / 100, h[2] / wg / 100,
                h[3] / wg / 100, h[4] / wg / 100, h[5] / wg / 100);
    }
    {   // NMSLIB_GPU_DEBUG & 2048: checksums of the intermediate buffers of this batch (determinism screens)
        static const int dbg = getenv("NMSLIB_GPU_DEBUG") ? atoi(getenv("NMSLIB_GPU_DEBUG")) : 0;
        if (dbg & 2048) {
            (void)hipStreamSynchronize(s);
            auto sum = [&](const void* d, size_t bytes) -> unsigned long long {
                std::vector<uint32_t> h(bytes / 4);
                (void)hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost);
                unsigned long long x = 1469598103934665603ull;
                for (uint32_t v : h) x = (x ^ v) * 1099511628211ull;
                return x;
            };
            std::vector<int> cnt((size_t)f.qpad * f.nsplit * 2);
            (void)hipMemcpy(cnt.data(), list_cnt, cnt.size() * 4, hipMemcpyDeviceToHost);
            std::vector<uint32_t> lst((size_t)f.qpad * f.nsplit * 2 * f.caph);
            (void)hipMemcpy(lst.data(), list, lst.size() * 4, hipMemcpyDeviceToHost);
            unsigned long long lx = 1469598103934665603ull;   // only the entries that exist
            long long rows = 0;
            const size_t nl = (size_t)f.nsplit * 2;
            for (size_t i = 0; i < cnt.size(); ++i) {
                rows += cnt[i];
                int have = 0;
                for (int j = 0; j < f.caph && have < cnt[i]; ++j) {
                    const uint32_t e = lst[((i / nl) * f.caph + j) * nl + i % nl];
                    lx = (lx ^ e) * 1099511628211ull;
                    have += __builtin_popcount(e & 0xffffu);
                }
            }
            fprintf(stderr, "[f32fast] top8 %016llx thr %016llx thr1 %016llx flags %016llx cnt %016llx list %016llx rows %lld\n",
                    sum(top8, (size_t)f.qpad * f.s_nsplit * 2 * 8 * 4), sum(thr, (size_t)f.qpad * 4), sum(thr1, (size_t)f.qpad * 4),
                    sum(tile_fail, (size_t)f.nqt * 8), sum(list_cnt, cnt.size() * 4), lx, rows);
        }
    }
    // 4. fallback: the adaptive f32 kernel + its re-rank (verified for l2, with its exact tail) for flagged query tiles
    //    (256 * qg queries = 2 * qg of its tiles)
    return launch_bf_adaptive_f32(f.fallback, space, dim, k, base_orig, sel_rows, aux, queries_orig,
                                  f.cosc ? queries_centred : queries_sel, f.cosc ? qaux_cosc : nullptr, bmax,
                                  cand_fb, cnt_fb, flags_fb, ext_ids, out_ids, out_dists, out_cnt, tile_fail, f.tq / 128, s,
                                  /*cleared=*/true);
}

hipError_t launch_row_aux_f32(const float* base, int n, int ldb, int dim, int space, float* aux,
                              hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(row_aux_f32_kernel, dim3((n + 3) / 4), dim3(256), 0, s, base, n, ldb, dim, space, aux);
    return hipGetLastError();
}
hipError_t launch_prepare_u8(const uint8_t* base, int n, uint8_t* rows_i8, int32_t* aux, int32_t* auxh, hipStream_t s) {
    const int n_pad = bf_u8_rows_padded(n);
    hipLaunchKernelGGL(prepare_u8_kernel, dim3((n_pad + 3) / 4), dim3(256), 0, s, base, n, n_pad, rows_i8, aux, auxh);
    return hipGetLastError();
}
hipError_t launch_pad_rows(const void* src, int rows, int dim, void* dst, int rows_pad, int ld,
                           int elem_bytes, hipStream_t s) {
    const size_t total = (size_t)rows_pad * ld * elem_bytes;
    if (total == 0) return hipSuccess;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pad_rows_kernel, dim3(grid), dim3(256), 0, s, (const uint8_t*)src, rows,
                       dim * elem_bytes, (uint8_t*)dst, rows_pad, ld * elem_bytes);
    return hipGetLastError();
}
hipError_t launch_normalize_rows(float* rows, int n, int ld, int dim, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, s, rows, n, ld, dim);
    return hipGetLastError();
}
hipError_t launch_row_aux_cosc(const float* orig, const float* centred, int n, int ldb, int dim, double mu_norm, float* aux,
                               hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(row_aux_cosc_kernel, dim3((n + 3) / 4), dim3(256), 0, s, orig, centred, n, ldb, dim, mu_norm, aux);
    return hipGetLastError();
}
hipError_t launch_query_aux_cosc(const float* orig, const float* centred, int nq, int ldb, int dim, double mu_norm,
                                 float* qaux, hipStream_t s) {
    if (nq == 0) return hipSuccess;
    hipLaunchKernelGGL(query_aux_cosc_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, orig, centred, nq, ldb, dim, mu_norm,
                       qaux);
    return hipGetLastError();
}
hipError_t launch_row_aug_cosc(const float* orig, const float* centred, int n, int ldb, int dim, double mu_norm, float lambda,
                               float* out, int ldo, int* zero_rows, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(row_aug_cosc_kernel, dim3((n + 3) / 4), dim3(256), 0, s, orig, centred, n, ldb, dim, mu_norm, lambda, out,
                       ldo, zero_rows);
    return hipGetLastError();
}
hipError_t launch_query_aug_cosc(const float* centred, const float* qaux, int nq, int qpad, int ldb, int dim, float lambda,
                                 float* out, int ldo, hipStream_t s) {
    if (qpad == 0) return hipSuccess;
    hipLaunchKernelGGL(query_aug_cosc_kernel, dim3((qpad + 3) / 4), dim3(256), 0, s, centred, qaux, nq, qpad, ldb, dim, lambda,
                       out, ldo);
    return hipGetLastError();
}
hipError_t launch_col_stats(const float* base, int n, int ldb, int dim, double* stats, hipStream_t s) {
    hipError_t e = hipMemsetAsync(stats, 0, ((size_t)ldb + 1) * 8, s);
    if (e != hipSuccess || n == 0) return e;
    int grid = (n + 255) / 256;
    if (grid > 2048) grid = 2048;
    const int rpb = (n + grid - 1) / grid;
    hipLaunchKernelGGL(col_stats_kernel, dim3(grid), dim3(256), 0, s, base, n, ldb, dim, rpb, stats);
    return hipGetLastError();
}
hipError_t launch_center_rows(const float* src, const float* mean, int rows, int rows_valid, int ld, int dim, float* dst,
                              hipStream_t s) {
    const size_t total = (size_t)rows * ld;
    if (total == 0) return hipSuccess;
    size_t grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(center_rows_kernel, dim3((unsigned)grid), dim3(256), 0, s, src, mean, rows, rows_valid, ld, dim, dst);
    return hipGetLastError();
}
hipError_t launch_pair_distance(int space, const void* a, const void* b, int dim, float* out,
                                hipStream_t s) {
    hipLaunchKernelGGL(pair_distance_kernel, dim3(1), dim3(64), 0, s, space, a, b, dim, out);
    return hipGetLastError();
}
hipError_t launch_merge_topk(const float* dists_in, const int32_t* ids_in, size_t shard_stride, int nshards, int nq,
                             int k, float* dists_out, int32_t* ids_out, hipStream_t s) {
    return launch_merge_topk_ex(dists_in, ids_in, shard_stride, nshards, nq, k, dists_out, ids_out, nullptr, nullptr, s);
}
hipError_t launch_merge_topk_ex(const float* dists_in, const int32_t* ids_in, size_t shard_stride, int nshards, int nq,
                                int k, float* dists_out, int32_t* ids_out, int32_t* cnt_out, const int32_t* ext_ids,
                                hipStream_t s) {
    const int P = host_next_pow2(nshards * k < 2 ? 2 : nshards * k);
    const size_t lds = (size_t)P * 8;
    if (lds > 64 * 1024) {  // beyond the LDS sort: rank every item by bisection in the other lists
        hipLaunchKernelGGL(merge_topk_big_kernel, dim3(nq), dim3(256), 0, s, dists_in, ids_in, shard_stride, nshards, nq, k,
                           dists_out, ids_out, cnt_out, ext_ids);
        return hipGetLastError();
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(merge_topk_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(merge_topk_kernel, dim3(nq), dim3(256), lds, s, dists_in, ids_in, shard_stride, nshards, nq,
                       k, dists_out, ids_out, cnt_out, ext_ids);
    return hipGetLastError();
}

}  // namespace gfxknn
