// HNSW search for small batches: one workgroup per query (see below).  Split from hnsw_kernels.hip.
#include <cstdio>
#include <cstdlib>

#include "hnsw_args.cuh"
#include "hnsw_common.cuh"
#include "kernels.hpp"

namespace gfxknn {

// =======================================================================================
// Multi-wave form of the same search for small batches (round 3): one WORKGROUP per query.
//   wave 0     - control: owns the sorted array and the visited table and takes every decision of
//                SearchV1Merge (hnsw_distfunc_opt.cc:200-274) exactly as hnsw_search_body does;
//   waves 1..4 - gather: each takes 8 of the <= 32 rows of a frontier (8 lanes x 16 bytes per row and step),
//                accumulates in the per-lane order of frontier_distances (bit-identical distances) and leaves
//                the results in LDS.
// Why: at batch 1024 the one-wave kernel puts one wave on each of the chip's 1024 SIMDs.  A lone wave issues one
// instruction per four clocks, so the ~1000 instructions of an expansion cost as much as the HBM gather they
// wait for, and neither hides the other (profiles/r02_hnsw_pmc.json: SQ_ACTIVE_INST_ANY 49 %, SQ_WAIT_ANY 49 %).
// Here the gather has its own instruction streams on other SIMDs (four loads per lane instead of sixteen), and
// the control wave works one expansion ahead: as soon as the distances of expansion i are in, it names the node
// of expansion i+1 -- the array's next unused item, or the closest item just accepted when that is closer; the
// sequential algorithm would pick the same node after the merge -- runs that node's visited filter, hands its
// rows to the gather waves, and only then merges the accepted items of expansion i into the array.  Every
// decision is the sequential algorithm's own, so ids, distances and the ndc / hops counters are unchanged.
// Hand-off: nbr[] / nd[] / ctl[] in LDS and two s_barrier per expansion (A: list published, B: distances ready).
// =======================================================================================
constexpr int MW_NW = 4;                       // gather waves per query
constexpr int MW_THREADS = 64 * (MW_NW + 1);
constexpr int MW_MAX_EMAX = 4;                 // sorted array up to 256 items

// [0] descent, [1] pick + scans, [2] filter + publish (not pipelined), [3] waiting for distances,
// [4] accept + decision + pipelined filter, [5] merge, [6] queries, [7] mispredictions (must stay 0)
__device__ unsigned long long g_hnsw_mw_prof[8];

// LDS traffic of this wave done, then the workgroup barrier (no vmcnt wait: an adjacency prefetch may be in flight)
__device__ __forceinline__ void mw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T>
__device__ __forceinline__ T dpp_row_mirror(T v) {
    return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
}
// minimum over the wave (every lane returns it): DPP inside the rows of 16, four readlanes across them
__device__ __forceinline__ float wave_min_f32(float v) {
    v = fminf(v, dpp_quad_xor1(v));
    v = fminf(v, dpp_quad_xor2(v));
    v = fminf(v, dpp_half_mirror(v));
    v = fminf(v, dpp_row_mirror(v));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fminf(fminf(r0, r1), fminf(r2, r3));
}

template <int SPACE>
__device__ __forceinline__ void mw_gather_loop(const HnswArgs& a, const float* qv, const int* nbr, float* nd,
                                               const int* ctl, const int w, const int lane) {
    const HnswDeviceGraph& g = a.g;
    const int g8 = lane >> 3, sub = lane & 7;
    const float* rows = reinterpret_cast<const float*>(g.rows);
    const int dlast = g.ldv - 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // (the guard only bounds the damage of a protocol error: a wave that ended lets s_barrier through)
    for (int guard = 0; guard < (1 << 26); ++guard) {
        mw_barrier();  // A: list published
        const int m = __builtin_amdgcn_readfirstlane(ctl[0]);
        if (m < 0) break;
        for (int base = w * 8; base < m; base += 8 * MW_NW) {
            const int idx = base + g8;
            const int id = nbr[idx];  // (the control wave pads the list to 64 entries with its last id)
            const float* rp = rows + (size_t)id * g.ldv;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
            auto fetch = [&](f32x4 (&bb)[4], int cb) __attribute__((always_inline)) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int d = cb + sub * 4 + 32 * it;
                    bb[it] = *reinterpret_cast<const f32x4*>(rp + (d < dlast ? d : dlast));
                }
            };
            auto consume = [&](const f32x4 (&bb)[4], int cb) __attribute__((always_inline)) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int d = cb + sub * 4 + 32 * it;
                    const bool ok = d < g.ldv;
                    f32x4 qq = *reinterpret_cast<const f32x4*>(qv + (d < dlast ? d : dlast));
                    qq = ok ? qq : zero;
                    accum4<SPACE>(qq, ok ? bb[it] : zero, s0, s1, s2);
                }
            };
            // 128 floats of the row per step; the next step's loads are requested before this step is consumed
            f32x4 cur[4], nxt[4];
            int cb = 0;
            fetch(cur, 0);
            while (true) {
                if (cb + 128 < g.ldv) {
                    fetch(nxt, cb + 128);
                    consume(cur, cb);
#pragma unroll
                    for (int it = 0; it < 4; ++it) cur[it] = nxt[it];
                    cb += 128;
                } else {
                    consume(cur, cb);
                    break;
                }
            }
            float r0, r1 = 0.f, r2 = 0.f;
            if constexpr (DistTraits<SPACE>::kMax) r0 = group8_max(s0);
            else r0 = group8_sum(s0);
            if constexpr (DistTraits<SPACE>::kThree) {
                r1 = group8_sum(s1);
                r2 = group8_sum(s2);
            }
            if (sub == 0 && idx < m) nd[idx] = finish_dist<SPACE>(r0, r1, r2);
        }
        mw_barrier();  // B: distances written
    }
}

template <int SPACE, int SA_EMAX>
__device__ __forceinline__ void mw_control(const HnswArgs& a, const int q, float* keys, int* idu, float* qv, int* nbr,
                                           float* nd, float* sk, int* si, int* ctl, uint32_t* table, const int lane) {
    const HnswDeviceGraph& g = a.g;
    int ndc = 0, hops = 0, hops_up = 0, nvisited = 0;
    bool overflow = false, inflight = false;

    auto visit = [&](uint32_t id) -> bool {
        uint32_t hsh = (id * 2654435761u) >> a.table_shift;
        const uint32_t mask = (uint32_t)a.table_size - 1u;
        for (int probe = 0; probe < a.table_size; ++probe) {
            const uint32_t old = atomicCAS(&table[hsh], HT_EMPTY, id);
            if (old == HT_EMPTY) return true;
            if (old == id) return false;
            hsh = (hsh + 1) & mask;
        }
        return false;
    };
    auto publish = [&](int m) __attribute__((always_inline)) {
        if (lane == 0) ctl[0] = m;
        mw_barrier();  // A
    };
    // adjacency of node c, shifted by one word: lane L holds neighbour L, the last lanes hold the count (word 0)
    auto load_adj0 = [&](int c) -> int {
        const int w = lane + 1 <= g.maxM0 ? lane + 1 : 0;
        return g.links0[(size_t)c * (g.maxM0 + 1) + w];
    };
    // visited filter of an adjacency list: the unvisited neighbours go to nbr[0..m), padded with the last one
    auto filter = [&](int v) -> int {
        const int cntn = __builtin_amdgcn_readlane(v, 63);
        bool isn = false;
        if (lane < cntn) isn = visit((uint32_t)v);
        const u64 nmask = __ballot(isn);
        const int m = __popcll(nmask);
        if (m > 0) {
            const int lastid = __builtin_amdgcn_readlane(v, 63 - __clzll((long long)nmask));
            if (isn) nbr[__popcll(nmask & ((1ull << lane) - 1ull))] = v;
            if (lane >= m) nbr[lane] = lastid;
        }
        nvisited += m;
        if (nvisited > (a.table_size - (a.table_size >> 3))) overflow = true;
        return m;
    };
    long long pc[6] = {0, 0, 0, 0, 0, 0};
    long long pt = a.prof ? (long long)__builtin_readcyclecounter() : 0;
    auto lap = [&](int ph) __attribute__((always_inline)) {
        if (a.prof) {
            const long long now = (long long)__builtin_readcyclecounter();
            pc[ph] += now - pt;
            pt = now;
        }
    };

    // ---- entry point + greedy descent (hnsw_distfunc_opt.cc:168-198) ----
    int cur = g.enterpoint;
    nbr[lane] = cur;
    publish(1);
    mw_barrier();  // B
    float curdist = nd[0];
    ndc += 1;
    for (int lvl = g.maxlevel; lvl > 0; --lvl) {
        bool changed = true;
        while (changed) {
            changed = false;
            const int64_t off = g.up_off[cur] + (int64_t)(lvl - 1) * (g.maxM + 1);
            const int v = g.up_links[off + (lane + 1 <= g.maxM ? lane + 1 : 0)];
            const int cntl = __builtin_amdgcn_readlane(v, 63);
            hops_up++;
            if (cntl > 0) {
                const int lastid = __builtin_amdgcn_readlane(v, cntl - 1);
                nbr[lane] = lane < cntl ? v : lastid;
                publish(cntl);
                mw_barrier();  // B
                ndc += cntl;
                // sequential "if (d < curdist)" scan == first index attaining the minimum
                const float dl = lane < cntl ? nd[lane] : INFINITY;
                const float dmin = wave_min_f32(dl);
                if (dmin < curdist) {
                    const u64 mm = __ballot(lane < cntl && dl == dmin);
                    curdist = dmin;
                    cur = __builtin_amdgcn_readlane(v, __ffsll((long long)mm) - 1);
                    changed = true;
                }
            }
        }
    }

    // ---- level 0 (hnsw_distfunc_opt.cc:200-274) ----
    int n = 1;
    if (lane == 0) {
        keys[0] = curdist;
        idu[0] = cur;
        (void)visit((uint32_t)cur);
    }
    nvisited = 1;
    __builtin_amdgcn_wave_barrier();

    auto first_unused = [&](int from) -> int {
        int fu = n;
        for (int base = from; base < n && fu == n; base += 64) {
            const int i = base + lane;
            const u64 mk = __ballot(i < n && idu[i] >= 0);
            if (mk) fu = base + (__ffsll((long long)mk) - 1);
        }
        return fu;
    };

    int cursor = 0;
    // pre_*: the array's first unused item behind the node being expanded, with its adjacency requested early
    int pre_node = -1, pre_v = 0;
    float pre_key = INFINITY;
    bool pre_ok = false;
    // pipe_*: the node of the NEXT expansion, named before the merge; its rows are already with the gather waves
    int pipe_node = -1, pipe_m = 0;
    bool bad = false;
    lap(0);
    while (true) {
        const int lim = n < a.ef ? n : a.ef;
        const int fu = first_unused(cursor);
        if (fu >= lim) break;
        const int c = idu[fu] & 0x7FFFFFFF;
        if (lane == 0) idu[fu] |= (int)0x80000000;
        cursor = fu + 1;
        hops++;
        const float topKey = keys[n - 1];
        const int size0 = n;
        int m;
        if (c == pipe_node) {
            m = pipe_m;
            pipe_node = -1;
            lap(1);
        } else {
            if (pipe_node >= 0) {  // cannot happen: the pipelined choice is the sequential one
                bad = true;
                break;
            }
            lap(1);
            const int v = (c == pre_node) ? pre_v : load_adj0(c);
            m = filter(v);
            if (overflow) break;
            if (m > 0) {
                publish(m);
                inflight = true;
            }
            lap(2);
        }
        {   // adjacency of the likely next expansion, one expansion early (a random HBM read: ~1 us)
            const int fu2 = first_unused(cursor);
            if (fu2 < lim) {
                const int node = idu[fu2] & 0x7FFFFFFF;
                if (node != pre_node) {
                    pre_node = node;
                    pre_v = load_adj0(node);
                }
                pre_key = keys[fu2];
                pre_ok = true;
            } else {
                pre_node = -1;
                pre_key = INFINITY;
                pre_ok = false;
            }
        }
        lap(1);
        if (m == 0) continue;
        ndc += m;
        mw_barrier();  // B: distances of this expansion
        inflight = false;
        lap(3);

        // accept d < topKey || size < ef   (:240)
        float dj = INFINITY;
        int idj = -1;
        bool acc = false;
        if (lane < m) {
            dj = nd[lane];
            idj = nbr[lane];
            acc = (dj < topKey) || (size0 < a.ef);
        }
        const u64 amask = __ballot(acc);
        const int m2 = __popcll(amask);

        // The next expansion, named before the merge.  pre_node is the first unused item of the array; the merge
        // moves it up by the number of accepted keys below it and puts nothing unused in front of it except those
        // accepted items.  So: no accepted key below pre_key -> pre_node is next (its position, unchanged, is
        // below ef); otherwise the closest accepted item is next (it lands in front of pre_node, hence below ef).
        // Equal keys leave the order to the merge: no early choice then.
        if (pre_ok) {
            int P = -1, Pv = 0;
            const u64 below = __ballot(acc && dj < pre_key);
            const u64 equal = __ballot(acc && dj == pre_key);
            if (!equal) {
                if (!below) {
                    P = pre_node;
                    Pv = pre_v;
                } else {
                    const float best = wave_min_f32(acc ? dj : INFINITY);
                    const u64 bm = __ballot(acc && dj == best);
                    if (__popcll(bm) == 1) {
                        P = __builtin_amdgcn_readlane(idj, __ffsll((long long)bm) - 1);
                        Pv = load_adj0(P);
                    }
                }
            }
            if (P >= 0) {
                pipe_m = filter(Pv);
                if (overflow) break;
                pipe_node = P;
                if (pipe_m > 0) {
                    publish(pipe_m);
                    inflight = true;
                }
            }
        }
        lap(4);
        if (m2 == 0) continue;

        // ascending order of the accepted items (std::sort, :251); ties keep list order
        int rank = 0;
        for (u64 mm = amask; mm;) {
            const int j = __ffsll((long long)mm) - 1;
            mm &= mm - 1;
            const float dother = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dj), j));
            rank += (dother < dj || (dother == dj && j < lane)) ? 1 : 0;
        }
        if (acc) {
            sk[rank] = dj;
            si[rank] = idj;
        }
        __builtin_amdgcn_wave_barrier();

        // the merge of hnsw_search_body: all accepted items at once unless two keys involved are equal
        bool tie = false;
        float kreg[SA_EMAX];
        int cntv[SA_EMAX];
#pragma unroll
        for (int e = 0; e < SA_EMAX; ++e) {
            const int i = lane + 64 * e;
            kreg[e] = (e * 64 < n && i < n) ? keys[i] : INFINITY;
            cntv[e] = 0;
        }
        const float mykey = lane < m2 ? sk[lane] : INFINITY;
        const int myid = lane < m2 ? si[lane] : -1;
        int myless = 0;
        for (int t = 0; t < m2; ++t) {
            const float skt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mykey), t));
            int less = 0;
#pragma unroll
            for (int e = 0; e < SA_EMAX; ++e) {
                if (e * 64 < n) {
                    less += __popcll(__ballot(kreg[e] < skt));
                    cntv[e] += (skt < kreg[e]) ? 1 : 0;
                    tie |= (skt == kreg[e]);
                }
            }
            if (lane == t) myless = less;
        }
        tie |= (lane + 1 < m2) && (mykey == __shfl_down(mykey, 1, 64));
        if (!__any(tie)) {
            int ireg[SA_EMAX];
#pragma unroll
            for (int e = 0; e < SA_EMAX; ++e) {
                const int i = lane + 64 * e;
                ireg[e] = (e * 64 < n && i < n) ? idu[i] : 0;
            }
            const int newn = n + m2 < a.cap ? n + m2 : a.cap;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e = 0; e < SA_EMAX; ++e) {
                const int i = lane + 64 * e;
                if (e * 64 < n && i < n && cntv[e] > 0) {
                    const int np = i + cntv[e];
                    if (np < newn) {
                        keys[np] = kreg[e];
                        idu[np] = ireg[e];
                    }
                }
            }
            if (lane < m2) {
                const int np = myless + lane;
                if (np < newn) {
                    keys[np] = mykey;
                    idu[np] = myid;
                }
            }
            const int first = __builtin_amdgcn_readlane(myless, 0);
            if (first < cursor) cursor = first;
            n = newn;
            __builtin_amdgcn_wave_barrier();
        } else {
            // SortArrBI::push_or_replace_non_empty_exp for each, in order (sort_arr_bi.h:159-199)
            for (int t = 0; t < m2; ++t) {
                const float key = sk[t];
                const int id = si[t];
                const float lastk = keys[n - 1];
                if (lastk <= key) {
                    if (n < a.cap) {
                        if (lane == 0) {
                            keys[n] = key;
                            idu[n] = id;
                        }
                        n++;
                    }
                } else {
                    int less = 0, leq = 0;
#pragma unroll
                    for (int e = 0; e < SA_EMAX; ++e) {
                        if (e * 64 < n) {
                            const int i = lane + 64 * e;
                            const float kv = i < n ? keys[i] : INFINITY;
                            less += __popcll(__ballot(kv < key));
                            leq += __popcll(__ballot(kv <= key));
                        }
                    }
                    int p = less;
                    if (leq != less) {
                        int curr = n - 1, prev = curr, dstep = 1;
                        while (curr > 0 && keys[curr] > key) {
                            prev = curr;
                            curr -= dstep;
                            dstep *= 2;
                            if (dstep > curr) dstep = curr;
                        }
                        p = curr;
                        for (int i = curr; i < prev && keys[i] < key; ++i) p = i + 1;
                    }
                    const int newn = n < a.cap ? n + 1 : a.cap;
                    float rk[SA_EMAX];
                    int ri[SA_EMAX];
#pragma unroll
                    for (int e = 0; e < SA_EMAX; ++e) {
                        if (e * 64 < newn) {
                            const int i = lane + 64 * e;
                            if (i > p && i < newn) {
                                rk[e] = keys[i - 1];
                                ri[e] = idu[i - 1];
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int e = 0; e < SA_EMAX; ++e) {
                        if (e * 64 < newn) {
                            const int i = lane + 64 * e;
                            if (i > p && i < newn) {
                                keys[i] = rk[e];
                                idu[i] = ri[e];
                            }
                        }
                    }
                    if (lane == 0) {
                        keys[p] = key;
                        idu[p] = id;
                    }
                    n = newn;
                    if (p < cursor) cursor = p;  // :261-266
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        lap(5);
    }
    // ---- release the gather waves (every published list is collected first) ----
    if (inflight) mw_barrier();  // B
    publish(-1);
    if (a.prof && lane == 0) {
        for (int i = 0; i < 6; ++i) atomicAdd(&g_hnsw_mw_prof[i], (unsigned long long)pc[i]);
        atomicAdd(&g_hnsw_mw_prof[6], 1ull);
    }
    if (bad && lane == 0) atomicAdd(&g_hnsw_mw_prof[7], 1ull);

    // ---- results: first k items, ties ordered by internal id (as hnsw_search_body) ----
    const bool redo = overflow || bad;
    const int kk = redo ? 0 : (a.k < n ? a.k : n);
    for (int i = lane; i < a.k; i += 64) {
        if (i < kk) {
            const float ki = keys[i];
            const int id = idu[i] & 0x7FFFFFFF;
            int r = i;
            for (int j = i - 1; j >= 0 && keys[j] == ki; --j) r -= ((idu[j] & 0x7FFFFFFF) > id) ? 1 : 0;
            for (int j = i + 1; j < kk && keys[j] == ki; ++j) r += ((idu[j] & 0x7FFFFFFF) < id) ? 1 : 0;
            a.out_ids[(size_t)q * a.k + r] = g.ext_ids ? g.ext_ids[id] : id;
            a.out_dists[(size_t)q * a.k + r] = ki;
        } else {
            a.out_ids[(size_t)q * a.k + i] = -1;
            a.out_dists[(size_t)q * a.k + i] = INFINITY;
        }
    }
    if (lane == 0) {
        a.out_cnt[q] = kk;
        if (a.out_ndc) a.out_ndc[q] = ndc;
        if (a.out_hops) a.out_hops[q] = hops;
        if (a.out_hops_up) a.out_hops_up[q] = hops_up;
        if (a.status) a.status[q] = redo ? 1 : 0;
        if (redo && a.fix_list) a.fix_list[atomicAdd(a.fix_count, 1)] = q;
    }
}

template <int SPACE, int SA_EMAX>
__global__ __launch_bounds__(MW_THREADS) void hnsw_search_mw_kernel(HnswArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const HnswDeviceGraph& g = a.g;
    const int q = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // ---- LDS carve-up (all offsets multiples of 16 bytes) ----
    float* keys = reinterpret_cast<float*>(smem);            // [capa]
    int* idu = reinterpret_cast<int*>(keys + a.capa);        // [capa]  id | used<<31
    float* qv = reinterpret_cast<float*>(idu + a.capa);      // [ldv]
    int* nbr = reinterpret_cast<int*>(qv + g.ldv);           // [64] rows of the frontier being gathered
    float* nd = reinterpret_cast<float*>(nbr + 64);          // [64] their distances
    float* sk = nd + 64;                                     // [64] accepted keys, sorted
    int* si = reinterpret_cast<int*>(sk + 64);               // [64] accepted ids
    int* ctl = si + 64;                                      // [4]  rows in nbr[] (-1: the search is over)
    uint32_t* table = reinterpret_cast<uint32_t*>(ctl + 4);  // [table_size]

    if (g.n == 0) {  // (uniform: every wave leaves)
        if (wave == 0) {
            for (int i = lane; i < a.k; i += 64) {
                a.out_ids[(size_t)q * a.k + i] = -1;
                a.out_dists[(size_t)q * a.k + i] = INFINITY;
            }
            if (lane == 0) {
                a.out_cnt[q] = 0;
                if (a.out_ndc) a.out_ndc[q] = 0;
                if (a.out_hops) a.out_hops[q] = 0;
                if (a.out_hops_up) a.out_hops_up[q] = 0;
                if (a.status) a.status[q] = 0;
            }
        }
        return;
    }
    // ---- stage the query, clear the visited table (all waves) ----
    {
        const float* src = reinterpret_cast<const float*>(a.queries) + (size_t)q * g.dim;
        if (wave == 0) {
            float ss = 0.f;
            for (int d = lane; d < g.ldv; d += 64) {
                const float v = d < g.dim ? src[d] : 0.f;
                qv[d] = v;
                ss = fmaf(v, v, ss);
            }
            if (g.normalize_query) {  // hnsw_distfunc_opt.cc:160-162
                ss = wave_sum(ss);
                if (ss != 0.0f) {
                    const float inv = 1.0f / sqrtf(ss);
                    for (int d = lane; d < g.dim; d += 64) qv[d] *= inv;
                }
            }
        }
        for (int i = threadIdx.x; i < a.table_size; i += MW_THREADS) table[i] = HT_EMPTY;
        if (threadIdx.x == 0) ctl[0] = 0;
    }
    __syncthreads();
    if (wave == 0) mw_control<SPACE, SA_EMAX>(a, q, keys, idu, qv, nbr, nd, sk, si, ctl, table, lane);
    else mw_gather_loop<SPACE>(a, qv, nbr, nd, ctl, wave - 1, lane);
}


template <int SPACE>
static hipError_t launch_mw_space(const HnswArgs& a, size_t lds, int sa_emax, hipStream_t s) {
    auto go = [&](auto kern) -> hipError_t {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(a.nq), dim3(MW_THREADS), lds, s, a);
        return hipGetLastError();
    };
    if (sa_emax <= 2) return go(hnsw_search_mw_kernel<SPACE, 2>);
    return go(hnsw_search_mw_kernel<SPACE, 4>);
}

hipError_t launch_hnsw_search_mw(const HnswArgs& a, size_t lds_bytes, int sa_emax, hipStream_t s) {
    if (sa_emax > MW_MAX_EMAX || a.g.maxM0 > 62 || a.g.maxM > 62) return hipErrorInvalidValue;
    switch (a.g.space) {
        case SP_L2SQR: return launch_mw_space<SP_L2SQR>(a, lds_bytes, sa_emax, s);
        case SP_L2: return launch_mw_space<SP_L2>(a, lds_bytes, sa_emax, s);
        case SP_L1: return launch_mw_space<SP_L1>(a, lds_bytes, sa_emax, s);
        case SP_LINF: return launch_mw_space<SP_LINF>(a, lds_bytes, sa_emax, s);
        case SP_NORMCOS: return launch_mw_space<SP_NORMCOS>(a, lds_bytes, sa_emax, s);
        case SP_COSINE: return launch_mw_space<SP_COSINE>(a, lds_bytes, sa_emax, s);
        case SP_ANGULAR: return launch_mw_space<SP_ANGULAR>(a, lds_bytes, sa_emax, s);
        case SP_NEGDOT: return launch_mw_space<SP_NEGDOT>(a, lds_bytes, sa_emax, s);
        default: return hipErrorInvalidValue;
    }
}

void hnsw_mw_read_prof(unsigned long long out[8]) {
    unsigned long long z[8] = {0};
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hnsw_mw_prof), sizeof(z));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_hnsw_mw_prof), z, sizeof(z));
}

}  // namespace gfxknn
