// HNSW search on gfx950: one wavefront (64 lanes) per query.
//
// Restates Hnsw::SearchV1Merge (src/method/hnsw_distfunc_opt.cc:152-283) and its generic twin
// baseSearchAlgorithmV1Merge (src/method/hnsw.cc:1174-1300) step for step:
//   1. greedy descent through the upper levels (:173-198);
//   2. level-0 best-first over a bounded sorted array (SortArrBI, include/sort_arr_bi.h) of
//      max(ef, k) items: expand the first unused item, evaluate its unvisited neighbours,
//      keep those with d < topKey (or while fewer than ef items), insert them in ascending
//      order with SortArrBI::push_or_replace_non_empty_exp semantics (:159-199);
//   3. emit the first k items (:276-281) through KNNQueue ordering (knnqueue.h:55-64).
// What changes is the execution shape: the <=32 neighbours of one expansion are a frontier
// batch.  Their ids are filtered through a visited set kept in LDS (open-addressing hash,
// exact) and the unvisited rows are gathered from HBM together - 8 lanes per row, 16 bytes
// per lane per step, up to 32 rows in flight per wave - with the distance fused into the
// gather and an 8-lane shuffle reduction.  The sorted array lives in LDS and is shifted by
// all lanes at once.  ndc / hops counters feed the roofline.
#include <cstdio>
#include <cstdlib>

#include "hnsw_args.hpp"
#include "hnsw_common_dev.hpp"
#include "kernels.hpp"

namespace gfxknn {


// [0] descent, [1] pick + adjacency, [2] visited filter, [3] gather + distances, [4] accept + sort, [5] inserts, [6] waves
__device__ unsigned long long g_hnsw_prof[8];

constexpr int SA_EMAX_MAX = 16;  // sorted array up to 64*16 = 1024 items

// EMAX = sorted-array items per lane (cap <= 64*EMAX): 2 for ef <= 128, 4 for <= 256, 16 otherwise.
// WIDE: level-0 lists of more than 62 neighbours (maxM0 up to 126, i.e. M >= 32): second adjacency chunk,
// 128-entry neighbour staging, two insertion rounds.  Kept out of the common instantiation.
// ---- adjacency lists of any length (round 3: M / maxM > 62, maxM0 > 126 -- hnsw.cc:189-208 takes any M) -------------
// The kernels above give a list one or two words per lane.  SearchOld and the HBM-array SearchV1Merge walk longer lists in
// chunks of 64 neighbours (list order kept: neighbour i is list word i + 1); their frontier arrays hold a.nbcap entries.
// unvisited neighbours of list[] -> nbr[0 .. m), returns m
template <class Visit>
__device__ __forceinline__ int collect_unvisited_any(const int* list, int* nbr, int lane, Visit&& visit) {
    const int cnt = __builtin_amdgcn_readfirstlane(list[0]);
    int m = 0;
    for (int c0 = 0; c0 < cnt; c0 += 64) {
        const int i = c0 + lane;
        const int nb = i < cnt ? list[i + 1] : 0;
        bool isn = false;
        if (i < cnt) isn = visit((uint32_t)nb);
        const u64 mask = __ballot(isn);
        if (isn) nbr[m + __popcll(mask & ((1ull << lane) - 1ull))] = nb;
        m += __popcll(mask);
    }
    return m;
}
// one greedy step on an upper level over a list of any length: the FIRST neighbour attaining the minimum, if it is closer
// than curdist (the sequential "if (d < curdist)" scan of hnsw.cc / hnsw_distfunc_opt.cc:176-196).  Returns the list length.
template <int SPACE>
__device__ __forceinline__ int greedy_step_any(const HnswDeviceGraph& g, const int* list, const float* qv, const uint8_t* qb,
                                               int qnorm, int* nbr, float* nd, int lane, int& cur, float& curdist,
                                               bool& changed) {
    const int cnt = __builtin_amdgcn_readfirstlane(list[0]);
    for (int c0 = 0; c0 < cnt; c0 += 64)
        if (c0 + lane < cnt) nbr[c0 + lane] = list[c0 + lane + 1];
    __builtin_amdgcn_wave_barrier();
    if (cnt > 0) {
        frontier_distances<SPACE>(g, qv, qb, qnorm, nbr, nd, cnt, lane);
        u64 key = ~0ull;
        for (int c0 = 0; c0 < cnt; c0 += 64) {
            const int i = c0 + lane;
            if (i < cnt) {
                const u64 k2 = ((u64)f32_ord(nd[i]) << 32) | (uint32_t)i;
                key = k2 < key ? k2 : key;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const u64 other = __shfl_xor(key, o, 64);
            key = other < key ? other : key;
        }
        const float dmin = ord_f32((uint32_t)(key >> 32));
        if (dmin < curdist) {
            curdist = dmin;
            cur = nbr[(uint32_t)key];
            changed = true;
        }
    }
    __builtin_amdgcn_wave_barrier();
    return cnt;
}

template <int SPACE, bool BITSET, int SA_EMAX, bool WIDE>
__device__ __forceinline__ void hnsw_search_body(const HnswArgs& a, char* smem, const int q, uint32_t* const bits) {
    const HnswDeviceGraph& g = a.g;
    const int lane = threadIdx.x;
    constexpr bool kU8 = DistTraits<SPACE>::kU8;

    // ---- LDS carve-up (all offsets multiples of 16 bytes) ----
    float* keys = reinterpret_cast<float*>(smem);                  // [capa]
    int* idu = reinterpret_cast<int*>(keys + a.capa);              // [capa]  id | used<<31
    const int qfloats = kU8 ? 32 : g.ldv;
    float* qv = reinterpret_cast<float*>(idu + a.capa);            // [ldv] (u8: 128 bytes)
    const int nbcap = WIDE ? a.nbcap : 64;                         // neighbours of one expansion (wide lists: a multiple of 64 >= maxM0, maxM)
    int* nbr = reinterpret_cast<int*>(qv + qfloats);               // [nbcap]
    float* nd = reinterpret_cast<float*>(nbr + nbcap);             // [nbcap]
    float* sk = nd + nbcap;                                        // [64] accepted keys, sorted
    int* si = reinterpret_cast<int*>(sk + 64);                     // [64] accepted ids
    uint32_t* table = reinterpret_cast<uint32_t*>(si + 64);        // [table_size]

    // ---- stage the query ----
    int qnorm = 0;
    if constexpr (kU8) {
        const uint8_t* src = a.query_rows ? reinterpret_cast<const uint8_t*>(g.rows) + (size_t)a.query_rows[q] * 128
                                          : reinterpret_cast<const uint8_t*>(a.queries) + (size_t)q * 128;
        const int x0 = src[2 * lane], x1 = src[2 * lane + 1];
        reinterpret_cast<uint8_t*>(qv)[2 * lane] = (uint8_t)x0;
        reinterpret_cast<uint8_t*>(qv)[2 * lane + 1] = (uint8_t)x1;
        qnorm = wave_sum_i(x0 * x0 + x1 * x1);
    } else {
        const float* src = a.query_rows ? reinterpret_cast<const float*>(g.rows) + (size_t)a.query_rows[q] * g.ldv
                                        : reinterpret_cast<const float*>(a.queries) + (size_t)q * g.dim;
        float ss = 0.f;
        for (int d = lane; d < g.ldv; d += 64) {
            const float v = d < g.dim ? src[d] : 0.f;
            qv[d] = v;
            ss = fmaf(v, v, ss);
        }
        if (g.normalize_query && !a.query_rows) {  // hnsw_distfunc_opt.cc:160-162 (stored rows are already normalised)
            ss = wave_sum(ss);
            if (ss != 0.0f) {
                const float inv = 1.0f / sqrtf(ss);
                for (int d = lane; d < g.dim; d += 64) qv[d] *= inv;
            }
        }
    }
    if constexpr (!BITSET) {
        for (int i = lane; i < a.table_size; i += 64) table[i] = HT_EMPTY;
    }
    __builtin_amdgcn_wave_barrier();
    const uint8_t* qb = reinterpret_cast<const uint8_t*>(qv);

    int ndc = 0, hops = 0, hops_up = 0, nvisited = 0;
    bool overflow = false;

    // visited test-and-set: true when id was NOT visited before (exact)
    auto visit = [&](uint32_t id) -> bool {
        if constexpr (BITSET) {
            const uint32_t bit = 1u << (id & 31);
            const uint32_t old = atomicOr(&bits[id >> 5], bit);
            return (old & bit) == 0;
        } else {
            uint32_t hsh = (id * 2654435761u) >> a.table_shift;
            const uint32_t mask = (uint32_t)a.table_size - 1u;
            for (int probe = 0; probe < a.table_size; ++probe) {
                const uint32_t old = atomicCAS(&table[hsh], HT_EMPTY, id);
                if (old == HT_EMPTY) return true;
                if (old == id) return false;
                hsh = (hsh + 1) & mask;
            }
            return false;
        }
    };

    if (g.n == 0) {
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
        return;
    }

    // ---- entry point + greedy descent (hnsw_distfunc_opt.cc:168-198) ----
    const int start = a.start_nodes ? a.start_nodes[q] : -1;
    int cur = start >= 0 ? start : g.enterpoint;
    if (lane == 0) nbr[0] = cur;
    __builtin_amdgcn_wave_barrier();
    frontier_distances<SPACE>(g, qv, qb, qnorm, nbr, nd, 1, lane);
    float curdist = nd[0];
    ndc += 1;
    for (int lvl = (start >= 0 ? 0 : g.maxlevel); lvl > a.level; --lvl) {
        bool changed = true;
        while (changed) {
            changed = false;
            const int64_t off = g.up_off[cur] + (int64_t)(lvl - 1) * (g.maxM + 1);
            if (WIDE && g.maxM > 62) {   // upper-level lists longer than one word per lane
                hops_up++;
                ndc += greedy_step_any<SPACE>(g, g.up_links + off, qv, qb, qnorm, nbr, nd, lane, cur, curdist, changed);
                continue;
            }
            const int v = (lane <= g.maxM) ? g.up_links[off + lane] : 0;
            const int cntl = __builtin_amdgcn_readfirstlane(v);
            const int nb = __shfl(v, lane + 1, 64);
            if (lane < cntl) nbr[lane] = nb;
            __builtin_amdgcn_wave_barrier();
            hops_up++;
            if (cntl > 0) {
                frontier_distances<SPACE>(g, qv, qb, qnorm, nbr, nd, cntl, lane);
                ndc += cntl;
                // sequential "if (d < curdist)" scan == first index attaining the minimum
                u64 key = ~0ull;
                if (lane < cntl) key = ((u64)f32_ord(nd[lane]) << 32) | (uint32_t)lane;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const u64 other = __shfl_xor(key, o, 64);
                    key = other < key ? other : key;
                }
                const float dmin = ord_f32((uint32_t)(key >> 32));
                if (dmin < curdist) {
                    curdist = dmin;
                    cur = nbr[(uint32_t)key];
                    changed = true;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

    // ---- level 0 (hnsw_distfunc_opt.cc:200-274) ----
    int n = 1;
    if (lane == 0) {
        keys[0] = curdist;
        idu[0] = cur;
    }
    {
        bool fresh = false;
        if (lane == 0) fresh = visit((uint32_t)cur);
        (void)fresh;
        nvisited = 1;
    }
    __builtin_amdgcn_wave_barrier();

    // cur: every item before it is used (the reference's currElem).  pre_*: adjacency of the item
    // that will most likely be expanded next, requested one expansion early (its latency hides
    // behind this expansion's gather); a wrong guess only costs the normal load.
    // adjacency word `lane` of node c on the level being searched ([count][ids...])
    auto load_adj = [&](int c) -> int {
        if (a.level == 0) return (lane <= g.maxM0) ? g.links0[(size_t)c * (g.maxM0 + 1) + lane] : 0;
        return (lane <= g.maxM) ? g.up_links[g.up_off[c] + (int64_t)(a.level - 1) * (g.maxM + 1) + lane] : 0;
    };
    int cursor = 0;
    int pre_node = -1, pre_v = 0, pre2_node = -1, pre2_v = 0;
    float pre_key = INFINITY;
    bool pre_ok = false;            // pre_node is the first unused item of the array, at a position below ef
    // Software pipeline across expansions: when the NEXT node to expand is already certain before the accepted items of
    // this expansion are merged in (it is the array's next unused item and every accepted key is larger), its visited
    // filter runs and its row gather is issued first, and the merge executes while those rows are in flight.
    int pipe_node = -1, pipe_m = 0;
    FrontierLoads<SPACE> fl;
    long long pc[6] = {0, 0, 0, 0, 0, 0};
    long long pt = a.prof ? (long long)__builtin_readcyclecounter() : 0;
    auto lap = [&](int ph) __attribute__((always_inline)) {
        if (a.prof) {
            const long long now = (long long)__builtin_readcyclecounter();
            pc[ph] += now - pt;
            pt = now;
        }
    };
    lap(0);
    while (true) {
        const int lim = n < a.ef ? n : a.ef;
        // first unused item at or after cur
        int fu = n;
        for (int base = cursor; base < n && fu == n; base += 64) {
            const int i = base + lane;
            const u64 mk = __ballot(i < n && idu[i] >= 0);
            if (mk) fu = base + (__ffsll((long long)mk) - 1);
        }
        if (fu >= lim) break;
        const int c = idu[fu] & 0x7FFFFFFF;
        if (lane == 0) idu[fu] |= (int)0x80000000;
        cursor = fu + 1;
        hops++;
        const float topKey = keys[n - 1];
        const int size0 = n;

        int m;
        if (c == pipe_node) {
            // visited filter done and rows requested during the previous expansion; the guess for the expansion after
            // this one was made there too
            m = pipe_m;
            pipe_node = -1;
            lap(1);
            lap(2);
        } else {
            // adjacency of c: [count][ids...]
            int v;
            if (c == pre_node) v = pre_v;
            else if (c == pre2_node) v = pre2_v;
            else v = load_adj(c);
            // Adjacency of the NEXT expansion, requested early so that its latency (a random HBM read under
            // load: ~2 us) hides behind this expansion's gather and inserts.  Two guesses: (A) now, the next
            // unused item of the array -- right unless this expansion finds something closer; (B) after the
            // distances, the best newly accepted item when it is closer than (A).
            const int cntn = __builtin_amdgcn_readfirstlane(v);
            lap(1);
            const int nb = __shfl(v, lane + 1, 64);   // neighbours 0..62 (list words 1..63)
            bool isn = false;
            if (lane < cntn && (!WIDE || lane < 63)) isn = visit((uint32_t)nb);
            const u64 nmask = __ballot(isn);
            m = __popcll(nmask);
            if (isn) nbr[__popcll(nmask & ((1ull << lane) - 1ull))] = nb;
            if (WIDE) {
                // wide level-0 lists (maxM0 > 62, i.e. M >= 32): neighbours 63.. are list words 64.., read on demand in chunks
                // of 64 (round 3: any number of chunks the frontier arrays hold -- a.nbcap)
                for (int c0 = 63; c0 < cntn; c0 += 64) {
                    int nb2 = 0;
                    if (c0 + 1 + lane <= g.maxM0) nb2 = g.links0[(size_t)c * (g.maxM0 + 1) + c0 + 1 + lane];
                    bool isn2 = false;
                    if (c0 + lane < cntn) isn2 = visit((uint32_t)nb2);
                    const u64 nmask2 = __ballot(isn2);
                    if (isn2) nbr[m + __popcll(nmask2 & ((1ull << lane) - 1ull))] = nb2;
                    m += __popcll(nmask2);
                }
            }
            nvisited += m;
            if (!BITSET && nvisited > (a.table_size - (a.table_size >> 3))) {
                overflow = true;  // visited table nearly full: give up, the bitset variant re-runs the query
                break;
            }
            __builtin_amdgcn_wave_barrier();
            lap(2);
            // (issued only now: v had to be waited for first, and vmcnt retires in order -- a younger
            //  outstanding load in front of that wait would put its whole latency into it)
            {
                int fu2 = n;
                for (int base = cursor; base < n && fu2 == n; base += 64) {
                    const int i = base + lane;
                    const u64 mk = __ballot(i < n && idu[i] >= 0);
                    if (mk) fu2 = base + (__ffsll((long long)mk) - 1);
                }
                if (fu2 < lim) {
                    pre_node = idu[fu2] & 0x7FFFFFFF;
                    pre_key = keys[fu2];
                    pre_v = load_adj(pre_node);
                    pre_ok = true;
                } else {
                    pre_node = -1;
                    pre_key = INFINITY;
                    pre_ok = false;
                }
                pre2_node = -1;
            }
            if (m > 0) frontier_issue<SPACE>(fl, g, nbr, m, lane);
        }
        if (m == 0) continue;
        ndc += m;
        frontier_finish<SPACE>(fl, g, qv, qb, qnorm, nbr, nd, m, lane);
        lap(3);

        // (more than 64 new rows only with wide lists: rounds of 64; without equal keys the final array does
        //  not depend on the order in which accepted items are merged in)
        for (int r0 = 0; r0 < (WIDE ? m : 1); r0 += 64) {  // one trip unless the lists are wide
            // accept d < topKey || size < ef   (:240)
            float dj = INFINITY;
            int idj = -1;
            bool acc = false;
            if (r0 + lane < m) {
                dj = nd[r0 + lane];
                idj = nbr[r0 + lane];
                acc = (dj < topKey) || (size0 < a.ef);
            }
            const u64 amask = __ballot(acc);
            const int m2 = __popcll(amask);
            if (m2 == 0) continue;
            // ascending order of the accepted items (std::sort, :251); ties keep list order
            int rank = 0;
            for (u64 mm = amask; mm;) {
                const int j = __ffsll((long long)mm) - 1;
                mm &= mm - 1;
                // (j comes from the ballot mask, so it is wave-uniform: v_readlane, not an LDS-routed shuffle)
                const float dother = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dj), j));
                rank += (dother < dj || (dother == dj && j < lane)) ? 1 : 0;
            }
            __builtin_amdgcn_wave_barrier();
            if (acc) {
                sk[rank] = dj;
                si[rank] = idj;
            }
            __builtin_amdgcn_wave_barrier();

            if (sk[0] <= pre_key) {  // guess (B)
                pre2_node = si[0];
                pre2_v = load_adj(pre2_node);
            } else if (!WIDE && pre_ok && !a.no_pipe) {
                // Every accepted key is larger than the key of the array's next unused item: that item (pre_node,
                // adjacency already here) IS the next expansion, whatever the merge below does to the positions
                // behind it.  Run its visited filter now and request its rows; the merge overlaps the gather.
                const int cntp = __builtin_amdgcn_readfirstlane(pre_v);
                const int nbp = __shfl(pre_v, lane + 1, 64);
                bool isp = false;
                if (lane < cntp) isp = visit((uint32_t)nbp);
                const u64 pmask = __ballot(isp);
                const int mp = __popcll(pmask);
                __builtin_amdgcn_wave_barrier();  // (nbr/nd of this expansion were consumed into registers above)
                if (isp) nbr[__popcll(pmask & ((1ull << lane) - 1ull))] = nbp;
                nvisited += mp;
                if (!BITSET && nvisited > (a.table_size - (a.table_size >> 3))) {
                    overflow = true;
                    break;
                }
                __builtin_amdgcn_wave_barrier();
                pipe_node = pre_node;
                pipe_m = mp;
                // guess for the expansion after that: the second unused item of the array, or the best accepted
                // key if that is closer (then its position is only known after the merge: no pipelining on it)
                int second = n;
                {
                    int seen = 0;
                    for (int base = cursor; base < n && second == n; base += 64) {
                        const int i = base + lane;
                        u64 mk = __ballot(i < n && idu[i] >= 0);
                        if (seen == 0 && mk) {
                            mk &= mk - 1;
                            seen = 1;
                        }
                        if (mk) second = base + (__ffsll((long long)mk) - 1);
                    }
                }
                const float k2 = second < n ? keys[second] : INFINITY;
                if (second < n && second < a.ef && k2 < sk[0]) {
                    pre_node = idu[second] & 0x7FFFFFFF;
                    pre_key = k2;
                    pre_ok = true;
                    pre_v = load_adj(pre_node);
                } else if (sk[0] < k2) {
                    pre_node = si[0];
                    pre_key = sk[0];
                    pre_ok = false;
                    pre_v = load_adj(pre_node);
                } else {
                    pre_node = -1;
                    pre_key = INFINITY;
                    pre_ok = false;
                }
                pre2_node = -1;
                if (mp > 0) frontier_issue<SPACE>(fl, g, nbr, mp, lane);
            }
            lap(4);
            // All accepted items at once when no two keys involved are equal (the normal case): the result of
            // the reference's sequential push_or_replace_non_empty_exp calls is then the merge of the two sorted
            // sequences cut at the capacity, so every old item moves up by the number of new keys below it and
            // new item t lands at (#old keys below it) + t; the scan cursor rewinds to the first new position
            // (hnsw_distfunc_opt.cc:261-266).  Equal keys (rare) take the exact sequential replay below.
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
            // (the shuffle runs in ALL lanes: under the short-circuit `&&` it sat in a divergent branch, lane m2-2 read
            //  its inactive neighbour as 0 and a tie between the two largest accepted keys went unnoticed -- found in
            //  round 3 by tests/test_gpu_hnsw_mw.py against the oracle)
            const float next_key = __shfl_down(mykey, 1, 64);
            tie |= (lane + 1 < m2) & (mykey == next_key);
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
                        // insertion index.  Without a key equal to the new one in the array, the reference's
                        // exponential probe + lower_bound (sort_arr_bi.h:172-186) is simply the number of
                        // smaller keys: one parallel count.  With equal keys present (rare) the probe is
                        // replayed so the item lands inside the run exactly where the reference puts it.
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
        }
        if (overflow) break;  // (set inside the round loop by the pipelined visited filter)
        lap(5);
    }
    if (a.prof && lane == 0) {
        for (int i = 0; i < 6; ++i) atomicAdd(&g_hnsw_prof[i], (unsigned long long)pc[i]);
        atomicAdd(&g_hnsw_prof[6], 1ull);
    }

    // ---- results: first k items, ties ordered by internal id (KNNQueue holds
    //      pair<dist, Object*> and data_rearranged_ addresses grow with the id) ----
    const int kk = overflow ? 0 : (a.k < n ? a.k : n);
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
        if (a.status) a.status[q] = overflow ? 1 : 0;
        if (!BITSET && overflow && a.fix_list) a.fix_list[atomicAdd(a.fix_count, 1)] = q;
    }
}

template <int SPACE, bool BITSET, int SA_EMAX, bool WIDE>
__global__ __launch_bounds__(64) void hnsw_search_kernel(HnswArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if constexpr (BITSET) {
        if (a.fix_mode) {
            // queries whose LDS visited table filled up in the previous launch, re-run on HBM bitsets: workgroup b
            // owns bitset slot b and clears it before every query it takes from the list
            const int cnt = *a.fix_count;
            uint32_t* bits = a.bitset + (size_t)blockIdx.x * a.bitset_words;
            for (int slot = blockIdx.x; slot < cnt; slot += gridDim.x) {
                for (size_t i = threadIdx.x; i < a.bitset_words; i += 64) bits[i] = 0u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                hnsw_search_body<SPACE, true, SA_EMAX, WIDE>(a, smem, a.fix_list[slot], bits);
                __builtin_amdgcn_wave_barrier();
            }
            return;
        }
        hnsw_search_body<SPACE, true, SA_EMAX, WIDE>(a, smem, blockIdx.x, a.bitset + (size_t)blockIdx.x * a.bitset_words);
    } else {
        hnsw_search_body<SPACE, false, SA_EMAX, WIDE>(a, smem, blockIdx.x, nullptr);
    }
}

// ---------------------------------------------------------------------------------------
// Hnsw::SearchOld (src/method/hnsw_distfunc_opt.cc:46-150; generic twin baseSearchAlgorithmOld, hnsw.cc:1083-1172):
// the reference's search for algoType=old and, in hybrid mode, for ef >= 1000 (hnsw.cc:724).  Item for item:
//   candidateQueuei  - std::priority_queue on -distance.  Its pop order among EQUAL keys depends on the binary-heap
//                      layout, so the heap is kept as a real array heap with libstdc++'s push_heap / pop_heap moves
//                      (__adjust_heap walks the larger child down to a leaf, then __push_heap lifts the value);
//   closestDistQueuei- only its top key and its size are ever observed, so it is kept as the sorted array of its key
//                      VALUES: neighbour j of an expansion is accepted iff fewer than ef values of (queue + earlier
//                      neighbours of this expansion) are <= d_j -- the closed form of the sequential
//                      "top > d || size < ef" test (:126), evaluated for all neighbours at once;
//   query result     - KNNQueue (knnquery.cc:66-75, knnqueue.h:55-64): strict "d < top", eviction of the largest
//                      (distance, object address = internal id) pair: a sorted (key, id) array, sequential inserts.
// No capacity limits: arrays that outgrow LDS live in a per-query HBM workspace (generic pointers).  One wave per
// query; the frontier gather is the same 8-lanes-per-row kernel as V1Merge.
// ---------------------------------------------------------------------------------------
struct OldWs {
    int capA, capR;          // ef, k
    int heap_lds;            // heap entries kept in LDS (the top levels); the rest spills to heap_hbm
    int heap_cap;            // total heap capacity per query; exceeding it sets status 2 (host retries with n)
    int a_in_lds, r_in_lds;
    float* a_hbm;            // [nq][capA]      when !a_in_lds
    u64* r_hbm;              // [nq][capR]      when !r_in_lds
    u64* heap_hbm;           // [nq][heap_cap - heap_lds]
};

// the arrays below may live in HBM (generic pointers): make one lane's stores visible to the other lanes of the wave
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ u64 pack_kid(float key, int id) { return ((u64)__float_as_uint(key) << 32) | (uint32_t)id; }
__device__ __forceinline__ float kid_key(u64 v) { return __uint_as_float((uint32_t)(v >> 32)); }
__device__ __forceinline__ int kid_id(u64 v) { return (int)(uint32_t)v; }

template <int SPACE, bool BITSET, bool WIDE>
__global__ __launch_bounds__(64) void hnsw_search_old_kernel(HnswArgs a, OldWs w) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const HnswDeviceGraph& g = a.g;
    const int q = blockIdx.x, lane = threadIdx.x;
    constexpr bool kU8 = DistTraits<SPACE>::kU8;
    const int nbcap = WIDE ? a.nbcap : 64;   // (wide lists: a multiple of 64 that holds maxM0 and maxM)

    // ---- LDS carve-up ----
    const int qfloats = kU8 ? 32 : g.ldv;
    float* qv = reinterpret_cast<float*>(smem);                          // [ldv]
    int* nbr = reinterpret_cast<int*>(qv + qfloats);                      // [nbcap]
    float* nd = reinterpret_cast<float*>(nbr + nbcap);                    // [nbcap]
    float* sk = nd + nbcap;                                               // [64] accepted keys, sorted
    u64* heap_l = reinterpret_cast<u64*>(sk + 64);                        // [heap_lds]
    float* a_l = reinterpret_cast<float*>(heap_l + w.heap_lds);           // [capA] when in LDS
    u64* r_l = reinterpret_cast<u64*>(a_l + (w.a_in_lds ? ((w.capA + 1) & ~1) : 0));  // [capR] when in LDS
    uint32_t* table = reinterpret_cast<uint32_t*>(r_l + (w.r_in_lds ? w.capR : 0));   // [table_size]
    uint32_t* bits = BITSET ? a.bitset + (size_t)q * a.bitset_words : nullptr;
    float* A = w.a_in_lds ? a_l : w.a_hbm + (size_t)q * w.capA;
    u64* R = w.r_in_lds ? r_l : w.r_hbm + (size_t)q * w.capR;
    u64* heap_g = w.heap_hbm + (size_t)q * (size_t)(w.heap_cap - w.heap_lds);
    auto hget = [&](int i) -> u64 { return i < w.heap_lds ? heap_l[i] : heap_g[i - w.heap_lds]; };
    auto hset = [&](int i, u64 v) {
        if (i < w.heap_lds) heap_l[i] = v;
        else heap_g[i - w.heap_lds] = v;
    };

    // ---- stage the query (as the V1Merge kernel) ----
    int qnorm = 0;
    if constexpr (kU8) {
        const uint8_t* src = reinterpret_cast<const uint8_t*>(a.queries) + (size_t)q * 128;
        const int x0 = src[2 * lane], x1 = src[2 * lane + 1];
        reinterpret_cast<uint8_t*>(qv)[2 * lane] = (uint8_t)x0;
        reinterpret_cast<uint8_t*>(qv)[2 * lane + 1] = (uint8_t)x1;
        qnorm = wave_sum_i(x0 * x0 + x1 * x1);
    } else {
        const float* src = reinterpret_cast<const float*>(a.queries) + (size_t)q * g.dim;
        float ss = 0.f;
        for (int d = lane; d < g.ldv; d += 64) {
            const float v = d < g.dim ? src[d] : 0.f;
            qv[d] = v;
            ss = fmaf(v, v, ss);
        }
        if (g.normalize_query) {  // hnsw_distfunc_opt.cc:55-57
            ss = wave_sum(ss);
            if (ss != 0.0f) {
                const float inv = 1.0f / sqrtf(ss);
                for (int d = lane; d < g.dim; d += 64) qv[d] *= inv;
            }
        }
    }
    if constexpr (!BITSET) {
        for (int i = lane; i < a.table_size; i += 64) table[i] = HT_EMPTY;
    }
    __builtin_amdgcn_wave_barrier();
    const uint8_t* qb = reinterpret_cast<const uint8_t*>(qv);

    auto visit = [&](uint32_t id) -> bool {
        if constexpr (BITSET) {
            const uint32_t bit = 1u << (id & 31);
            const uint32_t old = atomicOr(&bits[id >> 5], bit);
            return (old & bit) == 0;
        } else {
            uint32_t hsh = (id * 2654435761u) >> a.table_shift;
            const uint32_t mask = (uint32_t)a.table_size - 1u;
            for (int probe = 0; probe < a.table_size; ++probe) {
                const uint32_t old = atomicCAS(&table[hsh], HT_EMPTY, id);
                if (old == HT_EMPTY) return true;
                if (old == id) return false;
                hsh = (hsh + 1) & mask;
            }
            return false;
        }
    };

    int ndc = 0, hops = 0, hops_up = 0, nvisited = 0, status = 0;
    if (g.n == 0) {
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
        return;
    }

    // ---- entry point + greedy descent (:64-92) ----
    int cur = g.enterpoint;
    if (lane == 0) nbr[0] = cur;
    __builtin_amdgcn_wave_barrier();
    frontier_distances<SPACE>(g, qv, qb, qnorm, nbr, nd, 1, lane);
    float curdist = nd[0];
    ndc += 1;
    for (int lvl = g.maxlevel; lvl > 0; --lvl) {
        bool changed = true;
        while (changed) {
            changed = false;
            const int64_t off = g.up_off[cur] + (int64_t)(lvl - 1) * (g.maxM + 1);
            if (WIDE && g.maxM > 62) {   // upper-level lists of any length
                hops_up++;
                ndc += greedy_step_any<SPACE>(g, g.up_links + off, qv, qb, qnorm, nbr, nd, lane, cur, curdist, changed);
                continue;
            }
            const int v = (lane <= g.maxM) ? g.up_links[off + lane] : 0;
            const int cntl = __builtin_amdgcn_readfirstlane(v);
            const int nb = __shfl(v, lane + 1, 64);
            if (lane < cntl) nbr[lane] = nb;
            __builtin_amdgcn_wave_barrier();
            hops_up++;
            if (cntl > 0) {
                frontier_distances<SPACE>(g, qv, qb, qnorm, nbr, nd, cntl, lane);
                ndc += cntl;
                u64 key = ~0ull;
                if (lane < cntl) key = ((u64)f32_ord(nd[lane]) << 32) | (uint32_t)lane;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const u64 other = __shfl_xor(key, o, 64);
                    key = other < key ? other : key;
                }
                const float dmin = ord_f32((uint32_t)(key >> 32));
                if (dmin < curdist) {
                    curdist = dmin;
                    cur = nbr[(uint32_t)key];
                    changed = true;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

    // ---- level 0 (:94-147) ----
    int nA = 1, nR = 1, hn = 1;
    if (lane == 0) {
        A[0] = curdist;
        R[0] = pack_kid(curdist, cur);
        hset(0, pack_kid(curdist, cur));
        (void)visit((uint32_t)cur);
    }
    nvisited = 1;
    wave_sync();

    // std::push_heap of (key, id) at the end of the candidate heap.  comp(parent, value) on -distance
    // <=> parent.key > value.key: such parents move down.  Lane L looks at ancestor L of the new slot.
    auto heap_push = [&](float key, int id) {
        const int hole = hn;
        hn++;
        const int j = hole + 1;  // 1-based
        const int jl = lane < 31 ? (j >> lane) : 0;  // 1-based index of ancestor `lane` levels up (0 = none)
        const bool has = lane >= 1 && jl >= 1;
        const int anc = has ? jl - 1 : 0;
        const u64 av = has ? hget(anc) : 0ull;
        const bool moves = has && kid_key(av) > key;
        // first lane >= 1 whose ancestor does not move (or does not exist) ends the walk
        const u64 stop = __ballot(!moves) & ~1ull;
        const int sL = __ffsll((long long)stop) - 1;  // >= 1 (lane 63 never has an ancestor for heaps < 2^62)
        if (lane >= 1 && lane < sL) hset((j >> (lane - 1)) - 1, av);
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) hset((j >> (sL - 1)) - 1, pack_kid(key, id));
        wave_sync();
    };
    // std::pop_heap + pop_back (libstdc++ __adjust_heap): the hole walks down along the larger child (on -distance:
    // the child with the SMALLER key; equal keys -> the right child) to a leaf, then the last element is pushed up.
    auto heap_pop = [&]() {
        if (hn <= 1) {
            hn = 0;
            return;
        }
        const int len = hn - 1;
        const u64 val = hget(len);
        int hole = 0, child = 0;
        while (child < (len - 1) / 2) {
            child = 2 * (child + 1);
            const u64 cr = hget(child), cl = hget(child - 1);
            u64 cv = cr;
            if (kid_key(cr) > kid_key(cl)) {
                child--;
                cv = cl;
            }
            if (lane == 0) hset(hole, cv);
            hole = child;
        }
        if ((len & 1) == 0 && child == (len - 2) / 2) {
            child = 2 * (child + 1);
            if (lane == 0) hset(hole, hget(child - 1));
            hole = child - 1;
        }
        wave_sync();
        // __push_heap(first, hole, top = 0, val)
        hn = hole;  // heap_push appends at `hn`
        heap_push(kid_key(val), kid_id(val));
        hn = len;
    };

    while (hn > 0) {
        const u64 top = hget(0);
        const float lower = A[nA - 1];
        if (kid_key(top) > lower) break;  // :107-109
        heap_pop();
        const int c = kid_id(top);
        hops++;
        // adjacency [count][ids...]
        int m;
        if constexpr (WIDE) {   // lists of any length, chunks of 64 neighbours
            m = collect_unvisited_any(g.links0 + (size_t)c * (g.maxM0 + 1), nbr, lane, visit);
        } else {
            const int v = (lane <= g.maxM0) ? g.links0[(size_t)c * (g.maxM0 + 1) + lane] : 0;
            const int cntn = __builtin_amdgcn_readfirstlane(v);
            const int nb = __shfl(v, lane + 1, 64);
            bool isn = false;
            if (lane < cntn) isn = visit((uint32_t)nb);
            const u64 nmask = __ballot(isn);
            m = __popcll(nmask);
            if (isn) nbr[__popcll(nmask & ((1ull << lane) - 1ull))] = nb;
        }
        nvisited += m;
        if (!BITSET && nvisited > (a.table_size - (a.table_size >> 3))) {
            status = 1;  // visited table nearly full: the host re-runs with a bitset
            break;
        }
        __builtin_amdgcn_wave_barrier();
        if (m == 0) continue;
        ndc += m;
        frontier_distances<SPACE>(g, qv, qb, qnorm, nbr, nd, m, lane);

        for (int r0 = 0; r0 < m; r0 += 64) {  // list order; one trip unless the lists are wide
            const bool valid = r0 + lane < m;
            const float dj = valid ? nd[r0 + lane] : INFINITY;
            const int idj = valid ? nbr[r0 + lane] : -1;
            // #{a in A : a <= dj}: upper bound by bisection
            int lo = 0, hi = nA;
            while (__any(lo < hi)) {
                const int mid = (lo + hi) >> 1;
                const float av = (lo < hi) ? A[mid] : 0.f;
                if (lo < hi) {
                    if (av <= dj) lo = mid + 1;
                    else hi = mid;
                }
            }
            int cnt = lo;
            u64 vmask = __ballot(valid);
            for (u64 mm = vmask; mm;) {
                const int j = __ffsll((long long)mm) - 1;
                mm &= mm - 1;
                const float dother = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dj), j));
                cnt += (j < lane && dother <= dj) ? 1 : 0;
            }
            const bool acc = valid && cnt <= a.ef - 1;
            const u64 amask = __ballot(acc);
            const int m2 = __popcll(amask);
            if (m2 == 0) continue;

            // --- closestDistQueue values: merge the accepted keys, keep the ef smallest ---
            int rank = 0;
            for (u64 mm = amask; mm;) {
                const int j = __ffsll((long long)mm) - 1;
                mm &= mm - 1;
                const float dother = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dj), j));
                rank += (dother < dj || (dother == dj && j < lane)) ? 1 : 0;
            }
            __builtin_amdgcn_wave_barrier();
            if (acc) sk[rank] = dj;
            __builtin_amdgcn_wave_barrier();
            const int newn = nA + m2 < a.ef ? nA + m2 : a.ef;
            // old element i moves up by the number of new keys strictly below it; chunks from the top down so
            // that a chunk is read before anything lands on it
            for (int base = ((nA - 1) / 64) * 64; base >= 0; base -= 64) {
                const int i = base + lane;
                float av = 0.f;
                int sh = 0;
                if (i < nA) {
                    av = A[i];
                    int l2 = 0, h2 = m2;
                    while (l2 < h2) {  // #{sk < av}
                        const int mid = (l2 + h2) >> 1;
                        if (sk[mid] < av) l2 = mid + 1;
                        else h2 = mid;
                    }
                    sh = l2;
                }
                wave_sync();
                if (i < nA && sh > 0 && i + sh < newn) A[i + sh] = av;
                wave_sync();
            }
            // new key: behind the old keys <= it (`lo`, counted before the moves) plus its rank among the accepted
            __builtin_amdgcn_wave_barrier();
            if (acc) {
                const int np = lo + rank;
                if (np < newn) A[np] = dj;
            }
            nA = newn;
            wave_sync();

            // --- candidate heap + result queue: accepted neighbours in list order (:127-131) ---
            for (u64 mm = amask; mm;) {
                const int j = __ffsll((long long)mm) - 1;
                mm &= mm - 1;
                const float key = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dj), j));
                const int id = __builtin_amdgcn_readlane(idj, j);
                if (hn >= w.heap_cap) {
                    status = 2;
                    break;
                }
                heap_push(key, id);
                // KNNQuery::CheckAndAddToResult
                const bool full = nR >= a.k;
                if (full && !(key < kid_key(R[nR - 1]))) continue;
                // position by (key, id): pair<dist, Object*> ordering
                int p = 0;
                for (int base = 0; base < nR; base += 64) {
                    const int i = base + lane;
                    bool less = false;
                    if (i < nR) {
                        const u64 rv = R[i];
                        const float rk = kid_key(rv);
                        less = rk < key || (rk == key && kid_id(rv) < id);
                    }
                    p += __popcll(__ballot(less));
                }
                const int newr = full ? nR : nR + 1;
                for (int base = ((newr - 1) / 64) * 64; base >= 0; base -= 64) {
                    const int i = base + lane;
                    u64 rv = 0;
                    const bool mv = i > p && i < newr;
                    if (mv) rv = R[i - 1];
                    wave_sync();
                    if (mv) R[i] = rv;
                    wave_sync();
                }
                if (lane == 0) R[p] = pack_kid(key, id);
                nR = newr;
                wave_sync();
            }
            if (status) break;
        }
        if (status) break;
    }

    const int kk = status ? 0 : nR;
    for (int i = lane; i < a.k; i += 64) {
        if (i < kk) {
            const u64 rv = R[i];
            const int id = kid_id(rv);
            a.out_ids[(size_t)q * a.k + i] = g.ext_ids ? g.ext_ids[id] : id;
            a.out_dists[(size_t)q * a.k + i] = kid_key(rv);
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
        if (a.status) a.status[q] = status;
    }
}

// ---------------------------------------------------------------------------------------
// SearchV1Merge with a sorted array beyond the LDS kernels' 1024 items (round 3; only reachable with an explicit
// algoType=v1merge and efSearch > 1024, or k > 1024 below efSearch 1000: hybrid mode runs SearchOld from 1000 on).
// The reference sizes SortArrBI to max(ef, k) whatever that is (hnsw_distfunc_opt.cc:152-167).  Same algorithm, item for
// item; the array (keys, id | used << 31) lives in a per-query HBM workspace, the visited set is an HBM bitset, and the
// accepted items of an expansion are inserted one after the other with SortArrBI::push_or_replace_non_empty_exp
// (sort_arr_bi.h:159-199) -- no attempt at speed: one wave per query, chunks of 64.
// ---------------------------------------------------------------------------------------
template <int SPACE, bool WIDE>
__global__ __launch_bounds__(64) void hnsw_search_big_kernel(HnswArgs a, float* ws_keys, int* ws_idu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const HnswDeviceGraph& g = a.g;
    const int q = blockIdx.x, lane = threadIdx.x;
    constexpr bool kU8 = DistTraits<SPACE>::kU8;
    const int nbcap = WIDE ? a.nbcap : 64;   // (wide lists: a multiple of 64 that holds maxM0 and maxM)
    const int qfloats = kU8 ? 32 : g.ldv;
    float* qv = reinterpret_cast<float*>(smem);
    int* nbr = reinterpret_cast<int*>(qv + qfloats);
    float* nd = reinterpret_cast<float*>(nbr + nbcap);
    float* sk = nd + nbcap;
    int* si = reinterpret_cast<int*>(sk + 64);
    float* keys = ws_keys + (size_t)q * a.cap;
    int* idu = ws_idu + (size_t)q * a.cap;
    uint32_t* bits = a.bitset + (size_t)q * a.bitset_words;

    int qnorm = 0;
    if constexpr (kU8) {
        const uint8_t* src = reinterpret_cast<const uint8_t*>(a.queries) + (size_t)q * 128;
        const int x0 = src[2 * lane], x1 = src[2 * lane + 1];
        reinterpret_cast<uint8_t*>(qv)[2 * lane] = (uint8_t)x0;
        reinterpret_cast<uint8_t*>(qv)[2 * lane + 1] = (uint8_t)x1;
        qnorm = wave_sum_i(x0 * x0 + x1 * x1);
    } else {
        const float* src = reinterpret_cast<const float*>(a.queries) + (size_t)q * g.dim;
        float ss = 0.f;
        for (int d = lane; d < g.ldv; d += 64) {
            const float v = d < g.dim ? src[d] : 0.f;
            qv[d] = v;
            ss = fmaf(v, v, ss);
        }
        if (g.normalize_query) {
            ss = wave_sum(ss);
            if (ss != 0.0f) {
                const float inv = 1.0f / sqrtf(ss);
                for (int d = lane; d < g.dim; d += 64) qv[d] *= inv;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    const uint8_t* qb = reinterpret_cast<const uint8_t*>(qv);
    auto visit = [&](uint32_t id) -> bool {
        const uint32_t bit = 1u << (id & 31);
        return (atomicOr(&bits[id >> 5], bit) & bit) == 0;
    };
    int ndc = 0, hops = 0, hops_up = 0;
    if (g.n == 0) {
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
        return;
    }
    // ---- entry point + greedy descent (hnsw_distfunc_opt.cc:168-198) ----
    int cur = g.enterpoint;
    if (lane == 0) nbr[0] = cur;
    __builtin_amdgcn_wave_barrier();
    frontier_distances<SPACE>(g, qv, qb, qnorm, nbr, nd, 1, lane);
    float curdist = nd[0];
    ndc += 1;
    for (int lvl = g.maxlevel; lvl > 0; --lvl) {
        bool changed = true;
        while (changed) {
            changed = false;
            const int64_t off = g.up_off[cur] + (int64_t)(lvl - 1) * (g.maxM + 1);
            if (WIDE && g.maxM > 62) {   // upper-level lists of any length
                hops_up++;
                ndc += greedy_step_any<SPACE>(g, g.up_links + off, qv, qb, qnorm, nbr, nd, lane, cur, curdist, changed);
                continue;
            }
            const int v = (lane <= g.maxM) ? g.up_links[off + lane] : 0;
            const int cntl = __builtin_amdgcn_readfirstlane(v);
            const int nb = __shfl(v, lane + 1, 64);
            if (lane < cntl) nbr[lane] = nb;
            __builtin_amdgcn_wave_barrier();
            hops_up++;
            if (cntl > 0) {
                frontier_distances<SPACE>(g, qv, qb, qnorm, nbr, nd, cntl, lane);
                ndc += cntl;
                u64 key = ~0ull;
                if (lane < cntl) key = ((u64)f32_ord(nd[lane]) << 32) | (uint32_t)lane;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const u64 other = __shfl_xor(key, o, 64);
                    key = other < key ? other : key;
                }
                const float dmin = ord_f32((uint32_t)(key >> 32));
                if (dmin < curdist) {
                    curdist = dmin;
                    cur = nbr[(uint32_t)key];
                    changed = true;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    // ---- level 0 (hnsw_distfunc_opt.cc:200-274) ----
    int n = 1, cursor = 0;
    if (lane == 0) {
        keys[0] = curdist;
        idu[0] = cur;
        (void)visit((uint32_t)cur);
    }
    wave_sync();
    while (true) {
        const int lim = n < a.ef ? n : a.ef;
        int fu = n;
        for (int base = cursor; base < lim && fu == n; base += 64) {
            const int i = base + lane;
            const u64 mk = __ballot(i < n && idu[i] >= 0);
            if (mk) fu = base + (__ffsll((long long)mk) - 1);
        }
        if (fu >= lim) break;
        const int c = idu[fu] & 0x7FFFFFFF;
        wave_sync();
        if (lane == 0) idu[fu] |= (int)0x80000000;
        cursor = fu + 1;
        hops++;
        const float topKey = keys[n - 1];
        const int size0 = n;
        wave_sync();
        // adjacency [count][ids...] -> unvisited neighbours
        int m;
        if constexpr (WIDE) {   // lists of any length, chunks of 64 neighbours
            m = collect_unvisited_any(g.links0 + (size_t)c * (g.maxM0 + 1), nbr, lane, visit);
        } else {
            const int v = (lane <= g.maxM0) ? g.links0[(size_t)c * (g.maxM0 + 1) + lane] : 0;
            const int cntn = __builtin_amdgcn_readfirstlane(v);
            const int nb = __shfl(v, lane + 1, 64);
            bool isn = false;
            if (lane < cntn) isn = visit((uint32_t)nb);
            const u64 nmask = __ballot(isn);
            m = __popcll(nmask);
            if (isn) nbr[__popcll(nmask & ((1ull << lane) - 1ull))] = nb;
        }
        __builtin_amdgcn_wave_barrier();
        if (m == 0) continue;
        ndc += m;
        frontier_distances<SPACE>(g, qv, qb, qnorm, nbr, nd, m, lane);
        for (int r0 = 0; r0 < m; r0 += 64) {
            float dj = INFINITY;
            int idj = -1;
            bool acc = false;
            if (r0 + lane < m) {
                dj = nd[r0 + lane];
                idj = nbr[r0 + lane];
                acc = (dj < topKey) || (size0 < a.ef);
            }
            const u64 amask = __ballot(acc);
            const int m2 = __popcll(amask);
            if (m2 == 0) continue;
            int rank = 0;
            for (u64 mm = amask; mm;) {
                const int j = __ffsll((long long)mm) - 1;
                mm &= mm - 1;
                const float dother = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dj), j));
                rank += ((dother < dj) | ((dother == dj) & (j < lane))) ? 1 : 0;
            }
            __builtin_amdgcn_wave_barrier();
            if (acc) {
                sk[rank] = dj;
                si[rank] = idj;
            }
            __builtin_amdgcn_wave_barrier();
            for (int t = 0; t < m2; ++t) {   // push_or_replace_non_empty_exp, in ascending order (:251-266)
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
                    wave_sync();
                    continue;
                }
                int less = 0, leq = 0;
                for (int base = 0; base < n; base += 64) {
                    const int i = base + lane;
                    const float kv = i < n ? keys[i] : INFINITY;
                    less += __popcll(__ballot(kv < key));
                    leq += __popcll(__ballot(kv <= key));
                }
                int p = less;
                if (leq != less) {   // an equal key in the array: replay the probe (:172-186)
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
                // shift [p, newn - 1) up by one: chunks from the top down, every chunk read before it is written
                for (int base = ((newn - 1) / 64) * 64; base >= 0 && base + 64 > p; base -= 64) {
                    const int i = base + lane;
                    const bool mv = i > p && i < newn;
                    float kv = 0.f;
                    int iv = 0;
                    if (mv) {
                        kv = keys[i - 1];
                        iv = idu[i - 1];
                    }
                    wave_sync();
                    if (mv) {
                        keys[i] = kv;
                        idu[i] = iv;
                    }
                    wave_sync();
                }
                if (lane == 0) {
                    keys[p] = key;
                    idu[p] = id;
                }
                n = newn;
                if (p < cursor) cursor = p;
                wave_sync();
            }
        }
    }
    wave_sync();
    const int kk = a.k < n ? a.k : n;
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
        if (a.status) a.status[q] = 0;
    }
}

// ---------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------
static int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

HnswSearchPlan hnsw_make_plan(const HnswDeviceGraph& g, int nq, int k, int ef, bool force_bitset) {
    HnswSearchPlan p{};
    p.nq = nq;
    p.k = k;
    p.ef = ef;
    p.cap = ef > k ? ef : k;
    const bool u8 = g.space == SP_L2SQR_SIFT;
    const size_t fixed = (size_t)((p.cap + 3) & ~3) * 8 + (u8 ? 128 : (size_t)g.ldv * 4) + (size_t)(2 * hnsw_nbcap(g) + 2 * 64) * 4;
    // expected visited nodes ~ (maxM0 * expansions); expansions ~ ef.  Size the table for 2x that
    // and never let LDS push residency below 4 waves per CU (160 KB / 4).
    int want = 1 << ilog2((g.maxM0 > 0 ? g.maxM0 : 32) * p.cap * 2);
    if (want < 2048) want = 2048;
    const size_t budget = 40 * 1024 - 64;
    while ((size_t)want * 4 + fixed > budget && want > 2048) want >>= 1;
    // ~18 distance evaluations per unit of ef on 1M-row graphs: beyond half load
    // the exact hash set is replaced by a per-query bitset in HBM
    if (force_bitset || 18 * p.cap > want / 2 || (size_t)want * 4 + fixed > 64 * 1024) {
        p.table_size = 0;
        p.bitset_words = ((size_t)g.n + 31) / 32;
        p.lds_bytes = fixed + 16;
    } else {
        p.table_size = want;
        p.bitset_words = 0;
        p.lds_bytes = fixed + (size_t)want * 4;
    }
    return p;
}

template <int SPACE, int EMAX, bool WIDE>
static hipError_t launch_space_w(const HnswArgs& a, const HnswSearchPlan& p, hipStream_t s) {
    hipError_t e;
    if (p.table_size == 0) {
        auto kern = hnsw_search_kernel<SPACE, true, EMAX, WIDE>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(a.fix_mode ? a.fix_mode : a.nq), dim3(64), p.lds_bytes, s, a);
    } else {
        if constexpr (!WIDE && EMAX <= 4 && SPACE != SP_L2SQR_SIFT) {
            // one workgroup per query (control wave + gather waves): faster than one wave per query at EVERY batch size
            // measured (1M x 128, efS 128: 1 query 0.25 vs 0.43 ms; 1024: 0.28 vs 0.46; 4096: 1.25 vs 1.87; 16384: 4.6 vs
            // 7.1 ms; 1M x 768, 8192 queries: 9.2 vs 10.2 ms).  NMSLIB_HNSW_MW=0 switches it off, NMSLIB_HNSW_MW_MAXQ
            // bounds the batch size it serves (experiments; the results are the same bits either way)
            // (read per launch: the tests switch it inside one process)
            const char* em = getenv("NMSLIB_HNSW_MW");
            const char* eq = getenv("NMSLIB_HNSW_MW_MAXQ");
            const int mode = em ? atoi(em) : 1;
            const int max_nq = eq ? atoi(eq) : 0x7fffffff;
            if (mode && (mode == 2 || a.nq <= max_nq)) {   // (construction mode included: stored rows as queries, any level)
                const hipError_t me = launch_hnsw_search_mw(a, p.lds_bytes + 16, EMAX, s);
                if (me != hipSuccess) return me;
                return hipGetLastError();
            }
        }
        auto kern = hnsw_search_kernel<SPACE, false, EMAX, WIDE>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(a.nq), dim3(64), p.lds_bytes, s, a);
    }
    return hipGetLastError();
}

template <int SPACE, int EMAX>
static hipError_t launch_space_e(const HnswArgs& a, const HnswSearchPlan& p, hipStream_t s) {
    if (a.g.maxM0 > 62 || a.g.maxM > 62) return launch_space_w<SPACE, EMAX, true>(a, p, s);
    return launch_space_w<SPACE, EMAX, false>(a, p, s);
}

template <int SPACE>
static hipError_t launch_space(const HnswArgs& a, const HnswSearchPlan& p, hipStream_t s) {
    if (p.cap <= 128) return launch_space_e<SPACE, 2>(a, p, s);
    if (p.cap <= 256) return launch_space_e<SPACE, 4>(a, p, s);
    return launch_space_e<SPACE, SA_EMAX_MAX>(a, p, s);
}

// overflow-list arguments of the launch in progress on this thread (launch_hnsw_search_fix)
static thread_local int32_t* t_fix_list = nullptr;
static thread_local int32_t* t_fix_count = nullptr;
static thread_local int t_fix_mode = 0;

hipError_t launch_hnsw_search_ex(const HnswDeviceGraph& g, const HnswSearchPlan& p, const void* queries,
                                 const int32_t* query_rows, const int32_t* start_nodes, int level,
                              uint32_t* bitset, int32_t* out_ids, float* out_dists, int32_t* out_cnt,
                              int32_t* out_ndc, int32_t* out_hops, int32_t* out_hops_up,
                              int32_t* status, hipStream_t s) {
    if (p.nq == 0) return hipSuccess;
    if (p.cap > 64 * SA_EMAX_MAX || hnsw_nbcap(g) > HNSW_NBCAP_LDS) return hipErrorInvalidValue;
    HnswArgs a{};
    a.g = g;
    a.nbcap = hnsw_nbcap(g);
    a.queries = queries;
    a.query_rows = query_rows;
    a.start_nodes = start_nodes;
    a.level = level;
    a.bitset = bitset;
    a.bitset_words = p.bitset_words;
    a.out_ids = out_ids;
    a.out_dists = out_dists;
    a.out_cnt = out_cnt;
    a.out_ndc = out_ndc;
    a.out_hops = out_hops;
    a.out_hops_up = out_hops_up;
    a.status = status;
    a.fix_list = t_fix_list;
    a.fix_count = t_fix_count;
    a.fix_mode = (p.table_size == 0) ? t_fix_mode : 0;
    a.no_pipe = getenv("NMSLIB_HNSW_PIPE") ? atoi(getenv("NMSLIB_HNSW_PIPE")) == 0 : 0;
    static const int prof = getenv("NMSLIB_HNSW_PROF") ? atoi(getenv("NMSLIB_HNSW_PROF")) : 0;
    a.prof = (prof && !query_rows) ? 1 : 0;
    a.nq = p.nq;
    a.k = p.k;
    a.ef = p.ef;
    a.cap = p.cap;
    a.capa = (p.cap + 3) & ~3;
    a.table_size = p.table_size;
    a.table_shift = p.table_size ? 32 - ilog2(p.table_size) : 0;
    if (a.prof) {
        hipError_t pe = hipErrorInvalidValue;
        if (g.space == SP_L2SQR) pe = launch_space<SP_L2SQR>(a, p, s);
        else if (g.space == SP_NORMCOS) pe = launch_space<SP_NORMCOS>(a, p, s);
        unsigned long long h[8] = {0};
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_hnsw_prof), sizeof(h));
        unsigned long long z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_hnsw_prof), z, sizeof(z));
        const double w = h[6] ? (double)h[6] : 1.0;
        if (h[6])
            fprintf(stderr, "[hnsw_search] cycles/query: descent %.0f pick+adj %.0f visited %.0f gather %.0f accept %.0f insert %.0f\n",
                    h[0] / w, h[1] / w, h[2] / w, h[3] / w, h[4] / w, h[5] / w);
        unsigned long long hm[12] = {0};
        hnsw_mw_read_prof(hm);
        if (hm[6]) {
            const double wq = (double)hm[6];
            fprintf(stderr, "[hnsw_search_mw] control-wave cycles/query: descent %.0f pick %.0f slow-pick %.0f wait %.0f accept %.0f name-next %.0f probe %.0f list+barrierA %.0f merge %.0f  (mispredicted %llu of %llu)\n",
                    hm[0] / wq, hm[1] / wq, hm[2] / wq, hm[3] / wq, hm[8] / wq, hm[9] / wq, hm[10] / wq, hm[11] / wq, hm[5] / wq,
                    hm[7], hm[6]);
        }
        return pe;
    }
    switch (g.space) {
        case SP_L2SQR: return launch_space<SP_L2SQR>(a, p, s);
        case SP_L2: return launch_space<SP_L2>(a, p, s);
        case SP_L1: return launch_space<SP_L1>(a, p, s);
        case SP_LINF: return launch_space<SP_LINF>(a, p, s);
        case SP_NORMCOS: return launch_space<SP_NORMCOS>(a, p, s);
        case SP_COSINE: return launch_space<SP_COSINE>(a, p, s);
        case SP_ANGULAR: return launch_space<SP_ANGULAR>(a, p, s);
        case SP_NEGDOT: return launch_space<SP_NEGDOT>(a, p, s);
        case SP_L2SQR_SIFT: return launch_space<SP_L2SQR_SIFT>(a, p, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_hnsw_search_fix(const HnswDeviceGraph& g, const HnswSearchPlan& p, const void* queries,
                                  uint32_t* bitset, int fix_slots, int32_t* fix_list, int32_t* fix_count,
                                  int32_t* out_ids, float* out_dists, int32_t* out_cnt, int32_t* out_ndc,
                                  int32_t* out_hops, int32_t* out_hops_up, int32_t* status, hipStream_t s) {
    t_fix_list = fix_list;
    t_fix_count = fix_count;
    t_fix_mode = fix_slots;
    const hipError_t e = launch_hnsw_search_ex(g, p, queries, nullptr, nullptr, 0, bitset, out_ids, out_dists, out_cnt,
                                               out_ndc, out_hops, out_hops_up, status, s);
    t_fix_list = nullptr;
    t_fix_count = nullptr;
    t_fix_mode = 0;
    return e;
}

hipError_t launch_hnsw_search(const HnswDeviceGraph& g, const HnswSearchPlan& p, const void* queries,
                              uint32_t* bitset, int32_t* out_ids, float* out_dists, int32_t* out_cnt,
                              int32_t* out_ndc, int32_t* out_hops, int32_t* out_hops_up, int32_t* status,
                              hipStream_t s) {
    return launch_hnsw_search_ex(g, p, queries, nullptr, nullptr, 0, bitset, out_ids, out_dists, out_cnt, out_ndc,
                                 out_hops, out_hops_up, status, s);
}

// ---- SearchOld ---------------------------------------------------------------------------------------------------
HnswSearchPlan hnsw_make_plan_old(const HnswDeviceGraph& g, int nq, int k, int ef, bool force_bitset, int heap_cap) {
    HnswSearchPlan p{};
    p.nq = nq;
    p.k = k;
    p.ef = ef;
    p.cap = ef > k ? ef : k;
    const bool u8 = g.space == SP_L2SQR_SIFT;
    p.a_in_lds = ef <= 8192;
    p.r_in_lds = k <= 2048;
    // accepted items are a fraction of the evaluated ones (~18 per unit of ef on 1M-row graphs)
    long long hc = heap_cap > 0 ? heap_cap : 32ll * p.cap + 4096;
    if (hc > (long long)g.n + 1) hc = (long long)g.n + 1;
    p.heap_cap = (int)hc;
    p.heap_lds = p.heap_cap < 2048 ? p.heap_cap : 2048;
    const int nbcap = hnsw_nbcap(g);
    const size_t fixed = (u8 ? 128 : (size_t)g.ldv * 4) + (size_t)(2 * nbcap + 64) * 4 + (size_t)p.heap_lds * 8 +
                         (p.a_in_lds ? (size_t)((ef + 1) & ~1) * 4 : 0) + (p.r_in_lds ? (size_t)k * 8 : 0);
    int want = 1 << ilog2((g.maxM0 > 0 ? g.maxM0 : 32) * p.cap * 2);
    if (want < 2048) want = 2048;
    const size_t budget = 64 * 1024;
    while ((size_t)want * 4 + fixed > budget && want > 2048) want >>= 1;
    if (force_bitset || 18 * p.cap > want / 2 || (size_t)want * 4 + fixed > budget) {
        p.table_size = 0;
        p.bitset_words = ((size_t)g.n + 31) / 32;
        p.lds_bytes = fixed + 16;
    } else {
        p.table_size = want;
        p.bitset_words = 0;
        p.lds_bytes = fixed + (size_t)want * 4;
    }
    return p;
}

template <int SPACE>
static hipError_t launch_old_space(const HnswArgs& a, const OldWs& w, const HnswSearchPlan& p, hipStream_t s) {
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)p.lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(a.nq), dim3(64), p.lds_bytes, s, a, w);
        return hipGetLastError();
    };
    const bool wide = a.g.maxM0 > 62 || a.g.maxM > 62;
    if (p.table_size == 0) return wide ? go(hnsw_search_old_kernel<SPACE, true, true>) : go(hnsw_search_old_kernel<SPACE, true, false>);
    return wide ? go(hnsw_search_old_kernel<SPACE, false, true>) : go(hnsw_search_old_kernel<SPACE, false, false>);
}

hipError_t launch_hnsw_search_old(const HnswDeviceGraph& g, const HnswSearchPlan& p, const void* queries,
                                  uint32_t* bitset, void* ws_a, void* ws_r, void* ws_heap, int32_t* out_ids,
                                  float* out_dists, int32_t* out_cnt, int32_t* out_ndc, int32_t* out_hops,
                                  int32_t* out_hops_up, int32_t* status, hipStream_t s) {
    if (p.nq == 0) return hipSuccess;
    HnswArgs a{};
    a.g = g;
    a.nbcap = hnsw_nbcap(g);
    a.queries = queries;
    a.level = 0;
    a.bitset = bitset;
    a.bitset_words = p.bitset_words;
    a.out_ids = out_ids;
    a.out_dists = out_dists;
    a.out_cnt = out_cnt;
    a.out_ndc = out_ndc;
    a.out_hops = out_hops;
    a.out_hops_up = out_hops_up;
    a.status = status;
    a.nq = p.nq;
    a.k = p.k;
    a.ef = p.ef;
    a.cap = p.cap;
    a.table_size = p.table_size;
    a.table_shift = p.table_size ? 32 - ilog2(p.table_size) : 0;
    OldWs w{};
    w.capA = p.ef;
    w.capR = p.k;
    w.heap_lds = p.heap_lds;
    w.heap_cap = p.heap_cap;
    w.a_in_lds = p.a_in_lds;
    w.r_in_lds = p.r_in_lds;
    w.a_hbm = static_cast<float*>(ws_a);
    w.r_hbm = static_cast<u64*>(ws_r);
    w.heap_hbm = static_cast<u64*>(ws_heap);
    switch (g.space) {
        case SP_L2SQR: return launch_old_space<SP_L2SQR>(a, w, p, s);
        case SP_L2: return launch_old_space<SP_L2>(a, w, p, s);
        case SP_L1: return launch_old_space<SP_L1>(a, w, p, s);
        case SP_LINF: return launch_old_space<SP_LINF>(a, w, p, s);
        case SP_NORMCOS: return launch_old_space<SP_NORMCOS>(a, w, p, s);
        case SP_COSINE: return launch_old_space<SP_COSINE>(a, w, p, s);
        case SP_ANGULAR: return launch_old_space<SP_ANGULAR>(a, w, p, s);
        case SP_NEGDOT: return launch_old_space<SP_NEGDOT>(a, w, p, s);
        case SP_L2SQR_SIFT: return launch_old_space<SP_L2SQR_SIFT>(a, w, p, s);
        default: return hipErrorInvalidValue;
    }
}


// ---- SearchV1Merge beyond 1024 items (hnsw_search_big_kernel) ---------------------------------------------------
template <int SPACE>
static hipError_t launch_big_space(const HnswArgs& a, float* ws_keys, int32_t* ws_idu, size_t lds, hipStream_t s) {
    if (a.g.maxM0 > 62 || a.g.maxM > 62) hipLaunchKernelGGL((hnsw_search_big_kernel<SPACE, true>), dim3(a.nq), dim3(64), lds, s, a, ws_keys, ws_idu);
    else hipLaunchKernelGGL((hnsw_search_big_kernel<SPACE, false>), dim3(a.nq), dim3(64), lds, s, a, ws_keys, ws_idu);
    return hipGetLastError();
}

hipError_t launch_hnsw_search_big(const HnswDeviceGraph& g, int nq, int k, int ef, const void* queries, uint32_t* bitset,
                                  float* ws_keys, int32_t* ws_idu, int32_t* out_ids, float* out_dists, int32_t* out_cnt,
                                  int32_t* out_ndc, int32_t* out_hops, int32_t* out_hops_up, int32_t* status, hipStream_t s) {
    if (nq == 0) return hipSuccess;
    HnswArgs a{};
    a.g = g;
    a.nbcap = hnsw_nbcap(g);
    a.queries = queries;
    a.bitset = bitset;
    a.bitset_words = ((size_t)g.n + 31) / 32;
    a.out_ids = out_ids;
    a.out_dists = out_dists;
    a.out_cnt = out_cnt;
    a.out_ndc = out_ndc;
    a.out_hops = out_hops;
    a.out_hops_up = out_hops_up;
    a.status = status;
    a.nq = nq;
    a.k = k;
    a.ef = ef;
    a.cap = ef > k ? ef : k;
    const bool u8 = g.space == SP_L2SQR_SIFT;
    const size_t lds = (u8 ? 128 : (size_t)g.ldv * 4) + (size_t)(2 * hnsw_nbcap(g) + 2 * 64) * 4 + 16;
    switch (g.space) {
        case SP_L2SQR: return launch_big_space<SP_L2SQR>(a, ws_keys, ws_idu, lds, s);
        case SP_L2: return launch_big_space<SP_L2>(a, ws_keys, ws_idu, lds, s);
        case SP_L1: return launch_big_space<SP_L1>(a, ws_keys, ws_idu, lds, s);
        case SP_LINF: return launch_big_space<SP_LINF>(a, ws_keys, ws_idu, lds, s);
        case SP_NORMCOS: return launch_big_space<SP_NORMCOS>(a, ws_keys, ws_idu, lds, s);
        case SP_COSINE: return launch_big_space<SP_COSINE>(a, ws_keys, ws_idu, lds, s);
        case SP_ANGULAR: return launch_big_space<SP_ANGULAR>(a, ws_keys, ws_idu, lds, s);
        case SP_NEGDOT: return launch_big_space<SP_NEGDOT>(a, ws_keys, ws_idu, lds, s);
        case SP_L2SQR_SIFT: return launch_big_space<SP_L2SQR_SIFT>(a, ws_keys, ws_idu, lds, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace gfxknn
