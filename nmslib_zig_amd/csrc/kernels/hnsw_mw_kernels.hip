// HNSW search for small batches: one workgroup per query (see below).  Split from hnsw_kernels.hip.
#include <cstdio>
#include <cstdlib>

#include "hnsw_args.hpp"
#include "hnsw_common_dev.hpp"
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
// [8] accept (LDS read + ballots), [9] naming the next node, [10] visited probe loop, [11] list + barrier A
__device__ unsigned long long g_hnsw_mw_prof[12];

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
        // the row of the first round is read together with the count (the list is always padded to 64 valid ids)
        int id = nbr[w * 8 + g8];
        const int m = __builtin_amdgcn_readfirstlane(ctl[0]);
        if (m < 0) break;
        for (int base = w * 8; base < m; base += 8 * MW_NW) {
            const int idx = base + g8;
            if (base != w * 8) id = nbr[idx];
            const float* rp = rows + (size_t)id * g.ldv;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
            auto fetch = [&](f32x4 (&bb)[4], int cb) __attribute__((always_inline)) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int d = cb + sub * 4 + 32 * it;
                    bb[it] = *reinterpret_cast<const f32x4*>(rp + (d < dlast ? d : dlast));
                }
            };
            // a whole step of 128 floats: nothing to mask
            auto consume_full = [&](const f32x4 (&bb)[4], int cb) __attribute__((always_inline)) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const f32x4 qq = *reinterpret_cast<const f32x4*>(qv + cb + sub * 4 + 32 * it);
                    accum4<SPACE>(qq, bb[it], s0, s1, s2);
                }
            };
            // the last, partial step: elements past the row contribute zeros on both sides (same sums, bit for bit)
            auto consume_tail = [&](const f32x4 (&bb)[4], int cb) __attribute__((always_inline)) {
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
                    consume_full(cur, cb);
#pragma unroll
                    for (int it = 0; it < 4; ++it) cur[it] = nxt[it];
                    cb += 128;
                } else {
                    if (cb + 128 == g.ldv) consume_full(cur, cb);
                    else consume_tail(cur, cb);
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

// wave_shr:1 -- lane i takes lane i-1 of the whole wave (lane 0 keeps its own value)
template <typename T>
__device__ __forceinline__ T dpp_wave_shr1(T v) {
    const int x = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(x, x, 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float readlane_f(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
// lane mask of a per-lane condition, straight from the compare (no 0/1 round trip through a VGPR)
#define BALLOT(x) ((u64)__builtin_amdgcn_ballot_w64(x))
__device__ __forceinline__ int mbcnt64(u64 m) {  // set bits of m below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// The control wave.  The sorted array (SortArrBI, include/sort_arr_bi.h) lives in REGISTERS: item i = lane i%64 of
// register i/64 (key; id | used<<31; free slots: +inf, -1).  A lone wave pays ~10 clocks per instruction in code like
// this (scalar/vector ping-pong, short branches), so the loop is written for instruction count: wave-uniform control
// flow only (no per-lane branches: the compiler turns those into exec-mask state machines), the first unused item is
// the lowest set bit of ballot(id >= 0), picking it is a readlane, one accepted item is inserted with two ballots and
// one wave_shr:1 per register (SortArrBI::push_or_replace_non_empty_exp, :159-199), several at once by counting
// (hnsw_search_body's merge) through an LDS scatter.
template <int SPACE, int E, bool PROF>
__device__ __forceinline__ void mw_control(const HnswArgs& a, const int q, float* keys, int* idu, int* nbr, float* nd,
                                           int* ctl, uint32_t* table, const int lane) {
    const HnswDeviceGraph& g = a.g;
    int ndc = 0, hops = 0, hops_up = 0, nvisited = 0;
    bool inflight = false;
    const uint32_t tmask = (uint32_t)a.table_size - 1u;
    const int tfull = a.table_size - (a.table_size >> 3);

    long long pc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long pt = PROF ? (long long)__builtin_readcyclecounter() : 0;
    auto lap = [&](int ph) __attribute__((always_inline)) {
        if constexpr (PROF) {
            const long long now = (long long)__builtin_readcyclecounter();
            pc[ph] += now - pt;
            pt = now;
        }
    };

    // visited filter of an adjacency list v (lane L: neighbour L; lane 63: the count): test-and-set in the LDS hash
    // (exact), the unvisited neighbours go to nbr[0..m), the rest of nbr[] is padded with the last of them, ctl[0] = m.
    // Wave-uniform probe loop: lanes that are done keep issuing a compare-and-swap that cannot match.
    auto filter = [&](const int v) __attribute__((always_inline)) -> int {
        const int cntn = __builtin_amdgcn_readlane(v, 63);
        // (the loop state is kept as lane masks in scalar registers: per-lane booleans carried around a loop are
        //  rebuilt by the compiler through a 0/1 VGPR every iteration)
        u64 pm = BALLOT(lane < cntn), nm = 0;
        uint32_t h = ((uint32_t)v * 2654435761u) >> a.table_shift;
        while (pm) {
            const bool pend = __builtin_amdgcn_inverse_ballot_w64(pm);
            const uint32_t old = atomicCAS(&table[h], pend ? HT_EMPTY : 0xFFFFFFFEu, (uint32_t)v);
            const u64 got = BALLOT(old == HT_EMPTY) & pm;
            nm |= got;
            pm &= ~got & BALLOT(old != (uint32_t)v);
            h = (h + 1) & tmask;
        }
        const bool isn = __builtin_amdgcn_inverse_ballot_w64(nm);
        lap(10);
        const int m = __popcll(nm);
        if (m > 0) {
            const int lastid = __builtin_amdgcn_readlane(v, 63 - __clzll((long long)nm));
            const int pos = mbcnt64(nm);
            nbr[isn ? pos : m + lane - pos] = isn ? v : lastid;  // every lane writes exactly one of the 64 slots
            ctl[0] = m;
        }
        nvisited += m;
        return m;
    };
    // adjacency list of node c on the level being searched, shifted by one word: lane L holds neighbour L, the last
    // lanes hold the count (word 0).  (level > 0: the search step of the graph construction, hnsw_build_kernels.hip)
    auto load_adj0 = [&](int c) __attribute__((always_inline)) -> int {
        if (a.level == 0) {
            const int w = lane + 1 <= g.maxM0 ? lane + 1 : 0;
            return g.links0[(size_t)c * (g.maxM0 + 1) + w];
        }
        const int w = lane + 1 <= g.maxM ? lane + 1 : 0;
        return g.up_links[g.up_off[c] + (int64_t)(a.level - 1) * (g.maxM + 1) + w];
    };
    // ---- entry point + greedy descent (hnsw_distfunc_opt.cc:168-198) ----
    const int start = a.start_nodes ? a.start_nodes[q] : -1;   // construction mode: a given start node, or descend to level + 1
    int cur = start >= 0 ? start : g.enterpoint;
    nbr[lane] = cur;
    ctl[0] = 1;
    mw_barrier();  // A
    mw_barrier();  // B
    float curdist = nd[0];
    ndc += 1;
    for (int lvl = (start >= 0 ? 0 : g.maxlevel); lvl > a.level; --lvl) {
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
                ctl[0] = cntl;
                mw_barrier();  // A
                mw_barrier();  // B
                ndc += cntl;
                // sequential "if (d < curdist)" scan == first index attaining the minimum
                const float dl = lane < cntl ? nd[lane] : INFINITY;
                const float dmin = wave_min_f32(dl);
                if (dmin < curdist) {
                    const u64 mm = BALLOT((lane < cntl) & (dl == dmin));
                    curdist = dmin;
                    cur = __builtin_amdgcn_readlane(v, __ffsll((long long)mm) - 1);
                    changed = true;
                }
            }
        }
    }

    // ---- level 0 (hnsw_distfunc_opt.cc:200-274) ----
    float kreg[E];
    int ireg[E], posv[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        posv[e] = lane + 64 * e;
        kreg[e] = (e == 0 && lane == 0) ? curdist : INFINITY;
        ireg[e] = (e == 0 && lane == 0) ? cur : -1;
    }
    int n = 1;
    {   // the start node is visited (one lane inserts it; nothing else is in the table yet)
        const uint32_t h = ((uint32_t)cur * 2654435761u) >> a.table_shift;
        table[h] = (uint32_t)cur;
    }
    nvisited = 1;

    auto key_at = [&](int i) __attribute__((always_inline)) -> float {  // i uniform
        float r = readlane_f(kreg[0], i & 63);
#pragma unroll
        for (int e = 1; e < E; ++e) {
            const float x = readlane_f(kreg[e], i & 63);
            r = (i >> 6) == e ? x : r;
        }
        return r;
    };
    auto id_at = [&](int i) __attribute__((always_inline)) -> int {
        int r = __builtin_amdgcn_readlane(ireg[0], i & 63);
#pragma unroll
        for (int e = 1; e < E; ++e) {
            const int x = __builtin_amdgcn_readlane(ireg[e], i & 63);
            r = (i >> 6) == e ? x : r;
        }
        return r;
    };
    // first unused item: lowest set bit of ballot(id >= 0) (64*E when there is none)
    auto first_unused = [&]() __attribute__((always_inline)) -> int {
        int fu = 64 * E;
#pragma unroll
        for (int e = E - 1; e >= 0; --e) {
            const u64 U = BALLOT(ireg[e] >= 0);
            fu = U ? 64 * e + (int)__builtin_ctzll(U) : fu;
        }
        return fu;
    };
    // SortArrBI::push_or_replace_non_empty_exp (sort_arr_bi.h:159-199) for one item
    auto insert = [&](const float key, const int id) __attribute__((always_inline)) {
        const float lastk = key_at(n - 1);
        int p, newn;
        if (lastk <= key) {
            if (n >= a.cap) return;
            p = n;
            newn = n + 1;
        } else {
            // insertion index.  Without a key equal to the new one in the array, the reference's exponential probe +
            // lower_bound (:172-186) is simply the number of smaller keys; with equal keys present (rare) the probe
            // is replayed so that the item lands inside the run exactly where the reference puts it.
            int less = 0, leq = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                less += __popcll(BALLOT(kreg[e] < key));
                leq += __popcll(BALLOT(kreg[e] <= key));
            }
            p = less;
            if (leq != less) {
                int curr = n - 1, prev = curr, dstep = 1;
                while (curr > 0 && key_at(curr) > key) {
                    prev = curr;
                    curr -= dstep;
                    dstep *= 2;
                    if (dstep > curr) dstep = curr;
                }
                p = curr;
                for (int i = curr; i < prev && key_at(i) < key; ++i) p = i + 1;
            }
            newn = n < a.cap ? n + 1 : a.cap;
        }
        float shk[E];
        int shi[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            shk[e] = dpp_wave_shr1(kreg[e]);
            shi[e] = dpp_wave_shr1(ireg[e]);
        }
#pragma unroll
        for (int e = 1; e < E; ++e) {  // lane 0 of a register continues lane 63 of the one before
            const float ck = readlane_f(kreg[e - 1], 63);
            const int ci = __builtin_amdgcn_readlane(ireg[e - 1], 63);
            shk[e] = lane == 0 ? ck : shk[e];
            shi[e] = lane == 0 ? ci : shi[e];
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool mv = posv[e] > p && posv[e] < newn;
            const bool at = posv[e] == p;
            kreg[e] = at ? key : (mv ? shk[e] : kreg[e]);
            ireg[e] = at ? id : (mv ? shi[e] : ireg[e]);
        }
        n = newn;
    };
    // the accepted items of one expansion (lanes with acc: dj, idj) into the array, in ascending order of
    // (distance, list position) exactly as hnsw_distfunc_opt.cc:251-266 inserts them one after the other
    auto merge = [&](const float dj, const int idj, const bool acc, const u64 amask, const int m2) __attribute__((always_inline)) {
        if (m2 == 1) {
            const int j = (int)__builtin_ctzll(amask);
            insert(readlane_f(dj, j), __builtin_amdgcn_readlane(idj, j));
            return;
        }
        // several: rank of every accepted item among them, number of accepted keys below every old item, number of
        // old keys below every accepted item -- then every item knows its place in the merged array (cut at the
        // capacity).  Two equal keys anywhere (rare): one insertion at a time instead, in rank order.
        int rank = 0, less = 0;
        int cnt[E];
        u64 tie = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) cnt[e] = 0;
        for (u64 mm = amask; mm; mm &= mm - 1) {
            const int j = (int)__builtin_ctzll(mm);
            const float kt = readlane_f(dj, j);
            rank += ((kt < dj) | ((kt == dj) & (j < lane))) ? 1 : 0;
            int lt = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                lt += __popcll(BALLOT(kreg[e] < kt));
                cnt[e] += kt < kreg[e] ? 1 : 0;
                tie |= BALLOT(kt == kreg[e]);
            }
            tie |= BALLOT(acc & (kt == dj) & (j != lane));
            less = lane == j ? lt : less;
        }
        if (tie == 0) {
            const int newn = n + m2 < a.cap ? n + m2 : a.cap;
            const int first = __builtin_amdgcn_readlane(less, (int)__builtin_ctzll(BALLOT(acc & (rank == 0))));
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int np = posv[e] + cnt[e];
                if (cnt[e] > 0 && np < newn) {  // (free slots have key +inf: cnt = m2, np >= n + m2 >= newn)
                    keys[np] = kreg[e];
                    idu[np] = ireg[e];
                }
            }
            {
                const int np = less + rank;
                if (acc && np < newn) {
                    keys[np] = dj;
                    idu[np] = idj;
                }
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const bool ch = posv[e] >= first && posv[e] < newn;
                const float k2 = keys[ch ? posv[e] : 0];
                const int i2 = idu[ch ? posv[e] : 0];
                kreg[e] = ch ? k2 : kreg[e];
                ireg[e] = ch ? i2 : ireg[e];
            }
            n = newn;
        } else {
            for (int t = 0; t < m2; ++t) {
                const int j = (int)__builtin_ctzll(BALLOT(acc & (rank == t)));
                insert(readlane_f(dj, j), __builtin_amdgcn_readlane(idj, j));
            }
        }
    };

    // pre_*: the array's first unused item behind the node being expanded, with its adjacency requested early
    int pre_node = -1, pre_v = 0;
    float pre_key = INFINITY;
    bool pre_ok = false;
    // P: the node of the NEXT expansion, named before the merge (-1: not named; the sequential pick decides)
    int P = -1, Pv = 0;
    // accepted items of the expansion whose merge is pending
    float dj = INFINITY;
    int idj = -1, m2 = 0;
    bool acc = false;
    u64 amask = 0;
    int status = 0;  // 1: the visited table is nearly full, 2: the early choice was not the sequential one (cannot happen)
    lap(0);
    while (true) {
        if (P < 0) {
            // nothing named: finish the merge, then the sequential pick (:204-215)
            if (m2 > 0) merge(dj, idj, acc, amask, m2);
            m2 = 0;
            const int fu0 = first_unused();
            if (fu0 >= (n < a.ef ? n : a.ef)) break;
            P = id_at(fu0);
            Pv = (P == pre_node) ? pre_v : load_adj0(P);
            lap(2);
        }
        // ---- visited filter of P; its rows go to the gather waves ----
        const int m = filter(Pv);
        if (nvisited > tfull) {
            status = 1;
            break;
        }
        if (m > 0) {
            mw_barrier();  // A
            inflight = true;
        }
        lap(11);
        // ---- while they work: the merge of the previous expansion, then the pick of this one ----
        if (m2 > 0) merge(dj, idj, acc, amask, m2);
        m2 = 0;
        lap(5);
        const int lim = n < a.ef ? n : a.ef;
        const int fu = first_unused();
        const int c = fu < lim ? id_at(fu) : -2;
        if (c != P) {
            status = 2;
            break;
        }
#pragma unroll
        for (int e = 0; e < E; ++e) ireg[e] = posv[e] == fu ? (c | (int)0x80000000) : ireg[e];
        hops++;
        const float topKey = key_at(n - 1);
        const bool filling = n < a.ef;
        {   // the first unused item behind it, and its adjacency one expansion early (a random HBM read: ~1 us)
            const int fu2 = first_unused();
            pre_ok = fu2 < lim;
            const int node = pre_ok ? id_at(fu2) : -1;
            pre_key = pre_ok ? key_at(fu2) : INFINITY;
            if (pre_ok && node != pre_node) pre_v = load_adj0(node);
            pre_node = node;
        }
        P = -1;
        lap(1);
        if (m == 0) {
            if (pre_ok) {
                P = pre_node;
                Pv = pre_v;
            }
            continue;
        }
        ndc += m;
        mw_barrier();  // B: distances of this expansion
        inflight = false;
        lap(3);
        // accept d < topKey || size < ef   (:240)
        dj = nd[lane];
        idj = nbr[lane];
        acc = (lane < m) & ((dj < topKey) | filling);
        amask = BALLOT(acc);
        m2 = __popcll(amask);
        lap(8);
        // The next expansion, named before the merge.  pre_node is the first unused item of the array; the merge
        // moves it up by the number of accepted keys below it and puts nothing unused in front of it except those
        // accepted items.  So: no accepted key below pre_key -> pre_node is next (its position, unchanged, is
        // below ef); otherwise the closest accepted item is next (it lands in front of pre_node, hence below ef).
        // Equal keys leave the order to the merge: no early choice then.
        if (pre_ok) {
            const u64 below = BALLOT(acc & (dj < pre_key));
            const u64 equal = BALLOT(acc & (dj == pre_key));
            if (!equal) {
                if (!below) {
                    P = pre_node;
                    Pv = pre_v;
                } else {
                    const float best = wave_min_f32(acc ? dj : INFINITY);
                    const u64 bm = BALLOT(acc & (dj == best));
                    if (__popcll(bm) == 1) {
                        P = __builtin_amdgcn_readlane(idj, (int)__builtin_ctzll(bm));
                        Pv = load_adj0(P);
                    }
                }
            }
        }
        lap(9);
    }
    // ---- release the gather waves (every published list is collected first) ----
    if (inflight) mw_barrier();  // B
    ctl[0] = -1;
    mw_barrier();  // A
    if constexpr (PROF) {
        if (lane == 0) {
            for (int i = 0; i < 12; ++i)
                if (i != 6 && i != 7) atomicAdd(&g_hnsw_mw_prof[i], (unsigned long long)pc[i]);
            atomicAdd(&g_hnsw_mw_prof[6], 1ull);
        }
    }
    if (status == 2 && lane == 0) atomicAdd(&g_hnsw_mw_prof[7], 1ull);

    // ---- results: first k items, ties ordered by internal id (as hnsw_search_body) ----
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (posv[e] < a.capa) {
            keys[posv[e]] = kreg[e];
            idu[posv[e]] = ireg[e];
        }
    }
    __builtin_amdgcn_wave_barrier();
    const bool redo = status != 0;
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

template <int SPACE, int SA_EMAX, bool PROF>
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
    int* ctl = reinterpret_cast<int*>(nd + 64);              // [4]  rows in nbr[] (-1: the search is over)
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
        const float* src = a.query_rows ? reinterpret_cast<const float*>(g.rows) + (size_t)a.query_rows[q] * g.ldv
                                        : reinterpret_cast<const float*>(a.queries) + (size_t)q * g.dim;
        if (wave == 0) {
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
        for (int i = threadIdx.x; i < a.table_size; i += MW_THREADS) table[i] = HT_EMPTY;
        if (threadIdx.x == 0) ctl[0] = 0;
    }
    __syncthreads();
    if (wave == 0) mw_control<SPACE, SA_EMAX, PROF>(a, q, keys, idu, nbr, nd, ctl, table, lane);
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
    if constexpr (SPACE == SP_L2SQR) {
        if (a.prof && sa_emax <= 2) return go(hnsw_search_mw_kernel<SPACE, 2, true>);
    }
    if (sa_emax <= 2) return go(hnsw_search_mw_kernel<SPACE, 2, false>);
    return go(hnsw_search_mw_kernel<SPACE, 4, false>);
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

void hnsw_mw_read_prof(unsigned long long out[12]) {
    unsigned long long z[12] = {0};
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hnsw_mw_prof), sizeof(z));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_hnsw_mw_prof), z, sizeof(z));
}

}  // namespace gfxknn
