// HNSW graph construction on the host (multi-threaded).
//
// Restates Hnsw::CreateIndex / add / kSearchElementsWithAttemptsLevel (src/method/hnsw.cc:
// 183-470,534-708) and HnswNode::getNeighborsByHeuristic2 / addFriendlevel
// (include/method/hnsw.h:129-169,258-314) over flat arrays.  Searching the graph is the GPU's
// job (kernels/hnsw_kernels.hip); construction is the "next" row N2 of SURVEY.md 8f and runs
// here until the batched GPU builder replaces it.
//
// Fidelity: with threads == 1 and the same level stream (mt19937(0), hnsw.h:478-483) this
// produces the reference's adjacency exactly (tests/test_cabi_cpu.py checks it against the
// golden graph), which requires the index-time distances to round like the reference's SSE
// kernels: 4 lanes, product and sum rounded separately (this file is built with
// -ffp-contract=off).
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <queue>
#include <random>
#include <thread>
#include <unordered_set>

#include "engine.hpp"

namespace gfxknn {
namespace {

// ---- index-time distances (Space::IndexTimeDistance) ----------------------------------------
inline float lane4_l2sqr(const float* a, const float* b, size_t n) {  // distcomp_lp.cc:304-365
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    size_t i = 0, n4 = n / 4 * 4;
    for (; i < n4; i += 4) {
        float d0 = a[i] - b[i], d1 = a[i + 1] - b[i + 1], d2 = a[i + 2] - b[i + 2], d3 = a[i + 3] - b[i + 3];
        s0 = s0 + d0 * d0;
        s1 = s1 + d1 * d1;
        s2 = s2 + d2 * d2;
        s3 = s3 + d3 * d3;
    }
    float res = s0 + s1 + s2 + s3;
    for (; i < n; ++i) {
        float d = a[i] - b[i];
        res += d * d;
    }
    return res;
}
inline float lane4_l1(const float* a, const float* b, size_t n) {  // distcomp_lp.cc:190-251
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    size_t i = 0, n4 = n / 4 * 4;
    for (; i < n4; i += 4) {
        s0 = s0 + std::fabs(a[i] - b[i]);
        s1 = s1 + std::fabs(a[i + 1] - b[i + 1]);
        s2 = s2 + std::fabs(a[i + 2] - b[i + 2]);
        s3 = s3 + std::fabs(a[i + 3] - b[i + 3]);
    }
    double res = s0 + s1 + s2 + s3;
    for (; i < n; ++i) res += std::fabs(a[i] - b[i]);
    return (float)res;
}
inline float lane4_linf(const float* a, const float* b, size_t n) {  // distcomp_lp.cc:77-139
    float m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    size_t i = 0, n4 = n / 4 * 4;
    for (; i < n4; i += 4) {
        float d0 = std::fabs(a[i] - b[i]), d1 = std::fabs(a[i + 1] - b[i + 1]);
        float d2 = std::fabs(a[i + 2] - b[i + 2]), d3 = std::fabs(a[i + 3] - b[i + 3]);
        m0 = m0 > d0 ? m0 : d0;
        m1 = m1 > d1 ? m1 : d1;
        m2 = m2 > d2 ? m2 : d2;
        m3 = m3 > d3 ? m3 : d3;
    }
    float a01 = m0 > m1 ? m0 : m1, a23 = m2 > m3 ? m2 : m3;
    float res = a01 > a23 ? a01 : a23;
    for (; i < n; ++i) {
        float d = std::fabs(a[i] - b[i]);
        res = res > d ? res : d;
    }
    return res;
}
inline float lane4_dot(const float* a, const float* b, size_t n) {  // distcomp_scalar.cc:193-245
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    size_t i = 0, n4 = n / 4 * 4;
    for (; i < n4; i += 4) {
        s0 = s0 + a[i] * b[i];
        s1 = s1 + a[i + 1] * b[i + 1];
        s2 = s2 + a[i + 2] * b[i + 2];
        s3 = s3 + a[i + 3] * b[i + 3];
    }
    float res = s0 + s1 + s2 + s3;
    for (; i < n; ++i) res += a[i] * b[i];
    return res;
}
inline float lane4_normdot(const float* a, const float* b, size_t n) {  // distcomp_scalar.cc:83-168
    float p[4] = {0, 0, 0, 0}, x[4] = {0, 0, 0, 0}, y[4] = {0, 0, 0, 0};
    size_t i = 0, n4 = n / 4 * 4;
    for (; i < n4; i += 4)
        for (int j = 0; j < 4; ++j) {
            p[j] = p[j] + a[i + j] * b[i + j];
            x[j] = x[j] + a[i + j] * a[i + j];
            y[j] = y[j] + b[i + j] * b[i + j];
        }
    float sum = p[0] + p[1] + p[2] + p[3], n1 = x[0] + x[1] + x[2] + x[3], n2 = y[0] + y[1] + y[2] + y[3];
    for (; i < n; ++i) {
        sum += a[i] * b[i];
        n1 += a[i] * a[i];
        n2 += b[i] * b[i];
    }
    const float eps = FLT_MIN * 2;
    if (n1 < eps || n2 < eps) return 0;
    float v = sum / std::sqrt(n1) / std::sqrt(n2);
    v = v < 1.0f ? v : 1.0f;
    return v > -1.0f ? v : -1.0f;
}

struct DistFn {
    int space;
    const float* f = nullptr;
    const uint8_t* u = nullptr;
    const int32_t* unorm = nullptr;
    size_t dim = 0;
    float operator()(int i, int j) const {
        if (u) {  // distcomp_l2sqr_sift.cc:41-50
            const uint8_t *a = u + (size_t)i * 128, *b = u + (size_t)j * 128;
            int32_t dot = 0;
            for (int t = 0; t < 128; ++t) dot += (int32_t)a[t] * (int32_t)b[t];
            return (float)(unorm[i] + unorm[j] - 2 * dot);  // < 2^24: exact in float
        }
        const float *a = f + (size_t)i * dim, *b = f + (size_t)j * dim;
        switch (space) {
            case SP_L2: return std::sqrt(lane4_l2sqr(a, b, dim));
            case SP_L1: return lane4_l1(a, b, dim);
            case SP_LINF: return lane4_linf(a, b, dim);
            case SP_COSINE: {
                float v = 1 - lane4_normdot(a, b, dim);
                return v > 0 ? v : 0;
            }
            case SP_ANGULAR: return std::acos(lane4_normdot(a, b, dim));
            case SP_NEGDOT: return -lane4_dot(a, b, dim);
        }
        return 0;
    }
};

struct Closer {  // HnswNodeDistCloser, hnsw.h:433-452 (max-heap on distance)
    float d;
    int id;
    bool operator<(const Closer& o) const { return d < o.d; }
};
struct Farther {  // HnswNodeDistFarther, hnsw.h:389-408 (min-heap on distance)
    float d;
    int id;
    bool operator<(const Farther& o) const { return d > o.d; }
};

class SpinLock {
    std::atomic_flag f_ = ATOMIC_FLAG_INIT;

   public:
    void lock() {
        while (f_.test_and_set(std::memory_order_acquire)) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
    }
    void unlock() { f_.clear(std::memory_order_release); }
};
struct Guard {
    SpinLock& l;
    explicit Guard(SpinLock& x) : l(x) { l.lock(); }
    ~Guard() { l.unlock(); }
};

class Builder {
   public:
    Builder(const DistFn& dist, size_t n, const HnswBuildParams& bp)
        : dist_(dist), n_((int)n), M_(bp.M), maxM_(bp.maxM), maxM0_(bp.maxM0), efC_(bp.efConstruction),
          delaunay_(bp.delaunay), l0_((size_t)n * (bp.maxM0 + 2), 0), up_(n), level_(n, 0), locks_(n) {}

    int32_t* links(int node, int level) {
        return level == 0 ? &l0_[(size_t)node * (maxM0_ + 2)] : up_[node].get() + (size_t)(level - 1) * (maxM_ + 2);
    }
    void init_node(int id, int level) {
        level_[id] = level;
        if (level > 0) {
            up_[id].reset(new int32_t[(size_t)level * (maxM_ + 2)]);
            std::memset(up_[id].get(), 0, (size_t)level * (maxM_ + 2) * sizeof(int32_t));
        }
    }

    // getNeighborsByHeuristic2, hnsw.h:129-169
    void heuristic2(std::priority_queue<Closer>& rs, size_t NN) {
        if (rs.size() < NN) return;
        std::priority_queue<Farther> closest;
        std::vector<Farther> ret;
        while (!rs.empty()) {
            closest.push({rs.top().d, rs.top().id});
            rs.pop();
        }
        while (!closest.empty()) {
            if (ret.size() >= NN) break;
            Farther cur = closest.top();
            closest.pop();
            bool good = true;
            for (const Farther& r : ret) {
                if (dist_(r.id, cur.id) < cur.d) {
                    good = false;
                    break;
                }
            }
            if (good) ret.push_back(cur);
        }
        for (const Farther& r : ret) rs.push({r.d, r.id});
    }

    // getNeighborsByHeuristic1, hnsw.h:82-127: heuristic 2 that tops the list up with the closest rejected items
    void heuristic1(std::priority_queue<Closer>& rs, size_t NN) {
        if (rs.size() < NN) return;
        std::priority_queue<Farther> closest, rejected;
        std::vector<Farther> ret;
        while (!rs.empty()) {
            closest.push({rs.top().d, rs.top().id});
            rs.pop();
        }
        while (!closest.empty()) {
            if (ret.size() >= NN) break;
            const Farther cur = closest.top();
            closest.pop();
            bool good = true;
            for (const Farther& r : ret) {
                if (dist_(r.id, cur.id) < cur.d) {
                    good = false;
                    break;
                }
            }
            if (good) ret.push_back(cur);
            else rejected.push(cur);
        }
        while (ret.size() < NN && !rejected.empty()) {
            ret.push_back(rejected.top());
            rejected.pop();
        }
        for (const Farther& r : ret) rs.push({r.d, r.id});
    }

    // getNeighborsByHeuristic3, hnsw.h:171-256: the candidates and THEIR friends on the level are ranked again; items no
    // kept item is closer to go first ("high priority"), then those only a rejected item is closer to.  The reference gathers
    // the candidates in an unordered_set of node addresses: its iteration order (the allocator's) decides only the order
    // among equal distances; here they enter in order of first appearance.
    void heuristic3(std::priority_queue<Closer>& rs, size_t NN, int self, int level) {
        std::vector<int> cand;
        std::unordered_set<int> seen;
        while (!rs.empty()) {
            const int c = rs.top().id;
            rs.pop();
            if (seen.insert(c).second) cand.push_back(c);
            const int32_t* L = links(c, level);
            for (int i = 0; i < L[0]; ++i)
                if (seen.insert(L[1 + i]).second) cand.push_back(L[1 + i]);
        }
        for (int c : cand)
            if (c != self) rs.push({dist_(c, self), c});
        if (rs.size() < NN) return;
        std::vector<Closer> input(rs.size()), rejected, second, first;
        for (int i = (int)rs.size() - 1; i >= 0; --i) {
            input[(size_t)i] = rs.top();
            rs.pop();
        }
        for (const Closer& cur : input) {
            if (first.size() >= NN) break;
            int good = 2;
            for (const Closer& r : rejected)
                if (dist_(r.id, cur.id) < cur.d) {
                    good = 1;
                    break;
                }
            for (const Closer& r : first)
                if (dist_(r.id, cur.id) < cur.d) {
                    good = 0;
                    break;
                }
            if (good)
                for (const Closer& r : second)
                    if (dist_(r.id, cur.id) < cur.d) {
                        good = 0;
                        break;
                    }
            if (good == 2) first.push_back(cur);
            else if (good == 1) second.push_back(cur);
            else rejected.push_back(cur);
        }
        for (const Closer& r : first) {
            if (rs.size() >= NN) break;
            rs.push(r);
        }
        for (const Closer& r : second) {
            if (rs.size() >= NN) break;
            rs.push(r);
        }
    }

    // the selection of Hnsw::add (hnsw.cc:582-597) and of addFriendlevel's shrink (hnsw.h:283-289) by delaunay_type
    void select(std::priority_queue<Closer>& rs, size_t NN, int self, int level) {
        if (delaunay_ == 1) heuristic1(rs, NN);
        else if (delaunay_ == 3) heuristic3(rs, NN, self, level);
        else heuristic2(rs, NN);
    }

    // addFriendlevel, hnsw.h:258-314
    void add_friend(int node, int level, int elem) {
        Guard g(locks_[node]);
        int32_t* L = links(node, level);
        for (int i = 0; i < L[0]; ++i)
            if (L[1 + i] == elem) return;
        L[1 + L[0]] = elem;
        L[0]++;
        const int maxsz = level > 0 ? maxM_ : maxM0_;
        if (L[0] <= maxsz) return;
        if (delaunay_ > 0) {
            std::priority_queue<Closer> rs;
            for (int i = 0; i < L[0]; ++i) rs.push({dist_(node, L[1 + i]), L[1 + i]});
            select(rs, rs.size() - 1, node, level);
            L[0] = 0;
            while (!rs.empty()) {
                L[1 + L[0]] = rs.top().id;
                L[0]++;
                rs.pop();
            }
        } else {
            float mx = dist_(node, L[1]);
            int maxi = 0;
            for (int i = 1; i < L[0]; ++i) {
                float d = dist_(node, L[1 + i]);
                if (d > mx) {
                    mx = d;
                    maxi = i;
                }
            }
            std::memmove(&L[1 + maxi], &L[2 + maxi], (size_t)(L[0] - 1 - maxi) * sizeof(int32_t));
            L[0]--;
        }
    }

    // kSearchElementsWithAttemptsLevel, hnsw.cc:611-708
    void search_level(int q, size_t ef, std::priority_queue<Closer>& rs, int ep, int level,
                      std::vector<uint32_t>& vis, uint32_t& epoch) {
        if (++epoch == 0) {
            std::fill(vis.begin(), vis.end(), 0u);
            epoch = 1;
        }
        std::priority_queue<Farther> cand;
        float d = dist_(q, ep);
        cand.push({d, ep});
        rs.push({d, ep});
        vis[ep] = epoch;
        std::vector<int32_t> nbv((size_t)std::max(maxM0_, maxM_) + 1);   // (any M: hnsw.cc:189-208)
        int32_t* nb = nbv.data();
        while (!cand.empty()) {
            const Farther cur = cand.top();
            if (cur.d > rs.top().d) break;
            cand.pop();
            int cnt;
            {
                // the reference computes the distances while holding the node's lock; copying
                // the list out first is equivalent for the result and shortens the hold time
                Guard g(locks_[cur.id]);
                const int32_t* L = links(cur.id, level);
                cnt = L[0];
                std::memcpy(nb, L + 1, (size_t)cnt * sizeof(int32_t));
            }
            for (int j = 0; j < cnt; ++j) {
                const int t = nb[j];
                if (vis[t] == epoch) continue;
                vis[t] = epoch;
                d = dist_(q, t);
                if (rs.size() < ef || rs.top().d > d) {
                    rs.push({d, t});
                    cand.push({d, t});
                    if (rs.size() > ef) rs.pop();
                }
            }
        }
    }

    // Hnsw::add, hnsw.cc:534-609
    void add(int id, int curlevel, std::vector<uint32_t>& vis, uint32_t& epoch) {
        std::unique_lock<std::mutex> top_lock;
        if (curlevel > maxlevel_.load()) top_lock = std::unique_lock<std::mutex>(maxlevel_guard_);
        init_node(id, curlevel);
        const int maxlevelcopy = maxlevel_.load();
        int ep = enterpoint_.load();
        if (curlevel < maxlevelcopy) {
            float curdist = dist_(id, ep);
            int cur = ep;
            std::vector<int32_t> nbv((size_t)std::max(maxM0_, maxM_) + 1);
            int32_t* nb = nbv.data();
            for (int level = maxlevelcopy; level > curlevel; --level) {
                bool changed = true;
                while (changed) {
                    changed = false;
                    int cnt;
                    {
                        Guard g(locks_[cur]);
                        const int32_t* L = links(cur, level);
                        cnt = L[0];
                        std::memcpy(nb, L + 1, (size_t)cnt * sizeof(int32_t));
                    }
                    for (int i = 0; i < cnt; ++i) {
                        const float d = dist_(id, nb[i]);
                        if (d < curdist) {
                            curdist = d;
                            cur = nb[i];
                            changed = true;
                        }
                    }
                }
            }
            ep = cur;
        }
        for (int level = std::min(curlevel, maxlevelcopy); level >= 0; --level) {
            std::priority_queue<Closer> rs;
            search_level(id, (size_t)efC_, rs, ep, level, vis, epoch);
            if (delaunay_ == 0) {
                while (rs.size() > (size_t)M_) rs.pop();
            } else {
                select(rs, (size_t)M_, id, level);
            }
            while (!rs.empty()) {
                ep = rs.top().id;
                add_friend(rs.top().id, level, id);  // link(first = neighbour, second = new)
                add_friend(id, level, rs.top().id);
                rs.pop();
            }
        }
        if (curlevel > level_[enterpoint_.load()]) {
            enterpoint_.store(id);
            maxlevel_.store(curlevel);
        }
    }

    // reverse: node 0 first, then n-1, n-2, ..., 1 (the second index of the post-processing, hnsw.cc:262-276)
    void run(const std::vector<int>& levels, int threads, bool reverse = false) {
        if (n_ == 0) return;
        init_node(0, levels[0]);
        maxlevel_.store(levels[0]);
        enterpoint_.store(0);
        std::atomic<int> next(1);
        auto worker = [&]() {
            std::vector<uint32_t> vis((size_t)n_ + 1, 0u);
            uint32_t epoch = 0;
            for (;;) {
                const int pos = next.fetch_add(1);
                if (pos >= n_) break;
                const int id = reverse ? n_ - pos : pos;
                add(id, levels[id], vis, epoch);
            }
        };
        if (threads <= 1) {
            worker();
        } else {
            std::vector<std::thread> th;
            for (int t = 0; t < threads; ++t) th.emplace_back(worker);
            for (auto& t : th) t.join();
        }
    }

    // Post-processing (hnsw.cc:251-330): `this` is the second index (built in reverse order), `first` the original one.
    // Every node but node 0 gets new level-0 friends from the UNION of its friends in both indexes: post = 2 ranks the
    // union again (delaunay_type 0: the maxM0 closest; 1 and 2: heuristic 1; 3: heuristic 3), farthest first in the list;
    // post = 1 keeps the whole union in the iteration order of the reference's unordered_set<size_t> (the same libstdc++
    // container filled in the same order here) and widens maxM0 to the largest union.  The reference runs this loop in
    // parallel with unguarded reads of lists being replaced; here it runs in order, the reference's at indexThreadQty = 1.
    void post_process(Builder& first, int post, std::vector<std::vector<int32_t>>& lists0, int& maxM0_out) {
        size_t maxF = 0;
        lists0.assign((size_t)n_, {});
        for (int id = 1; id < n_; ++id) {
            std::unordered_set<size_t> uni;
            const int32_t* f1 = links(id, 0);
            for (int i = 0; i < f1[0]; ++i) uni.insert((size_t)f1[1 + i]);
            const int32_t* f2 = first.links(id, 0);
            for (int i = 0; i < f2[0]; ++i) uni.insert((size_t)f2[1 + i]);
            if (uni.size() > maxF) maxF = uni.size();
            std::vector<int32_t>& rez = lists0[(size_t)id];
            if (post == 2) {
                std::priority_queue<Closer> rs;
                for (size_t cur : uni) rs.push({dist_((int)cur, id), (int)cur});
                if (delaunay_ == 0) {
                    while (rs.size() > (size_t)maxM0_) rs.pop();
                } else if (delaunay_ == 3) {
                    heuristic3(rs, (size_t)maxM0_, id, 0);
                } else {
                    heuristic1(rs, (size_t)maxM0_);
                }
                while (!rs.empty()) {
                    rez.push_back(rs.top().id);
                    rs.pop();
                }
                int32_t* L = links(id, 0);   // (heuristic 3 of later nodes reads the lists already replaced)
                L[0] = (int32_t)rez.size();
                std::memcpy(L + 1, rez.data(), rez.size() * sizeof(int32_t));
            } else {
                for (size_t cur : uni) rez.push_back((int32_t)cur);
            }
        }
        const int32_t* L0 = links(0, 0);
        lists0[0].assign(L0 + 1, L0 + 1 + L0[0]);
        // post = 1: "maxM0_ = maxF" (hnsw.cc:322); node 0 keeps its list, which the reference's flattened record must hold too
        maxM0_out = post == 1 ? (int)std::max<size_t>(maxF, (size_t)L0[0]) : maxM0_;
    }

    // the flattened graph with level-0 lists given from outside (post-processing)
    void export_graph_lists0(HostGraph& g, const std::vector<std::vector<int32_t>>& lists0, int maxM0) {
        export_graph(g);
        g.maxM0 = maxM0;
        g.links0.assign((size_t)n_ * (maxM0 + 1), 0);
        for (int i = 0; i < n_; ++i) {
            int32_t* L = &g.links0[(size_t)i * (maxM0 + 1)];
            L[0] = (int32_t)lists0[(size_t)i].size();
            std::memcpy(L + 1, lists0[(size_t)i].data(), lists0[(size_t)i].size() * sizeof(int32_t));
        }
    }

    void export_graph(HostGraph& g) {
        g.n = n_;
        g.M = M_;
        g.maxM = maxM_;
        g.maxM0 = maxM0_;
        g.efConstruction = efC_;
        g.delaunay = delaunay_;
        g.maxlevel = maxlevel_.load();
        g.enterpoint = enterpoint_.load();
        g.levels.assign(level_.begin(), level_.end());
        g.links0.assign((size_t)n_ * (maxM0_ + 1), 0);
        g.up_off.assign(n_, -1);
        g.up_links.clear();
        for (int i = 0; i < n_; ++i) {
            const int32_t* L = links(i, 0);
            std::memcpy(&g.links0[(size_t)i * (maxM0_ + 1)], L, (size_t)(L[0] + 1) * sizeof(int32_t));
            if (level_[i] > 0) {
                g.up_off[i] = (int64_t)g.up_links.size();
                for (int l = 1; l <= level_[i]; ++l) {
                    const int32_t* U = links(i, l);
                    const size_t at = g.up_links.size();
                    g.up_links.resize(at + maxM_ + 1, 0);
                    std::memcpy(&g.up_links[at], U, (size_t)(U[0] + 1) * sizeof(int32_t));
                }
            }
        }
    }

   private:
    DistFn dist_;
    int n_, M_, maxM_, maxM0_, efC_, delaunay_;
    std::vector<int32_t> l0_;
    std::vector<std::unique_ptr<int32_t[]>> up_;
    std::vector<int> level_;
    std::vector<SpinLock> locks_;
    std::atomic<int> maxlevel_{0}, enterpoint_{0};
    std::mutex maxlevel_guard_;
};

}  // namespace

void hnsw_check_params(const HnswBuildParams& bp) {
    // any M, as the reference (hnsw.cc:189-208).  Up to M / maxM 62 and maxM0 126 one wavefront holds a list in one or two
    // words per lane (the fast search kernels, the GPU builder); longer lists are built on the host and searched by the
    // kernels that walk lists in chunks (hnsw_kernels.hip, collect_unvisited_any).  The bound below only keeps sizes sane.
    if (bp.M < 1 || bp.M > 4096 || bp.maxM < 1 || bp.maxM > 4096 || bp.maxM0 < 1 || bp.maxM0 > 8192)
        throw EngineError(Err::IndexBuildFailed, "HNSW: M / maxM must be in [1, 4096] and maxM0 in [1, 8192]");
    if (bp.delaunay < 0 || bp.delaunay > 3) throw EngineError(Err::IndexBuildFailed, "HNSW: delaunay_type must be 0, 1, 2 or 3");
    if (bp.post < 0 || bp.post > 2) throw EngineError(Err::IndexBuildFailed, "HNSW: post must be 0, 1 or 2");
}

// `draws` levels of the one stream (the post-processing's second index keeps drawing from it)
static std::vector<int32_t> random_level_draws(size_t draws, const HnswBuildParams& bp) {
    const double mult = bp.mult > 0 ? bp.mult : 1.0 / std::log(1.0 * bp.M);
    std::mt19937 gen(0);
    std::uniform_real_distribution<float> uni(0, 1);
    std::vector<int32_t> levels(draws);
    for (size_t i = 0; i < draws; ++i) {
        float r = -std::log(uni(gen)) * mult;
        levels[i] = (int32_t)r;
    }
    return levels;
}

std::vector<int32_t> hnsw_random_levels(size_t n, const HnswBuildParams& bp) {
    // getRandomLevel (hnsw.h:478-483): one mt19937 stream seeded with the library seed 0
    // (init.cc:34-38), consumed in insertion order
    const double mult = bp.mult > 0 ? bp.mult : 1.0 / std::log(1.0 * bp.M);
    std::mt19937 gen(0);
    std::uniform_real_distribution<float> uni(0, 1);
    std::vector<int32_t> levels(n);
    for (size_t i = 0; i < n; ++i) {
        float r = -std::log(uni(gen)) * mult;
        levels[i] = (int32_t)r;
    }
    return levels;
}

void hnsw_build_host(int space, const void* rows, size_t n, size_t dim, const HnswBuildParams& bp,
                     HostGraph& out) {
    hnsw_check_params(bp);
    DistFn dist;
    dist.space = space;
    dist.dim = dim;
    std::vector<int32_t> unorm;
    if (space == SP_L2SQR_SIFT) {
        dist.u = static_cast<const uint8_t*>(rows);
        unorm.resize(n);
        for (size_t i = 0; i < n; ++i) {
            int32_t s = 0;
            for (int t = 0; t < 128; ++t) s += (int32_t)dist.u[i * 128 + t] * (int32_t)dist.u[i * 128 + t];
            unorm[i] = s;
        }
        dist.unorm = unorm.data();
    } else {
        dist.f = static_cast<const float*>(rows);
    }
    const std::vector<int32_t> lv = hnsw_random_levels(n, bp);
    std::vector<int> levels(lv.begin(), lv.end());
    int threads = bp.threads > 0 ? bp.threads : (int)std::thread::hardware_concurrency();
    if (threads < 1) threads = 1;
    Builder b(dist, n, bp);
    b.run(levels, threads);
    if (bp.post == 0 || n == 0) {
        b.export_graph(out);
        return;
    }
    // post = 1, 2 (hnsw.cc:251-330): the same index once more in reverse order -- node 0 with the stream's next level,
    // then n-1, ..., 1 -- and new level-0 lists from both
    const std::vector<int32_t> draws = random_level_draws(2 * n, bp);
    std::vector<int> levels2(n);
    levels2[0] = draws[n];
    for (size_t pos = 1; pos < n; ++pos) levels2[n - pos] = draws[n + pos];
    Builder b2(dist, n, bp);
    b2.run(levels2, threads, /*reverse=*/true);
    std::vector<std::vector<int32_t>> lists0;
    int maxM0 = bp.maxM0;
    b2.post_process(b, bp.post, lists0, maxM0);
    b2.export_graph_lists0(out, lists0, maxM0);
}

}  // namespace gfxknn
