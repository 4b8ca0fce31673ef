// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A small command-line driver (this repo's own code) over the *reference's*
// C++ API.  It is linked against object files compiled from /root/reference by
// oracle/Makefile and lands in oracle/_ref/ref_driver.  It exists because the
// reference's C shim cannot do three things the oracle needs:
//   * set efSearch (nmslib_c.cpp:330,986 forces efSearch=200 on every query);
//   * build HNSW deterministically (indexThreadQty=1) and save the optimized
//     index so the GPU search can be checked on the *same graph*;
//   * time queries with T harness threads the way upstream's
//     Experiments::Execute does (include/experiments.h:175-213,265-266):
//     query q is served by thread q mod T, QPS = queries / wall-clock.
//
// Usage:
//   ref_driver --space l2 --method hnsw --data base.bin --n N --dim D
//              --queries q.bin --nq Q --k K --out PREFIX
//              [--u8] [--index-params "M=16,efConstruction=200"]
//              [--query-params "efSearch=128"] [--threads T] [--repeat R]
//              [--save PATH] [--load PATH] [--ids ids.i32]
// Files are raw little-endian row-major arrays (f32, or u8 with --u8).
// Outputs PREFIX.ids.i32 / PREFIX.dists.f32 (Q x K, padded with -1 / +inf),
// PREFIX.cnt.i32 (results per query), PREFIX.ndc.i64 (distance computations per
// query as counted by Query::DistanceComputations -- non-zero on the generic
// search path only), and one JSON line on stdout.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "init.h"
#include "index.h"
#include "knnquery.h"
#include "knnqueue.h"
#include "methodfactory.h"
#include "object.h"
#include "params.h"
#include "space.h"
#include "space/space_l2sqr_sift.h"
#include "space/space_vector.h"
#include "spacefactory.h"

using namespace similarity;

static std::vector<std::string> split_csv(const std::string& s) {
    std::vector<std::string> out;
    std::stringstream ss(s);
    std::string tok;
    while (std::getline(ss, tok, ','))
        if (!tok.empty()) out.push_back(tok);
    return out;
}

template <typename T>
static std::vector<T> read_raw(const std::string& path, size_t count) {
    std::vector<T> v(count);
    std::ifstream f(path, std::ios::binary);
    if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
    f.read(reinterpret_cast<char*>(v.data()), count * sizeof(T));
    if (size_t(f.gcount()) != count * sizeof(T)) {
        fprintf(stderr, "short read on %s\n", path.c_str());
        exit(2);
    }
    return v;
}

template <typename T>
static void write_raw(const std::string& path, const std::vector<T>& v) {
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(v.data()), v.size() * sizeof(T));
}

struct Args {
    std::string space = "l2", method = "hnsw", data, queries, out, save, load, ids;
    std::string index_params, query_params;
    size_t n = 0, dim = 0, nq = 0, k = 10, threads = 1, repeat = 1;
    bool u8 = false;
};

template <typename dist_t>
static Object* make_obj(const Space<dist_t>* space, bool u8, const void* p, size_t dim, int id) {
    if (u8) {
        auto sift = dynamic_cast<const SpaceL2SqrSift*>(
            reinterpret_cast<const Space<int>*>(space));
        const uint8_t* b = static_cast<const uint8_t*>(p);
        std::vector<uint8_t> v(b, b + dim);
        return sift->CreateObjFromUint8Vect(id, -1, v);
    }
    auto vs = dynamic_cast<const VectorSpaceSimpleStorage<float>*>(
        reinterpret_cast<const Space<float>*>(space));
    const float* f = static_cast<const float*>(p);
    std::vector<float> v(f, f + dim);
    return vs->CreateObjFromVect(id, -1, v);
}

template <typename dist_t>
static int run(const Args& a) {
    using clk = std::chrono::steady_clock;
    std::unique_ptr<Space<dist_t>> space(
        SpaceFactoryRegistry<dist_t>::Instance().CreateSpace(a.space, AnyParams()));
    const size_t esz = a.u8 ? 1 : 4;
    std::vector<char> base = read_raw<char>(a.data, a.n * a.dim * esz);
    std::vector<char> qs = read_raw<char>(a.queries, a.nq * a.dim * esz);
    std::vector<int32_t> ids;
    if (!a.ids.empty()) ids = read_raw<int32_t>(a.ids, a.n);

    ObjectVector data;
    data.reserve(a.n);
    for (size_t i = 0; i < a.n; ++i)
        data.push_back(make_obj(space.get(), a.u8, base.data() + i * a.dim * esz, a.dim,
                                ids.empty() ? int(i) : ids[i]));

    std::unique_ptr<Index<dist_t>> index(MethodFactoryRegistry<dist_t>::Instance().CreateMethod(
        false, a.method, a.space, *space, data));
    double build_s = 0;
    if (!a.load.empty()) {
        index->LoadIndex(a.load);
    } else {
        auto t0 = clk::now();
        index->CreateIndex(AnyParams(split_csv(a.index_params)));
        build_s = std::chrono::duration<double>(clk::now() - t0).count();
    }
    if (!a.save.empty()) index->SaveIndex(a.save);
    index->SetQueryTimeParams(AnyParams(split_csv(a.query_params)));

    std::vector<int32_t> out_ids(a.nq * a.k, -1), out_cnt(a.nq, 0);
    std::vector<float> out_d(a.nq * a.k, std::numeric_limits<float>::infinity());
    std::vector<int64_t> out_ndc(a.nq, 0);

    std::vector<Object*> qobjs(a.nq);
    for (size_t q = 0; q < a.nq; ++q)
        qobjs[q] = make_obj(space.get(), a.u8, qs.data() + q * a.dim * esz, a.dim, 0);

    const size_t T = std::max<size_t>(1, a.threads);
    double best_s = 1e30;
    for (size_t rep = 0; rep < std::max<size_t>(1, a.repeat); ++rep) {
        auto worker = [&](size_t tid) {
            for (size_t q = tid; q < a.nq; q += T) {
                // Hnsw cosine normalises the query payload in place
                // (hnsw_distfunc_opt.cc:160-162): give every run a fresh copy.
                std::unique_ptr<Object> qcopy(qobjs[q]->Clone());
                KNNQuery<dist_t> knn(*space, qcopy.get(), unsigned(a.k));
                index->Search(&knn);
                std::unique_ptr<KNNQueue<dist_t>> res(knn.Result()->Clone());
                size_t found = res->Size();
                out_cnt[q] = int32_t(found);
                out_ndc[q] = int64_t(knn.DistanceComputations());
                for (size_t j = found; j-- > 0;) {  // heap pops worst first
                    out_d[q * a.k + j] = float(res->TopDistance());
                    out_ids[q * a.k + j] = res->TopObject()->id();
                    res->Pop();
                }
            }
        };
        auto t0 = clk::now();
        if (T == 1) {
            worker(0);
        } else {
            std::vector<std::thread> th;
            for (size_t t = 0; t < T; ++t) th.emplace_back(worker, t);
            for (auto& t : th) t.join();
        }
        best_s = std::min(best_s, std::chrono::duration<double>(clk::now() - t0).count());
    }

    if (!a.out.empty()) {
        write_raw(a.out + ".ids.i32", out_ids);
        write_raw(a.out + ".dists.f32", out_d);
        write_raw(a.out + ".cnt.i32", out_cnt);
        write_raw(a.out + ".ndc.i64", out_ndc);
    }
    double ndc_mean = 0;
    for (auto v : out_ndc) ndc_mean += double(v);
    ndc_mean /= double(std::max<size_t>(1, a.nq));
    printf("{\"space\":\"%s\",\"method\":\"%s\",\"n\":%zu,\"dim\":%zu,\"nq\":%zu,\"k\":%zu,"
           "\"threads\":%zu,\"build_s\":%.4f,\"query_s\":%.6f,\"qps\":%.2f,\"ndc_mean\":%.1f}\n",
           a.space.c_str(), a.method.c_str(), a.n, a.dim, a.nq, a.k, T, build_s, best_s,
           double(a.nq) / best_s, ndc_mean);
    // objects are intentionally leaked at exit (the index holds raw pointers).
    return 0;
}

int main(int argc, char** argv) {
    Args a;
    for (int i = 1; i < argc; ++i) {
        std::string f = argv[i];
        auto next = [&]() -> std::string {
            if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", f.c_str()); exit(2); }
            return argv[++i];
        };
        if (f == "--space") a.space = next();
        else if (f == "--method") a.method = next();
        else if (f == "--data") a.data = next();
        else if (f == "--queries") a.queries = next();
        else if (f == "--out") a.out = next();
        else if (f == "--save") a.save = next();
        else if (f == "--load") a.load = next();
        else if (f == "--ids") a.ids = next();
        else if (f == "--index-params") a.index_params = next();
        else if (f == "--query-params") a.query_params = next();
        else if (f == "--n") a.n = std::stoull(next());
        else if (f == "--dim") a.dim = std::stoull(next());
        else if (f == "--nq") a.nq = std::stoull(next());
        else if (f == "--k") a.k = std::stoull(next());
        else if (f == "--threads") a.threads = std::stoull(next());
        else if (f == "--repeat") a.repeat = std::stoull(next());
        else if (f == "--u8") a.u8 = true;
        else { fprintf(stderr, "unknown flag %s\n", f.c_str()); return 2; }
    }
    initLibrary(0, LIB_LOGNONE, nullptr);
    try {
        if (a.u8) return run<int>(a);
        return run<float>(a);
    } catch (const std::exception& e) {
        fprintf(stderr, "ref_driver: %s\n", e.what());
        return 1;
    }
}
