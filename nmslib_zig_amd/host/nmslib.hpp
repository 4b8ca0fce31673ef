// nmslib.hpp -- header-only C++17 mirror of the reference's Zig binding (lib.zig:260-1270) over the
// same C ABI (include/nmslib_c.h).  The reference's host language is Zig; no Zig toolchain exists in
// this image, so the operator surface (Params, Index.init / addDenseBatch / addUInt8Batch /
// buildIndex / knnQuery / knnQueryBatch / getDistance / save / load / ...) is mirrored here in
// C++ with the same names, argument meaning and error mapping (lib.zig:56-73).  Nothing in this
// header computes: every call is one C-ABI call into libnmslib_c.so.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/nmslib_c.h"

namespace nmslib {

// error.NullPointer ... error.IndexNotBuilt (lib.zig:11-27,56-73)
struct Error : std::runtime_error {
    nmslib_error_t code;
    Error(nmslib_error_t c, const std::string& what) : std::runtime_error(what), code(c) {}
};

enum class DataType { DenseVector = 0, SparseVector = 1, DenseUInt8Vector = 2, ObjectAsString = 3 };
enum class DistType { Float = 0, Int = 1 };

// The allocator bridge of lib.zig:192-257: every block the library takes is tracked so leaks show.
class TrackingAllocator {
   public:
    TrackingAllocator() : c_{&TrackingAllocator::alloc, &TrackingAllocator::release, this} {}
    const nmslib_allocator_t* c() const { return &c_; }
    size_t live() const { return live_.size(); }

   private:
    static void* alloc(size_t n, void* ctx) {
        void* p = std::malloc(n ? n : 1);
        if (p) static_cast<TrackingAllocator*>(ctx)->live_[p] = n;
        return p;
    }
    static void release(void* p, void* ctx) {
        if (!p) return;
        static_cast<TrackingAllocator*>(ctx)->live_.erase(p);
        std::free(p);
    }
    nmslib_allocator_t c_;
    std::map<void*, size_t> live_;
};

inline void check(nmslib_error_t rc, const TrackingAllocator& a, const char* what) {
    if (rc == NMSLIB_SUCCESS) return;
    std::string msg = what;
    nmslib_error_detail_t d{};
    if (nmslib_get_last_error_detail(&d, a.c()) == NMSLIB_SUCCESS) {
        msg += ": ";
        msg += d.message;
        nmslib_free_string(const_cast<char*>(d.message), a.c());
        nmslib_free_string(const_cast<char*>(d.file), a.c());
    }
    throw Error(rc, msg);
}

// lib.zig:260-348
class Params {
   public:
    explicit Params(const TrackingAllocator& a) : a_(a), h_(nmslib_create_params(a.c())) {
        if (!h_) throw Error(NMSLIB_ERROR_OUT_OF_MEMORY, "nmslib_create_params");
    }
    ~Params() { if (h_) nmslib_free_params(h_); }
    Params(const Params&) = delete;
    Params& operator=(const Params&) = delete;
    Params& add(const std::string& k, int v) { check(nmslib_add_param(h_, k.c_str(), 0, &v), a_, "add_param"); return *this; }
    Params& add(const std::string& k, double v) { check(nmslib_add_param(h_, k.c_str(), 1, &v), a_, "add_param"); return *this; }
    Params& add(const std::string& k, const std::string& v) { check(nmslib_add_param(h_, k.c_str(), 2, v.c_str()), a_, "add_param"); return *this; }
    nmslib_params_handle_t handle() const { return h_; }

   private:
    const TrackingAllocator& a_;
    nmslib_params_handle_t h_;
};

struct QueryResult {  // lib.zig:380-401
    std::vector<int32_t> ids;
    std::vector<float> distances;
};

// lib.zig:495-1270
class Index {
   public:
    Index(std::string space, const std::string& method, DataType dt = DataType::DenseVector,
          DistType dist = DistType::Float)
        : data_type_(dt) {
        nmslib_init();
        if (space == "cosine") space = "cosinesimil";  // lib.zig:530-533
        check(nmslib_index_create(space.c_str(), nullptr, method.c_str(), static_cast<nmslib_data_type_t>(dt),
                                  static_cast<nmslib_dist_type_t>(dist), alloc_.c(), &h_),
              alloc_, "nmslib_index_create");
    }
    ~Index() { if (h_) nmslib_index_destroy(h_); }
    Index(const Index&) = delete;
    Index& operator=(const Index&) = delete;

    void addDenseBatch(const float* rows, size_t count, size_t dim, const int32_t* ids = nullptr) {
        check(nmslib_add_data_point_batch(h_, rows, count, dim, ids, nullptr), alloc_, "addDenseBatch");
    }
    void addUInt8Batch(const uint8_t* rows, size_t count, size_t dim, const int32_t* ids = nullptr) {
        check(nmslib_add_data_point_batch_uint8(h_, rows, count, dim, ids), alloc_, "addUInt8Batch");
    }
    void buildIndex(const Params* p = nullptr, bool print_progress = false) {
        check(nmslib_create_index(h_, p ? p->handle() : nullptr, print_progress ? 1 : 0), alloc_, "buildIndex");
    }
    void setQueryTimeParams(const Params& p) { check(nmslib_set_query_time_params(h_, p.handle()), alloc_, "setQueryTimeParams"); }

    QueryResult knnQuery(const void* query, size_t elem_count, size_t k) {
        nmslib_initialize_pool(h_);  // lib.zig:802
        size_t cap = 0;
        check(nmslib_knn_query_get_size(h_, query, elem_count, k, &cap, 0), alloc_, "knn_query_get_size");
        QueryResult r;
        r.ids.resize(cap);
        r.distances.resize(cap);
        nmslib_result_t c{r.ids.data(), r.distances.data(), 0, cap};
        check(nmslib_knn_query_fill(h_, query, elem_count, k, &c, 0), alloc_, "knnQuery");
        r.ids.resize(c.size);
        r.distances.resize(c.size);
        return r;
    }
    // lib.zig:933-965: the size estimate (128) sizes the buffers; HNSW -> Error(SPACE_INCOMPATIBLE)
    QueryResult rangeQuery(const void* query, size_t elem_count, double radius) {
        nmslib_initialize_pool(h_);
        size_t cap = 0;
        check(nmslib_range_query_get_size(h_, query, elem_count, radius, &cap, 0), alloc_, "range_query_get_size");
        QueryResult r;
        r.ids.resize(cap);
        r.distances.resize(cap);
        nmslib_result_t c{r.ids.data(), r.distances.data(), 0, cap};
        check(nmslib_range_query_fill(h_, query, elem_count, radius, &c, 0), alloc_, "rangeQuery");
        r.ids.resize(c.size);
        r.distances.resize(c.size);
        return r;
    }
    // One GPU batch (the reference loops per query, lib.zig:889-931).
    std::vector<QueryResult> knnQueryBatch(const void* queries, size_t count, size_t elem_count, size_t k) {
        std::vector<QueryResult> out(count);
        std::vector<nmslib_result_t> res(count);
        for (size_t i = 0; i < count; ++i) {
            out[i].ids.resize(k);
            out[i].distances.resize(k);
            res[i] = nmslib_result_t{out[i].ids.data(), out[i].distances.data(), 0, k};
        }
        check(nmslib_knn_query_batch(h_, queries, count, elem_count, k, res.data(), nullptr, 0), alloc_, "knnQueryBatch");
        for (size_t i = 0; i < count; ++i) {
            out[i].ids.resize(res[i].size);
            out[i].distances.resize(res[i].size);
        }
        return out;
    }
    float getDistance(size_t a, size_t b) {
        float d = 0;
        check(nmslib_get_distance(h_, a, b, &d), alloc_, "getDistance");
        return d;
    }
    size_t dataQty() const { return nmslib_data_qty(h_); }
    std::string getSpaceType() { return str(&nmslib_get_space_type) == "cosinesimil" ? "cosine" : str(&nmslib_get_space_type); }
    std::string getMethod() { return str(&nmslib_get_method); }
    void save(const std::string& path, bool save_data) { check(nmslib_save_index(h_, path.c_str(), save_data), alloc_, "save"); }
    void setThreadPoolSize(size_t n) { check(nmslib_set_thread_pool_size(h_, n), alloc_, "setThreadPoolSize"); }
    size_t getThreadPoolSize() const { return nmslib_get_thread_pool_size(h_); }
    size_t liveBlocks() const { return alloc_.live(); }
    nmslib_index_handle_t handle() const { return h_; }

   private:
    template <typename Fn>
    std::string str(Fn fn) {
        const char* p = nullptr;
        size_t n = 0;
        check(fn(h_, &p, &n, alloc_.c()), alloc_, "get string");
        std::string s(p, n);
        nmslib_free_string(const_cast<char*>(p), alloc_.c());
        return s;
    }
    TrackingAllocator alloc_;
    nmslib_index_handle_t h_ = nullptr;
    DataType data_type_;
};

}  // namespace nmslib
