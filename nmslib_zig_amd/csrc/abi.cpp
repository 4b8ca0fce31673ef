// The nmslib_c.h C ABI (include/nmslib_c.h) over the GPU engine.
//
// Conventions kept from the reference shim (nmslib_c.cpp): every function catches all
// exceptions and returns an nmslib_error_t; details go to a thread_local record read back
// with nmslib_get_last_error_detail (:36-41,684-715); handles, parameter objects and returned
// strings live in caller-allocator memory (:355-364,547-558); rows are copied on add.
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <thread>
#include <string>
#include <vector>

#include "../../include/nmslib_c.h"
#include "../../include/nmslib_gpu.h"
#include "engine.hpp"

using gfxknn::Engine;
using gfxknn::EngineError;
using gfxknn::Err;

namespace {

struct LastError {
    nmslib_error_t code = NMSLIB_SUCCESS;
    std::string message = "No error";
    std::string file = __FILE__;
    int line = 0;
};
thread_local LastError g_last;

void set_last(nmslib_error_t code, const std::string& msg, int line) {
    g_last.code = code;
    g_last.message = msg.empty() ? "No error" : msg;
    g_last.file = __FILE__;
    g_last.line = line;
}
#define SET_LAST(code, msg) set_last(code, msg, __LINE__)
#define FAIL(code, msg)      \
    do {                     \
        SET_LAST(code, msg); \
        return code;         \
    } while (0)

// What an index handle points at.  The header must come first: callers (and the reference's
// own dispatch, nmslib_c.cpp:197-198) read data_type/dist_type from the first bytes.
struct HandleBlock {
    nmslib_index_header_t header;
    Engine* engine;
    nmslib_allocator_t allocator;
};
struct ParamsBlock {
    std::vector<std::string>* params;
    nmslib_allocator_t allocator;
};

HandleBlock* H(nmslib_index_handle_t h) { return reinterpret_cast<HandleBlock*>(h); }
ParamsBlock* P(nmslib_params_handle_t p) { return reinterpret_cast<ParamsBlock*>(p); }
std::vector<std::string> params_of(nmslib_params_handle_t p) {
    return p ? *P(p)->params : std::vector<std::string>();
}

char* dup_string(const std::string& s, const nmslib_allocator_t* a) {
    char* r = static_cast<char*>(a->alloc(s.size() + 1, a->ctx));
    if (!r) return nullptr;
    std::memcpy(r, s.c_str(), s.size() + 1);
    return r;
}

nmslib_error_t map_err(Err e) { return static_cast<nmslib_error_t>(static_cast<int>(e)); }

// Runs fn(); maps exceptions.  `generic` is what a plain std::exception becomes in this entry
// point (the reference uses a different code per function).
template <typename Fn>
nmslib_error_t guarded(nmslib_error_t generic, const char* what, Fn&& fn) {
    try {
        fn();
        SET_LAST(NMSLIB_SUCCESS, std::string(what) + ": ok");
        return NMSLIB_SUCCESS;
    } catch (const std::bad_alloc& e) {
        FAIL(NMSLIB_ERROR_OUT_OF_MEMORY, std::string("Memory allocation failed: ") + e.what());
    } catch (const EngineError& e) {
        // engine errors carry their own code unless the entry point pins one
        nmslib_error_t c = map_err(e.code);
        if (e.code == Err::Runtime && generic != NMSLIB_ERROR_RUNTIME) c = generic;
        FAIL(c, std::string(what) + ": " + e.what());
    } catch (const std::exception& e) {
        FAIL(generic, std::string(what) + ": " + e.what());
    } catch (...) {
        FAIL(generic, std::string(what) + ": unknown error");
    }
}

void borrowed_free(void* ptr);

}  // namespace

extern "C" {

void nmslib_init(void) {}

nmslib_error_t nmslib_index_create(const char* space, nmslib_params_handle_t space_params, const char* method,
                                   nmslib_data_type_t data_type, nmslib_dist_type_t dist_type,
                                   const nmslib_allocator_t* allocator, nmslib_index_handle_t* out_handle) {
    if (!space || !method || !allocator || !allocator->alloc || !allocator->free || !out_handle)
        FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid arguments");
    (void)space_params;  // dense space factories ignore their params, e.g. "dim" (factory/space/space_lp.h:37-40)
    Engine* eng = nullptr;
    try {
        eng = new Engine(space, method, (int)data_type, (int)dist_type);
    } catch (const std::bad_alloc&) {
        FAIL(NMSLIB_ERROR_OUT_OF_MEMORY, "Failed to allocate index");
    } catch (const std::exception& e) {
        FAIL(NMSLIB_ERROR_SPACE_INCOMPATIBLE, std::string("Failed to create space: ") + e.what());
    }
    void* mem = allocator->alloc(sizeof(HandleBlock), allocator->ctx);
    if (!mem) {
        delete eng;
        FAIL(NMSLIB_ERROR_OUT_OF_MEMORY, "Failed to allocate index");
    }
    HandleBlock* hb = new (mem) HandleBlock{{data_type, dist_type}, eng, *allocator};
    *out_handle = reinterpret_cast<nmslib_index_handle_t>(hb);
    SET_LAST(NMSLIB_SUCCESS, "Index created");
    return NMSLIB_SUCCESS;
}

void nmslib_index_destroy(nmslib_index_handle_t handle) {
    if (!handle) return;
    HandleBlock* hb = H(handle);
    delete hb->engine;
    hb->engine = nullptr;
    nmslib_allocator_t a = hb->allocator;
    a.free(hb, a.ctx);
}

nmslib_error_t nmslib_create_index(nmslib_index_handle_t handle, nmslib_params_handle_t index_params,
                                   int print_progress) {
    (void)print_progress;
    if (!handle) FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid index");
    return guarded(NMSLIB_ERROR_INDEX_BUILD_FAILED, "Failed to create index", [&] {
        Engine* e = H(handle)->engine;
        std::lock_guard<std::mutex> lk(e->mu);
        e->create_index(params_of(index_params));
    });
}

nmslib_error_t nmslib_reset_index(nmslib_index_handle_t handle) {
    if (!handle) FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid index");
    return guarded(NMSLIB_ERROR_RUNTIME, "Failed to reset index", [&] {
        Engine* e = H(handle)->engine;
        std::lock_guard<std::mutex> lk(e->mu);
        e->reset();
    });
}

nmslib_params_handle_t nmslib_create_params(const nmslib_allocator_t* allocator) {
    if (!allocator || !allocator->alloc || !allocator->free) {
        SET_LAST(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid allocator");
        return nullptr;
    }
    void* mem = allocator->alloc(sizeof(ParamsBlock), allocator->ctx);
    if (!mem) {
        SET_LAST(NMSLIB_ERROR_OUT_OF_MEMORY, "Failed to allocate memory for params");
        return nullptr;
    }
    std::vector<std::string>* v = new (std::nothrow) std::vector<std::string>();
    if (!v) {
        allocator->free(mem, allocator->ctx);
        SET_LAST(NMSLIB_ERROR_OUT_OF_MEMORY, "Failed to create params");
        return nullptr;
    }
    ParamsBlock* pb = new (mem) ParamsBlock{v, *allocator};
    SET_LAST(NMSLIB_SUCCESS, "Parameters created successfully");
    return reinterpret_cast<nmslib_params_handle_t>(pb);
}

nmslib_error_t nmslib_add_param(nmslib_params_handle_t params, const char* name, int type, const void* value) {
    if (!params || !name || !value) FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid arguments");
    return guarded(NMSLIB_ERROR_RUNTIME, "Failed to add parameter", [&] {
        std::string p = std::string(name) + "=";
        switch (type) {  // nmslib_c.cpp:575-589
            case 0: p += std::to_string(*static_cast<const int*>(value)); break;
            case 1: p += std::to_string(*static_cast<const double*>(value)); break;
            case 2: p += static_cast<const char*>(value); break;
            default: throw EngineError(Err::InvalidArgument, "Invalid parameter type");
        }
        P(params)->params->push_back(p);
    });
}

void nmslib_free_params(nmslib_params_handle_t params) {
    if (!params || !P(params)->allocator.free) {
        SET_LAST(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid params or allocator");
        return;
    }
    ParamsBlock* pb = P(params);
    delete pb->params;
    nmslib_allocator_t a = pb->allocator;
    a.free(pb, a.ctx);
}

static nmslib_error_t get_string(const std::string& s, const char** out, size_t* out_len,
                                 const nmslib_allocator_t* allocator, const char* what) {
    *out_len = s.size();
    *out = dup_string(s, allocator);
    if (!*out) FAIL(NMSLIB_ERROR_OUT_OF_MEMORY, std::string("Failed to allocate memory for ") + what);
    SET_LAST(NMSLIB_SUCCESS, std::string(what) + " retrieved successfully");
    return NMSLIB_SUCCESS;
}

nmslib_error_t nmslib_get_space_type(nmslib_index_handle_t handle, const char** space_type, size_t* space_type_len,
                                     const nmslib_allocator_t* allocator) {
    if (!handle || !space_type || !space_type_len || !allocator || !allocator->alloc || !allocator->free)
        FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid arguments");
    return get_string(H(handle)->engine->space_name(), space_type, space_type_len, allocator, "space type");
}

nmslib_error_t nmslib_get_method(nmslib_index_handle_t handle, const char** method, size_t* method_len,
                                 const nmslib_allocator_t* allocator) {
    if (!handle || !method || !method_len || !allocator || !allocator->alloc || !allocator->free)
        FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid arguments");
    return get_string(H(handle)->engine->method_name(), method, method_len, allocator, "method");
}

void nmslib_free_string(char* str, const nmslib_allocator_t* allocator) {
    if (!str || !allocator || !allocator->free) {
        SET_LAST(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid string or allocator");
        return;
    }
    allocator->free(str, allocator->ctx);
}

nmslib_error_t nmslib_get_last_error_detail(nmslib_error_detail_t* detail, const nmslib_allocator_t* allocator) {
    if (!detail || !allocator || !allocator->alloc || !allocator->free)
        FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid detail or allocator pointer");
    detail->code = g_last.code;
    detail->message = dup_string(g_last.message, allocator);
    if (!detail->message) FAIL(NMSLIB_ERROR_OUT_OF_MEMORY, "Failed to allocate memory for error message");
    detail->file = dup_string(g_last.file, allocator);
    if (!detail->file) {
        allocator->free(const_cast<char*>(detail->message), allocator->ctx);
        FAIL(NMSLIB_ERROR_OUT_OF_MEMORY, "Failed to allocate memory for error file");
    }
    detail->line = g_last.line;
    SET_LAST(NMSLIB_SUCCESS, "Error detail retrieved successfully");
    return NMSLIB_SUCCESS;
}

// ---- adding data ---------------------------------------------------------------------------

nmslib_error_t nmslib_add_data_point(nmslib_index_handle_t handle, const void* data, size_t element_count,
                                     int32_t id) {
    if (!handle || !data || element_count == 0) FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid inputs for adding data point");
    return guarded(NMSLIB_ERROR_RUNTIME, "Failed to add data point", [&] {
        Engine* e = H(handle)->engine;
        std::lock_guard<std::mutex> lk(e->mu);
        e->add_row(data, element_count, id);
    });
}

nmslib_error_t nmslib_add_data_point_batch(nmslib_index_handle_t handle, const void* data, size_t count,
                                           size_t element_count, const int32_t* ids, const size_t* num_elements) {
    (void)num_elements;  // sparse only
    if (!handle || !data || count == 0 || element_count == 0) FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid batch inputs");
    return guarded(NMSLIB_ERROR_RUNTIME, "Failed to add batch", [&] {
        Engine* e = H(handle)->engine;
        std::lock_guard<std::mutex> lk(e->mu);
        const size_t stride = element_count * e->elem_bytes();  // nmslib_c.cpp:778-789
        for (size_t i = 0; i < count; ++i)
            e->add_row(static_cast<const char*>(data) + i * stride, element_count, ids ? ids[i] : (int32_t)i);
    });
}

nmslib_error_t nmslib_add_data_point_batch_uint8(nmslib_index_handle_t handle, const unsigned char* data,
                                                 size_t count, size_t element_count, const int32_t* ids) {
    if (!handle || !data || count == 0 || element_count == 0)
        FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid uint8 batch inputs");
    if (H(handle)->header.data_type != NMSLIB_DATATYPE_DENSE_UINT8_VECTOR)
        FAIL(NMSLIB_ERROR_SPACE_INCOMPATIBLE, "Not uint8 vector space");
    return guarded(NMSLIB_ERROR_RUNTIME, "Failed to add uint8 batch", [&] {
        Engine* e = H(handle)->engine;
        std::lock_guard<std::mutex> lk(e->mu);
        for (size_t i = 0; i < count; ++i)
            e->add_row(data + i * element_count, element_count, ids ? ids[i] : (int32_t)i);
    });
}

nmslib_error_t nmslib_add_data_point_batch_string(nmslib_index_handle_t handle, const char* const* data,
                                                  size_t count, const int32_t* ids) {
    (void)ids;
    if (!handle || !data || count == 0) FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid string batch inputs");
    FAIL(NMSLIB_ERROR_SPACE_INCOMPATIBLE, "Not string space");  // nmslib_c.cpp:883-887
}

nmslib_error_t nmslib_add_data_point_batch_pointers(nmslib_index_handle_t handle, nmslib_data_mode_t data_mode,
                                                    const void* const* data_ptrs, size_t count,
                                                    size_t element_count, const int32_t* ids,
                                                    const size_t* num_elements) {
    (void)num_elements;
    if (!handle || !data_ptrs || count == 0) FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid pointer batch inputs");
    for (size_t i = 0; i < count; ++i)
        if (!data_ptrs[i]) FAIL(NMSLIB_ERROR_NULL_POINTER, "Null pointer in batch");
    const nmslib_data_type_t dt = H(handle)->header.data_type;
    if (data_mode == NMSLIB_DATA_MODE_SPARSE) FAIL(NMSLIB_ERROR_SPACE_INCOMPATIBLE, "Not sparse space");
    if (data_mode == NMSLIB_DATA_MODE_DENSE_FLOAT && dt != NMSLIB_DATATYPE_DENSE_VECTOR)
        FAIL(NMSLIB_ERROR_SPACE_INCOMPATIBLE, "Not dense float space");
    if (data_mode == NMSLIB_DATA_MODE_UINT8 && dt != NMSLIB_DATATYPE_DENSE_UINT8_VECTOR)
        FAIL(NMSLIB_ERROR_SPACE_INCOMPATIBLE, "Not uint8 space");
    if (data_mode != NMSLIB_DATA_MODE_DENSE_FLOAT && data_mode != NMSLIB_DATA_MODE_UINT8)
        FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Unsupported data mode");
    if (element_count == 0) FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "element_count must be positive");
    return guarded(NMSLIB_ERROR_RUNTIME, "Failed to add pointer batch", [&] {
        Engine* e = H(handle)->engine;
        std::lock_guard<std::mutex> lk(e->mu);
        for (size_t i = 0; i < count; ++i) e->add_row(data_ptrs[i], element_count, ids ? ids[i] : (int32_t)i);
    });
}

// ---- k-NN ------------------------------------------------------------------------------------

nmslib_error_t nmslib_knn_query_get_size(nmslib_index_handle_t index, const void* query,
                                         size_t query_size_or_elem_count, size_t k, size_t* out_size,
                                         size_t num_elements) {
    (void)query_size_or_elem_count;
    (void)num_elements;
    if (!index || !query || k == 0 || !out_size) FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid knn query inputs");
    *out_size = k;  // nmslib_c.cpp:931
    SET_LAST(NMSLIB_SUCCESS, "KNN size retrieved");
    return NMSLIB_SUCCESS;
}

// extract_knn_results (nmslib_c.cpp:293-328): an undersized buffer yields size = 0 and SUCCESS
static void fill_result(nmslib_result_t* r, const int32_t* ids, const float* dists, size_t found) {
    r->size = found;
    if (found == 0) return;
    if (found > r->capacity) {
        SET_LAST(NMSLIB_ERROR_BUFFER_TOO_SMALL, "Result buffers too small for " + std::to_string(found));
        r->size = 0;
        return;
    }
    std::memcpy(r->ids, ids, found * sizeof(int32_t));
    std::memcpy(r->distances, dists, found * sizeof(float));
}

nmslib_error_t nmslib_knn_query_batch(nmslib_index_handle_t index, const void* queries, size_t query_count,
                                      size_t query_size_or_elem_count, size_t k, nmslib_result_t* results,
                                      const size_t* num_elements, size_t thread_pool_size) {
    (void)num_elements;
    (void)thread_pool_size;
    if (!index || !queries || query_count == 0 || !results) FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid batch knn inputs");
    if (query_size_or_elem_count == 0 || k == 0) FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid KNN query inputs");
    for (size_t i = 0; i < query_count; ++i)
        if (!results[i].ids || !results[i].distances || results[i].capacity == 0)
            FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Result buffers invalid");
    Engine* e = H(index)->engine;
    if (!e->index_created()) FAIL(NMSLIB_ERROR_INDEX_BUILD_FAILED, "Index not built");  // nmslib_c.cpp:963-967
    bool too_small = false;
    nmslib_error_t rc = guarded(NMSLIB_ERROR_QUERY_EXECUTION_FAILED, "KNN query failed", [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        const int32_t *ids = nullptr, *cnt = nullptr;
        const float* dists = nullptr;
        e->knn_host(queries, query_count, query_size_or_elem_count, k, &ids, &dists, &cnt);
        for (size_t i = 0; i < query_count; ++i) {
            fill_result(&results[i], &ids[i * k], &dists[i * k], (size_t)cnt[i]);
            too_small |= ((size_t)cnt[i] > results[i].capacity);
        }
    });
    if (rc != NMSLIB_SUCCESS) {
        for (size_t i = 0; i < query_count; ++i) results[i].size = 0;
        return rc;
    }
    if (too_small) SET_LAST(NMSLIB_ERROR_BUFFER_TOO_SMALL, "Result buffers too small");
    return NMSLIB_SUCCESS;
}

nmslib_error_t nmslib_knn_query_fill(nmslib_index_handle_t index, const void* query,
                                     size_t query_size_or_elem_count, size_t k, nmslib_result_t* result,
                                     size_t num_elements) {
    (void)num_elements;
    if (!index || !query || query_size_or_elem_count == 0 || !result)
        FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid KNN query inputs");
    if (!result->ids || !result->distances || result->capacity == 0)
        FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Result buffers invalid");
    if (k == 0) FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid KNN query inputs");
    return nmslib_knn_query_batch(index, query, 1, query_size_or_elem_count, k, result, nullptr, 0);
}

// ---- range queries -----------------------------------------------------------------------------

nmslib_error_t nmslib_range_query_get_size(nmslib_index_handle_t index, const void* query,
                                           size_t query_size_or_elem_count, double radius, size_t* out_size,
                                           size_t num_elements) {
    (void)query_size_or_elem_count;
    (void)num_elements;
    if (!index || !query || radius < 0 || !out_size) FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid range query inputs");
    *out_size = 128;  // nmslib_c.cpp:1045
    SET_LAST(NMSLIB_SUCCESS, "Range query size estimated");
    return NMSLIB_SUCCESS;
}

nmslib_error_t nmslib_range_query_fill(nmslib_index_handle_t index, const void* query,
                                       size_t query_size_or_elem_count, double radius, nmslib_result_t* result,
                                       size_t num_elements) {
    (void)num_elements;
    if (!index || !query || !result || result->capacity == 0) FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid range fill inputs");
    Engine* e = H(index)->engine;
    if (!e->index_created()) FAIL(NMSLIB_ERROR_INDEX_BUILD_FAILED, "Index not built");
    result->size = 0;
    if (e->method_name() == "hnsw")  // Hnsw::Search(RangeQuery*) throws (hnsw.cc:710-715) -> nmslib_c.cpp:1131-1141
        FAIL(NMSLIB_ERROR_SPACE_INCOMPATIBLE, "Range query not supported by method: Range search is not supported!");
    try {
        std::lock_guard<std::mutex> lk(e->mu);
        result->size = e->range_host(query, query_size_or_elem_count, radius, result->capacity, result->ids,
                                     result->distances);
    } catch (const EngineError& ex) {
        FAIL(static_cast<nmslib_error_t>(ex.code), std::string("Range query exception: ") + ex.what());
    } catch (const std::bad_alloc& ex) {
        FAIL(NMSLIB_ERROR_OUT_OF_MEMORY, std::string("Range query alloc failed: ") + ex.what());
    } catch (const std::exception& ex) {
        FAIL(NMSLIB_ERROR_RUNTIME, std::string("Range query exception: ") + ex.what());
    }
    SET_LAST(NMSLIB_SUCCESS, "Range query filled successfully");
    return NMSLIB_SUCCESS;
}

// ---- stored data -------------------------------------------------------------------------------

nmslib_error_t nmslib_get_distance(nmslib_index_handle_t index, size_t pos1, size_t pos2, float* distance) {
    if (!index || pos1 >= nmslib_data_qty(index) || pos2 >= nmslib_data_qty(index) || !distance)
        FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid distance inputs");
    return guarded(NMSLIB_ERROR_RUNTIME, "Failed to compute distance", [&] {
        Engine* e = H(index)->engine;
        std::lock_guard<std::mutex> lk(e->mu);
        *distance = e->pair_distance(pos1, pos2);
    });
}

nmslib_error_t nmslib_get_data_point_size(nmslib_index_handle_t index, size_t position, size_t* size) {
    if (!index || position >= nmslib_data_qty(index) || !size) FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid data point size inputs");
    *size = H(index)->engine->stored_row_bytes();  // Object::datalength()
    SET_LAST(NMSLIB_SUCCESS, "Data point size retrieved");
    return NMSLIB_SUCCESS;
}

nmslib_error_t nmslib_get_data_point_fill(nmslib_index_handle_t index, size_t position, void* data, size_t size) {
    if (!index || !data || size == 0 || position >= nmslib_data_qty(index))
        FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid data point fill inputs");
    Engine* e = H(index)->engine;
    if (size < e->stored_row_bytes()) FAIL(NMSLIB_ERROR_BUFFER_TOO_SMALL, "Buffer too small for data point");
    e->stored_row(position, data);
    SET_LAST(NMSLIB_SUCCESS, "Data point filled");
    return NMSLIB_SUCCESS;
}

nmslib_error_t nmslib_get_data_point_string(nmslib_index_handle_t index, size_t position, const char** data,
                                            size_t* data_len, const nmslib_allocator_t* allocator) {
    if (!index || !data || !data_len || !allocator || position >= nmslib_data_qty(index))
        FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid string data point inputs");
    FAIL(NMSLIB_ERROR_SPACE_INCOMPATIBLE, "Invalid data type for string");
}

namespace {
// borrowed copies: [allocator][payload]; free_fn receives the payload pointer
struct BorrowHeader {
    nmslib_allocator_t allocator;
    uint64_t magic;
};
constexpr uint64_t kBorrowMagic = 0x676678626f72726fULL;
void borrowed_free(void* ptr) {
    if (!ptr) return;
    BorrowHeader* h = reinterpret_cast<BorrowHeader*>(static_cast<char*>(ptr) - sizeof(BorrowHeader));
    if (h->magic != kBorrowMagic) return;
    nmslib_allocator_t a = h->allocator;
    h->magic = 0;
    a.free(h, a.ctx);
}
}  // namespace

nmslib_error_t nmslib_borrow_data_dense(nmslib_index_handle_t index, size_t position, void** data, size_t* size,
                                        void (**free_fn)(void*)) {
    if (!index || !data || !size || !free_fn || position >= nmslib_data_qty(index))
        FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid dense borrow inputs");
    HandleBlock* hb = H(index);
    if (hb->header.data_type != NMSLIB_DATATYPE_DENSE_VECTOR) FAIL(NMSLIB_ERROR_SPACE_INCOMPATIBLE, "Not dense vector");
    Engine* e = hb->engine;
    const size_t bytes = e->row_bytes();
    char* blk = static_cast<char*>(hb->allocator.alloc(sizeof(BorrowHeader) + bytes, hb->allocator.ctx));
    if (!blk) FAIL(NMSLIB_ERROR_OUT_OF_MEMORY, "Failed to allocate data copy");
    BorrowHeader* bh = reinterpret_cast<BorrowHeader*>(blk);
    bh->allocator = hb->allocator;
    bh->magic = kBorrowMagic;
    std::memcpy(blk + sizeof(BorrowHeader), e->host_row(position), bytes);
    *data = blk + sizeof(BorrowHeader);
    *size = bytes;
    *free_fn = borrowed_free;
    SET_LAST(NMSLIB_SUCCESS, "Dense data borrowed");
    return NMSLIB_SUCCESS;
}

nmslib_error_t nmslib_borrow_data_sparse(nmslib_index_handle_t index, size_t position, void** data, size_t* size,
                                         void (**free_fn)(void*)) {
    if (!index || !data || !size || !free_fn || position >= nmslib_data_qty(index))
        FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid sparse borrow inputs");
    FAIL(NMSLIB_ERROR_SPACE_INCOMPATIBLE, "Not sparse vector");
}

// ---- persistence -------------------------------------------------------------------------------

nmslib_error_t nmslib_save_index(nmslib_index_handle_t handle, const char* path, int save_data) {
    if (!handle || !path) FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid save inputs");
    Engine* e = H(handle)->engine;
    if (!e->index_created()) FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Index not built");  // nmslib_c.cpp:1378-1382
    return guarded(NMSLIB_ERROR_DATA_IO_FAILED, "Failed to save index", [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        try {
            e->save(path, save_data != 0);
        } catch (const EngineError& x) {
            throw EngineError(Err::DataIO, x.what());  // every failure here is DATA_IO (nmslib_c.cpp:1391-1395)
        }
    });
}

nmslib_error_t nmslib_load_index(const char* path, nmslib_data_type_t data_type, nmslib_dist_type_t dist_type,
                                 const nmslib_allocator_t* allocator, int load_data,
                                 nmslib_index_handle_t* out_handle) {
    if (!path || !allocator || !allocator->alloc || !allocator->free || !out_handle)
        FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid load inputs");
    if (dist_type != NMSLIB_DISTTYPE_FLOAT && dist_type != NMSLIB_DISTTYPE_INT)
        FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid dist type for load");
    std::unique_ptr<Engine> eng;
    nmslib_error_t rc = guarded(NMSLIB_ERROR_DATA_IO_FAILED, "Failed to load index", [&] {
        try {
            eng = Engine::load(path, (int)data_type, (int)dist_type, load_data != 0);
        } catch (const EngineError& x) {
            throw EngineError(Err::DataIO, x.what());
        }
    });
    if (rc != NMSLIB_SUCCESS) return rc;
    void* mem = allocator->alloc(sizeof(HandleBlock), allocator->ctx);
    if (!mem) FAIL(NMSLIB_ERROR_OUT_OF_MEMORY, "Failed to allocate index");
    HandleBlock* hb = new (mem) HandleBlock{{data_type, dist_type}, eng.release(), *allocator};
    *out_handle = reinterpret_cast<nmslib_index_handle_t>(hb);
    SET_LAST(NMSLIB_SUCCESS, "Index loaded successfully");
    return NMSLIB_SUCCESS;
}

// ---- knobs ---------------------------------------------------------------------------------------

nmslib_error_t nmslib_set_query_time_params(nmslib_index_handle_t handle, nmslib_params_handle_t params) {
    if (!handle) FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid index");
    Engine* e = H(handle)->engine;
    if (!e->index_created()) FAIL(NMSLIB_ERROR_INDEX_BUILD_FAILED, "Index not built");
    return guarded(NMSLIB_ERROR_RUNTIME, "Failed to set query time params", [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        e->set_query_params(params_of(params));
    });
}

nmslib_error_t nmslib_set_thread_pool_size(nmslib_index_handle_t handle, size_t size) {
    if (!handle || size == 0 || size > 1024) FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid thread pool size");
    H(handle)->engine->thread_pool_size = size;  // stored, unused: nmslib_c.cpp:1507-1535
    SET_LAST(NMSLIB_SUCCESS, "Thread pool size set");
    return NMSLIB_SUCCESS;
}

size_t nmslib_get_thread_pool_size(nmslib_index_handle_t handle) {
    if (!handle) {
        SET_LAST(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid index");
        return std::thread::hardware_concurrency();
    }
    return H(handle)->engine->thread_pool_size;
}

size_t nmslib_data_qty(nmslib_index_handle_t handle) {
    if (!handle) {
        SET_LAST(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid index");
        return 0;
    }
    return H(handle)->engine->size();
}

size_t nmslib_index_memory_usage(nmslib_index_handle_t handle) {
    if (!handle) return 0;
    return H(handle)->engine->memory_usage();
}

void nmslib_initialize_pool(nmslib_index_handle_t handle) {
    if (!handle) return;
    Engine* e = H(handle)->engine;
    if (!e->index_created()) return;
    try {
        std::lock_guard<std::mutex> lk(e->mu);
        e->finalize();
    } catch (const std::exception& x) {
        fprintf(stderr, "NMSLIB pool initialization failed: %s\n", x.what());
        SET_LAST(NMSLIB_ERROR_RUNTIME, std::string("pool initialization failed: ") + x.what());
    }
}

void nmslib_free_result(nmslib_result_t* result, const nmslib_allocator_t* allocator) {
    if (!result || !allocator || !allocator->free) return;
    if (result->ids) allocator->free(result->ids, allocator->ctx);
    if (result->distances) allocator->free(result->distances, allocator->ctx);
    result->ids = nullptr;
    result->distances = nullptr;
    result->size = 0;
    result->capacity = 0;
}

// ---- device-resident extensions (include/nmslib_gpu.h) ----------------------------------------

int nmslib_gpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

nmslib_error_t nmslib_gpu_finalize(nmslib_index_handle_t handle) {
    if (!handle) FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid index");
    return guarded(NMSLIB_ERROR_INDEX_BUILD_FAILED, "finalize", [&] {
        Engine* e = H(handle)->engine;
        std::lock_guard<std::mutex> lk(e->mu);
        e->finalize();
    });
}

nmslib_error_t nmslib_gpu_knn_query_batch_device(nmslib_index_handle_t handle, const void* d_queries,
                                                 size_t query_count, size_t elem_count, size_t k,
                                                 int32_t* d_ids, float* d_dists, int32_t* d_counts, void* stream) {
    if (!handle || !d_queries || !d_ids || !d_dists || k == 0 || elem_count == 0)
        FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid device batch inputs");
    Engine* e = H(handle)->engine;
    if (!e->index_created()) FAIL(NMSLIB_ERROR_INDEX_BUILD_FAILED, "Index not built");
    return guarded(NMSLIB_ERROR_QUERY_EXECUTION_FAILED, "KNN device batch failed", [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        e->knn_device(d_queries, query_count, elem_count, k, d_ids, d_dists, d_counts,
                      static_cast<hipStream_t>(stream));
    });
}

nmslib_error_t nmslib_gpu_last_batch_counters(nmslib_index_handle_t handle, const int32_t** d_ndc,
                                              const int32_t** d_hops, const int32_t** d_hops_up) {
    if (!handle) FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid index");
    Engine* e = H(handle)->engine;
    if (d_ndc) *d_ndc = e->last_ndc();
    if (d_hops) *d_hops = e->last_hops();
    if (d_hops_up) *d_hops_up = e->last_hops_up();
    return NMSLIB_SUCCESS;
}

nmslib_error_t nmslib_gpu_merge_topk_strided(const float* d_dists_in, const int32_t* d_ids_in, size_t shard_stride,
                                             size_t nshards, size_t query_count, size_t k, float* d_dists_out,
                                             int32_t* d_ids_out, void* stream) {
    if (!d_dists_in || !d_ids_in || !d_dists_out || !d_ids_out || nshards == 0 || k == 0 || shard_stride < query_count * k)
        FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid merge inputs");
    if (nshards * k > 8192) FAIL(NMSLIB_ERROR_QUERY_TOO_LARGE, "nshards * k must not exceed 8192");
    return guarded(NMSLIB_ERROR_RUNTIME, "merge_topk", [&] {
        if (query_count)
            gfxknn::hip_check(gfxknn::launch_merge_topk(d_dists_in, d_ids_in, shard_stride, (int)nshards, (int)query_count,
                                                        (int)k, d_dists_out, d_ids_out, static_cast<hipStream_t>(stream)),
                              "merge_topk");
    });
}

nmslib_error_t nmslib_gpu_merge_topk(const float* d_dists_in, const int32_t* d_ids_in, size_t nshards,
                                     size_t query_count, size_t k, float* d_dists_out, int32_t* d_ids_out,
                                     void* stream) {
    return nmslib_gpu_merge_topk_strided(d_dists_in, d_ids_in, query_count * k, nshards, query_count, k, d_dists_out,
                                         d_ids_out, stream);
}

nmslib_error_t nmslib_gpu_kernel_timing(nmslib_index_handle_t handle, int enable, double* total_ms,
                                        uint64_t* launches) {
    if (!handle) FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid index");
    return guarded(NMSLIB_ERROR_RUNTIME, "kernel_timing", [&] {
        Engine* e = H(handle)->engine;
        std::lock_guard<std::mutex> lk(e->mu);
        if (total_ms || launches) e->collect_profile(total_ms, launches);
        e->set_profiling(enable != 0);
    });
}

nmslib_error_t nmslib_gpu_get_stats(nmslib_index_handle_t handle, nmslib_gpu_stats_t* out) {
    if (!handle || !out) FAIL(NMSLIB_ERROR_INVALID_ARGUMENT, "Invalid arguments");
    Engine* e = H(handle)->engine;
    std::memset(out, 0, sizeof(*out));
    return guarded(NMSLIB_ERROR_RUNTIME, "Failed to read statistics", [&] {
        std::lock_guard<std::mutex> lk(e->mu);
        out->upload_seconds = e->upload_seconds;
        out->build_seconds = e->build_seconds;
        out->hbm_bytes = e->hbm_bytes();
        out->rows = e->size();
        out->dim = e->dim();
        out->shards = e->shard_count() ? e->shard_count() : 1;
        out->last_path = (size_t)e->last_path;
        e->fast_tile_counts(&out->fast_tiles, &out->fast_tiles_precise, &out->fast_tiles_fallback);
        out->hnsw_redone = e->hnsw_redone();
    });
}

}  // extern "C"
