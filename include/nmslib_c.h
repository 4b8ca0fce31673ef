/*
 * nmslib_c.h -- the drop-in boundary of the MI355X k-NN engine.
 *
 * This header declares, with identical names, argument order and enum values,
 * the C ABI that B-R-P/NMSLIB-ZIG's lib.zig binds through @cImport
 * (reference: nmslib_c.h:12-535, lib.zig:5-8).  A program linked against the
 * reference's static library can be re-linked against libnmslib_c.so from this
 * repository without source changes.  Behind these entry points the k-NN hot
 * path (distance kernels, sequential scan, HNSW search) runs as HIP kernels on
 * gfx950; there is no CPU search path in the library.
 *
 * Every declaration names the reference definition it replaces
 * (file:line in /root/reference).  Deviations from the reference's behaviour are
 * listed in INTEGRATION.md ("Behavioural differences").
 */
#ifndef NMSLIB_C_H
#define NMSLIB_C_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums (values are ABI: nmslib_c.h:12-46) ----------------------------- */

typedef enum {
    NMSLIB_DATATYPE_DENSE_VECTOR,        /* float32 rows                      */
    NMSLIB_DATATYPE_SPARSE_VECTOR,       /* accepted by the ABI, not by the GPU engine */
    NMSLIB_DATATYPE_DENSE_UINT8_VECTOR,  /* 128-byte SIFT descriptors         */
    NMSLIB_DATATYPE_OBJECT_AS_STRING     /* accepted by the ABI, not by the GPU engine */
} nmslib_data_type_t;

typedef enum { NMSLIB_DISTTYPE_FLOAT, NMSLIB_DISTTYPE_INT } nmslib_dist_type_t;

typedef enum {
    NMSLIB_SUCCESS = 0,
    NMSLIB_ERROR_NULL_POINTER = 1,
    NMSLIB_ERROR_INVALID_ARGUMENT = 2,
    NMSLIB_ERROR_OUT_OF_MEMORY = 3,
    NMSLIB_ERROR_BUFFER_TOO_SMALL = 4,
    NMSLIB_ERROR_SPACE_INCOMPATIBLE = 5,
    NMSLIB_ERROR_QUERY_TOO_LARGE = 6,
    NMSLIB_ERROR_INVALID_SPARSE_ELEMENT = 7,
    NMSLIB_ERROR_INDEX_BUILD_FAILED = 8,
    NMSLIB_ERROR_QUERY_EXECUTION_FAILED = 9,
    NMSLIB_ERROR_DATA_IO_FAILED = 10,
    NMSLIB_ERROR_PLUGIN_REGISTRATION_FAILED = 11,
    NMSLIB_ERROR_INTERNAL = 12,
    NMSLIB_ERROR_RUNTIME = 13,
    NMSLIB_ERROR_INDEX_NOT_BUILT = 14
} nmslib_error_t;

typedef enum {
    NMSLIB_DATA_MODE_DENSE_FLOAT = 0,
    NMSLIB_DATA_MODE_SPARSE = 1,
    NMSLIB_DATA_MODE_UINT8 = 2
} nmslib_data_mode_t;

/* ---- plain structs (layout is ABI: nmslib_c.h:49-81) ---------------------- */

typedef struct {
    uint32_t id;
    float value;
} nmslib_sparse_elem_float_t;

/* Caller-owned result buffers; `capacity` slots each, `size` filled in. */
typedef struct {
    int32_t* ids;
    float* distances;
    size_t size;
    size_t capacity;
} nmslib_result_t;

/* Every handle, parameter object and returned string lives in memory obtained
 * from the caller's allocator (lib.zig:192-257 tracks each block). */
typedef struct {
    void* (*alloc)(size_t size, void* ctx);
    void (*free)(void* ptr, void* ctx);
    void* ctx;
} nmslib_allocator_t;

typedef struct {
    nmslib_error_t code;
    const char* message;
    const char* file;
    int line;
} nmslib_error_detail_t;

/* First bytes of every index handle (nmslib_c.cpp:136-139,197-198). */
typedef struct {
    nmslib_data_type_t data_type;
    nmslib_dist_type_t dist_type;
} nmslib_index_header_t;

typedef struct nmslib_index_t* nmslib_index_handle_t;
typedef struct nmslib_params_t* nmslib_params_handle_t;

/* ---- life cycle ------------------------------------------------------------ */

/* nmslib_c.cpp:339 -- idempotent library initialisation. */
void nmslib_init(void);

/* nmslib_c.cpp:341-447 -- create an empty index for (space, method).  Dense
 * float spaces: l2, l1, linf, cosinesimil, angulardist, negdotprod; uint8
 * space: l2sqr_sift.  Methods: hnsw, brute_force / seq_search. */
nmslib_error_t nmslib_index_create(const char* space, nmslib_params_handle_t space_params,
                                   const char* method, nmslib_data_type_t data_type,
                                   nmslib_dist_type_t dist_type,
                                   const nmslib_allocator_t* allocator,
                                   nmslib_index_handle_t* out_handle);

/* nmslib_c.cpp:449-477 */
void nmslib_index_destroy(nmslib_index_handle_t handle);

/* nmslib_c.cpp:479-517 -- parse index-time parameters and build over the data
 * added so far (HNSW: M, efConstruction, maxM, maxM0, mult, delaunay_type, post,
 * indexThreadQty, skip_optimized_index, searchMethod; brute force: copyMem,
 * multiThread, threadQty).  Unknown names fail, as AnyParamManager::CheckUnused
 * does (include/params.h:241-251). */
nmslib_error_t nmslib_create_index(nmslib_index_handle_t index,
                                   nmslib_params_handle_t index_params, int print_progress);

/* nmslib_c.cpp:519-538 */
nmslib_error_t nmslib_reset_index(nmslib_index_handle_t index);

/* ---- parameters (nmslib_c.cpp:540-614) ------------------------------------- */

nmslib_params_handle_t nmslib_create_params(const nmslib_allocator_t* allocator);
/* type: 0 = int (const int*), 1 = double (const double*), 2 = C string. */
nmslib_error_t nmslib_add_param(nmslib_params_handle_t params, const char* name, int type,
                                const void* value);
void nmslib_free_params(nmslib_params_handle_t params);

/* ---- metadata / errors (nmslib_c.cpp:616-715) ------------------------------ */

nmslib_error_t nmslib_get_space_type(nmslib_index_handle_t index, const char** space_type,
                                     size_t* space_type_len,
                                     const nmslib_allocator_t* allocator);
nmslib_error_t nmslib_get_method(nmslib_index_handle_t index, const char** method,
                                 size_t* method_len, const nmslib_allocator_t* allocator);
void nmslib_free_string(char* str, const nmslib_allocator_t* allocator);
nmslib_error_t nmslib_get_last_error_detail(nmslib_error_detail_t* detail,
                                            const nmslib_allocator_t* allocator);

/* ---- adding data: rows are copied (nmslib_c.cpp:717-918, 1567-1669) -------- */

nmslib_error_t nmslib_add_data_point(nmslib_index_handle_t index, const void* data,
                                     size_t element_count, int32_t id);
nmslib_error_t nmslib_add_data_point_batch(nmslib_index_handle_t index, const void* data,
                                           size_t count, size_t element_count,
                                           const int32_t* ids, const size_t* num_elements);
nmslib_error_t nmslib_add_data_point_batch_uint8(nmslib_index_handle_t index,
                                                 const unsigned char* data, size_t count,
                                                 size_t element_count, const int32_t* ids);
nmslib_error_t nmslib_add_data_point_batch_string(nmslib_index_handle_t index,
                                                  const char* const* data, size_t count,
                                                  const int32_t* ids);
nmslib_error_t nmslib_add_data_point_batch_pointers(nmslib_index_handle_t handle,
                                                    nmslib_data_mode_t data_mode,
                                                    const void* const* data_ptrs, size_t count,
                                                    size_t element_count, const int32_t* ids,
                                                    const size_t* num_elements);

/* ---- k-NN queries: THE HOT ENTRY POINTS (nmslib_c.cpp:920-1031) ------------- */

nmslib_error_t nmslib_knn_query_get_size(nmslib_index_handle_t index, const void* query,
                                         size_t query_size_or_elem_count, size_t k,
                                         size_t* out_size, size_t num_elements);
/* One query (nmslib_c.cpp:941-1001).  Runs as a batch of one on the GPU. */
nmslib_error_t nmslib_knn_query_fill(nmslib_index_handle_t index, const void* query,
                                     size_t query_size_or_elem_count, size_t k,
                                     nmslib_result_t* result, size_t num_elements);
/* `query_count` queries stored back to back (nmslib_c.cpp:1003-1031): one GPU
 * batch.  Rows are strided by the index's element size (4 for float, 1 for
 * uint8); the reference strides uint8 rows by 4*elem_count, nmslib_c.cpp:1018-1019. */
nmslib_error_t nmslib_knn_query_batch(nmslib_index_handle_t index, const void* queries,
                                      size_t query_count, size_t query_size_or_elem_count,
                                      size_t k, nmslib_result_t* results,
                                      const size_t* num_elements, size_t thread_pool_size);

/* ---- range queries (nmslib_c.cpp:1033-1153) -------------------------------- */

nmslib_error_t nmslib_range_query_get_size(nmslib_index_handle_t index, const void* query,
                                           size_t query_size_or_elem_count, double radius,
                                           size_t* out_size, size_t num_elements);
nmslib_error_t nmslib_range_query_fill(nmslib_index_handle_t index, const void* query,
                                       size_t query_size_or_elem_count, double radius,
                                       nmslib_result_t* result, size_t num_elements);

/* ---- stored data access (nmslib_c.cpp:1155-1367) --------------------------- */

nmslib_error_t nmslib_get_distance(nmslib_index_handle_t index, size_t pos1, size_t pos2,
                                   float* distance);
nmslib_error_t nmslib_get_data_point_size(nmslib_index_handle_t index, size_t position,
                                          size_t* size);
nmslib_error_t nmslib_get_data_point_fill(nmslib_index_handle_t index, size_t position,
                                          void* data, size_t size);
nmslib_error_t nmslib_get_data_point_string(nmslib_index_handle_t index, size_t position,
                                            const char** data, size_t* data_len,
                                            const nmslib_allocator_t* allocator);
nmslib_error_t nmslib_borrow_data_dense(nmslib_index_handle_t index, size_t position,
                                        void** data, size_t* size, void (**free_fn)(void*));
nmslib_error_t nmslib_borrow_data_sparse(nmslib_index_handle_t index, size_t position,
                                         void** data, size_t* size, void (**free_fn)(void*));

/* ---- persistence (nmslib_c.cpp:1369-1479; formats hnsw.cc:774-806, space.cc:88-105) */

nmslib_error_t nmslib_save_index(nmslib_index_handle_t index, const char* path, int save_data);
nmslib_error_t nmslib_load_index(const char* path, nmslib_data_type_t data_type,
                                 nmslib_dist_type_t dist_type,
                                 const nmslib_allocator_t* allocator, int load_data,
                                 nmslib_index_handle_t* out_handle);

/* ---- knobs (nmslib_c.cpp:1481-1565) ---------------------------------------- */

/* HNSW: ef / efSearch, algoType in {old, v1merge, hybrid}, searchMethod. */
nmslib_error_t nmslib_set_query_time_params(nmslib_index_handle_t index,
                                            nmslib_params_handle_t params);
nmslib_error_t nmslib_set_thread_pool_size(nmslib_index_handle_t index, size_t size);
size_t nmslib_get_thread_pool_size(nmslib_index_handle_t index);
size_t nmslib_data_qty(nmslib_index_handle_t index);
size_t nmslib_index_memory_usage(nmslib_index_handle_t handle);

/* nmslib_c.cpp:1682-1704 -- lib.zig calls this before every query (lib.zig:802).
 * Here it finalises a dirty index (uploads new rows, builds the graph) once;
 * it is a no-op on a clean index. */
void nmslib_initialize_pool(nmslib_index_handle_t index);

/* Defined by the reference (nmslib_c.cpp:1671-1680) and declared `extern` by
 * lib.zig:8, although absent from the reference header. */
void nmslib_free_result(nmslib_result_t* result, const nmslib_allocator_t* allocator);

#ifdef __cplusplus
}
#endif
#endif /* NMSLIB_C_H */
