/* Compiled and run by tests/test_gpu_hnsw.py (needs a GPU): plain-C replay of the call sequence of
 * zig/nmslib_gpu_batch.zig (no Zig toolchain in this image).  The batched sequence
 *     initialize_pool -> ONE nmslib_knn_query_batch over a flat copy, caller-owned buffers of capacity k
 * must return exactly what the reference's per-query sequence of lib.zig:889-931
 *     initialize_pool -> for each query: knn_query_get_size + knn_query_fill
 * returns, for a dense float index and for a uint8 index (packed 128-byte rows). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/nmslib_c.h"

static void* a_alloc(size_t n, void* ctx) { (void)ctx; return malloc(n); }
static void a_free(void* p, void* ctx) { (void)ctx; free(p); }
#define CHECK(x) do { nmslib_error_t rc_ = (x); if (rc_ != NMSLIB_SUCCESS) { fprintf(stderr, "%s -> %d\n", #x, (int)rc_); return 10; } } while (0)

static unsigned lcg(unsigned* s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

static int run(int u8) {
    enum { N = 3000, Q = 40, K = 7 };
    const size_t dim = u8 ? 128 : 24, esz = u8 ? 1 : 4;
    nmslib_allocator_t al = {a_alloc, a_free, NULL};
    nmslib_index_handle_t h = NULL;
    CHECK(nmslib_index_create(u8 ? "l2sqr_sift" : "l2", NULL, "hnsw", u8 ? NMSLIB_DATATYPE_DENSE_UINT8_VECTOR : NMSLIB_DATATYPE_DENSE_VECTOR,
                              u8 ? NMSLIB_DISTTYPE_INT : NMSLIB_DISTTYPE_FLOAT, &al, &h));
    /* lib.zig's order: create_index on the empty index, then the rows, then initialize_pool before the queries */
    nmslib_params_handle_t p = nmslib_create_params(&al);
    if (!p) return 1;
    const int M = 8, efc = 60;
    CHECK(nmslib_add_param(p, "M", 0, &M));
    CHECK(nmslib_add_param(p, "efConstruction", 0, &efc));
    CHECK(nmslib_create_index(h, p, 0));
    nmslib_free_params(p);
    unsigned seed = 12345u + (unsigned)u8;
    unsigned char* rows = malloc(N * dim * esz);
    for (size_t i = 0; i < N * dim; ++i) {
        if (u8) rows[i] = (unsigned char)(lcg(&seed) % 200);
        else ((float*)rows)[i] = (float)(lcg(&seed) % 2000) / 1000.0f - 1.0f;
    }
    if (u8) {
        CHECK(nmslib_add_data_point_batch_uint8(h, rows, N, dim, NULL));
    } else {
        CHECK(nmslib_add_data_point_batch(h, rows, N, dim, NULL, NULL));
    }
    nmslib_initialize_pool(h);

    const unsigned char* queries = rows + 100 * dim * esz; /* rows 100..139 as queries */
    /* (a) per query, as lib.zig does */
    int32_t ids_a[Q][K];
    float ds_a[Q][K];
    size_t n_a[Q];
    for (int i = 0; i < Q; ++i) {
        size_t cap = 0;
        CHECK(nmslib_knn_query_get_size(h, queries + i * dim * esz, dim, K, &cap, 0));
        if (cap != K) return 2;
        nmslib_result_t r = {ids_a[i], ds_a[i], 0, cap};
        CHECK(nmslib_knn_query_fill(h, queries + i * dim * esz, dim, K, &r, 0));
        n_a[i] = r.size;
    }
    /* (b) one batch, as zig/nmslib_gpu_batch.zig does */
    unsigned char* flat = malloc(Q * dim * esz);
    memcpy(flat, queries, Q * dim * esz);
    int32_t ids_b[Q][K];
    float ds_b[Q][K];
    nmslib_result_t res[Q];
    for (int i = 0; i < Q; ++i) {
        res[i].ids = ids_b[i];
        res[i].distances = ds_b[i];
        res[i].size = 0;
        res[i].capacity = K;
    }
    nmslib_initialize_pool(h);
    CHECK(nmslib_knn_query_batch(h, flat, Q, dim, K, res, NULL, 0));
    for (int i = 0; i < Q; ++i) {
        if (res[i].size != n_a[i] || res[i].size != K) return 3;
        if (memcmp(ids_a[i], ids_b[i], sizeof(int32_t) * K) || memcmp(ds_a[i], ds_b[i], sizeof(float) * K)) return 4;
        if (!u8 && (ids_b[i][0] != 100 + i || ds_b[i][0] != 0.0f)) return 5; /* a stored row finds itself first */
    }
    free(flat);
    free(rows);
    nmslib_index_destroy(h);
    return 0;
}

int main(void) {
    nmslib_init();
    int rc = run(0);
    if (rc) { fprintf(stderr, "dense: %d\n", rc); return rc; }
    rc = run(1);
    if (rc) { fprintf(stderr, "uint8: %d\n", rc); return 20 + rc; }
    puts("zig batch sequence ok");
    return 0;
}
