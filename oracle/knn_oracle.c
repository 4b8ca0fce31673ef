/* oracle/knn_oracle.c -- TEST INFRASTRUCTURE ONLY (see knn_oracle.h).
 *
 * Scalar C restatement of the reference's dense k-NN hot path.  The SIMD lane
 * structure of each reference kernel is kept (4 SSE lanes / 8 AVX lanes, same
 * horizontal-sum order) and the file is compiled with -ffp-contract=off, which
 * is the arithmetic the reference has when built by clang (zig cc): products
 * and sums are separate roundings.  Integer paths are exact by construction.
 */
#include "knn_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SIFT_DIM 128 /* include/distcomp.h:267 */

/* ======================================================================== */
/* Distance kernels                                                         */
/* ======================================================================== */

/* distcomp_lp.cc:304-365 -- 4 SSE lanes over the first 4*(n/4) elements (the
 * 16-element unrolled loop and the 4-element loop update the same accumulator
 * in the same order), lanes summed [0]+[1]+[2]+[3], scalar tail. */
float orc_l2sqr_simd(const float* a, const float* b, size_t n) {
    float s[4] = {0, 0, 0, 0};
    size_t n4 = n / 4 * 4, i;
    for (i = 0; i < n4; i += 4)
        for (int j = 0; j < 4; ++j) {
            float d = a[i + j] - b[i + j];
            s[j] = s[j] + d * d;
        }
    float res = s[0] + s[1] + s[2] + s[3];
    for (; i < n; ++i) {
        float d = a[i] - b[i];
        res += d * d;
    }
    return res;
}

/* distcomp_lp.cc:367-371 */
float orc_l2_simd(const float* a, const float* b, size_t n) { return sqrtf(orc_l2sqr_simd(a, b, n)); }

/* distcomp_lp.cc:190-251 -- lane sums in float, then a double accumulator for
 * the horizontal sum and the tail, returned as float. */
float orc_l1_simd(const float* a, const float* b, size_t n) {
    float s[4] = {0, 0, 0, 0};
    size_t n4 = n / 4 * 4, i;
    for (i = 0; i < n4; i += 4)
        for (int j = 0; j < 4; ++j) s[j] = s[j] + fabsf(a[i + j] - b[i + j]);
    double res = s[0] + s[1] + s[2] + s[3]; /* float adds, widened once */
    for (; i < n; ++i) res += fabs(a[i] - b[i]); /* float diff widened, double add */
    return (float)res;
}

/* distcomp_lp.cc:77-139 */
float orc_linf_simd(const float* a, const float* b, size_t n) {
    float m[4] = {0, 0, 0, 0};
    size_t n4 = n / 4 * 4, i;
    for (i = 0; i < n4; i += 4)
        for (int j = 0; j < 4; ++j) {
            float d = fabsf(a[i + j] - b[i + j]);
            /* _mm_max_ps(MAX, x): returns x when MAX is not greater */
            m[j] = (m[j] > d) ? m[j] : d;
        }
    float m01 = m[0] > m[1] ? m[0] : m[1], m23 = m[2] > m[3] ? m[2] : m[3];
    float res = m01 > m23 ? m01 : m23;
    for (; i < n; ++i) {
        float d = fabsf(a[i] - b[i]);
        res = res > d ? res : d;
    }
    return res;
}

/* distcomp_scalar.cc:193-245 */
float orc_dot_simd(const float* a, const float* b, size_t n) {
    float s[4] = {0, 0, 0, 0};
    size_t n4 = n / 4 * 4, i;
    for (i = 0; i < n4; i += 4)
        for (int j = 0; j < 4; ++j) s[j] = s[j] + a[i + j] * b[i + j];
    float res = s[0] + s[1] + s[2] + s[3];
    for (; i < n; ++i) res += a[i] * b[i];
    return res;
}

/* distcomp_scalar.cc:83-168 -- dot and both squared norms in one pass; either
 * norm below 2*FLT_MIN gives similarity 0; result clamped to [-1,1]. */
float orc_normdot_simd(const float* a, const float* b, size_t n) {
    float sp[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    size_t n4 = n / 4 * 4, i;
    for (i = 0; i < n4; i += 4)
        for (int j = 0; j < 4; ++j) {
            sp[j] = sp[j] + a[i + j] * b[i + j];
            s1[j] = s1[j] + a[i + j] * a[i + j];
            s2[j] = s2[j] + b[i + j] * b[i + j];
        }
    float sum = sp[0] + sp[1] + sp[2] + sp[3];
    float n1 = s1[0] + s1[1] + s1[2] + s1[3];
    float n2 = s2[0] + s2[1] + s2[2] + s2[3];
    for (; i < n; ++i) {
        sum += a[i] * b[i];
        n1 += a[i] * a[i];
        n2 += b[i] * b[i];
    }
    const float eps = FLT_MIN * 2;
    if (n1 < eps || n2 < eps) return 0;
    float v = sum / sqrtf(n1) / sqrtf(n2);
    v = v < 1.0f ? v : 1.0f;   /* min(1, v) */
    v = v > -1.0f ? v : -1.0f; /* max(-1, .) */
    return v;
}

/* distcomp_scalar.cc:267-271 */
float orc_cosine(const float* a, const float* b, size_t n) {
    float v = 1 - orc_normdot_simd(a, b, n);
    return v > 0 ? v : 0;
}

/* distcomp_scalar.cc:254-258 */
float orc_angular(const float* a, const float* b, size_t n) { return acosf(orc_normdot_simd(a, b, n)); }

/* hnsw_distfunc_opt_impl_inline.h:42-70 (AVX: 8 lanes over 16*(n/16) elements,
 * lanes summed left to right; elements past the last multiple of 16 are ignored,
 * the function is only selected when n % 16 == 0, hnsw.cc:379-385). */
float orc_l2sqr16_avx(const float* a, const float* b, size_t n) {
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t n16 = n / 16 * 16;
    for (size_t i = 0; i < n16; i += 8)
        for (int j = 0; j < 8; ++j) {
            float d = a[i + j] - b[i + j];
            s[j] = s[j] + d * d;
        }
    return s[0] + s[1] + s[2] + s[3] + s[4] + s[5] + s[6] + s[7];
}

/* hnsw_distfunc_opt_impl_inline.h:72-122 */
float orc_l2sqr_avx(const float* a, const float* b, size_t n) {
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t n16 = n / 16 * 16, n4 = n / 4 * 4, i;
    for (i = 0; i < n16; i += 8)
        for (int j = 0; j < 8; ++j) {
            float d = a[i + j] - b[i + j];
            s[j] = s[j] + d * d;
        }
    float t[4];
    for (int j = 0; j < 4; ++j) t[j] = s[j] + s[j + 4];
    for (; i < n4; i += 4)
        for (int j = 0; j < 4; ++j) {
            float d = a[i + j] - b[i + j];
            t[j] = t[j] + d * d;
        }
    float res = t[0] + t[1] + t[2] + t[3];
    for (; i < n; ++i) {
        float d = a[i] - b[i];
        res += d * d;
    }
    return res;
}

/* hnsw_distfunc_opt_impl_inline.h:124-173 */
float orc_dot_avx(const float* a, const float* b, size_t n) {
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t n16 = n / 16 * 16, n4 = n / 4 * 4, i;
    for (i = 0; i < n16; i += 8)
        for (int j = 0; j < 8; ++j) s[j] = s[j] + a[i + j] * b[i + j];
    float t[4];
    for (int j = 0; j < 4; ++j) t[j] = s[j] + s[j + 4];
    for (; i < n4; i += 4)
        for (int j = 0; j < 4; ++j) t[j] = t[j] + a[i + j] * b[i + j];
    float res = t[0] + t[1] + t[2] + t[3];
    for (; i < n; ++i) res += a[i] * b[i];
    return res;
}

/* space_l2sqr_sift.cc:136-150 -- sum of squares stored after the 128 bytes */
int32_t orc_sift_norm(const uint8_t* a) {
    int32_t s = 0;
    for (int i = 0; i < SIFT_DIM; ++i) s += (int32_t)a[i] * (int32_t)a[i];
    return s;
}

/* distcomp_l2sqr_sift.cc:41-50 (the SSE2/AVX2 variants :52-151 are the same
 * integer sum in a different order, hence the same value). */
int32_t orc_l2sqr_sift(const uint8_t* a, int32_t na, const uint8_t* b, int32_t nb) {
    int32_t dot = 0;
    for (int i = 0; i < SIFT_DIM; ++i) dot += (int32_t)a[i] * (int32_t)b[i];
    return na + nb - 2 * dot;
}

/* hnsw.h:486-497 */
void orc_normalize(float* v, size_t n) {
    float sum = 0;
    for (size_t i = 0; i < n; ++i) sum += v[i] * v[i];
    if (sum != 0.0f) {
        sum = 1 / sqrtf(sum);
        for (size_t i = 0; i < n; ++i) v[i] *= sum;
    }
}

/* Space::IndexTimeDistance: space_lp.h:49-67, space_scalar.cc:27-68,
 * space_l2sqr_sift.h:74-79. */
double orc_space_distance(int space, const void* a, const void* b, size_t dim) {
    const float* x = (const float*)a;
    const float* y = (const float*)b;
    switch (space) {
        case ORC_L2: return orc_l2_simd(x, y, dim);
        case ORC_L1: return orc_l1_simd(x, y, dim);
        case ORC_LINF: return orc_linf_simd(x, y, dim);
        case ORC_COSINE: return orc_cosine(x, y, dim);
        case ORC_ANGULAR: return orc_angular(x, y, dim);
        case ORC_NEGDOT: return -orc_dot_simd(x, y, dim);
        case ORC_L2SQR_SIFT: {
            const uint8_t* p = (const uint8_t*)a;
            const uint8_t* q = (const uint8_t*)b;
            return orc_l2sqr_sift(p, orc_sift_norm(p), q, orc_sift_norm(q));
        }
    }
    return NAN;
}

/* hnsw.cc:70-102 (wrappers) + :369-412 (selection). */
double orc_hnsw_opt_distance(int space, const float* q, const float* b, size_t dim) {
    switch (space) {
        case ORC_L2: return (dim % 16 == 0) ? orc_l2sqr16_avx(q, b, dim) : orc_l2sqr_avx(q, b, dim);
        case ORC_L1: return orc_l1_simd(q, b, dim);
        case ORC_LINF: return orc_linf_simd(q, b, dim);
        case ORC_COSINE: { /* NormCosine, hnsw.cc:78-81: inputs pre-normalised */
            float s = orc_dot_avx(q, b, dim);
            s = s < 1.0f ? s : 1.0f;
            s = s > -1.0f ? s : -1.0f;
            float v = 1 - s;
            return v > 0.0f ? v : 0.0f;
        }
        case ORC_NEGDOT: return -orc_dot_avx(q, b, dim);
    }
    return NAN;
}

/* ======================================================================== */
/* Binary heaps with libstdc++'s push_heap / pop_heap element movement, so  */
/* that ties inside std::priority_queue resolve the same way.               */
/* ======================================================================== */
typedef struct {
    double key; /* float and int32 distances are exact in a double */
    int32_t id;
} orc_item;

typedef struct {
    orc_item* v;
    size_t n, cap;
    int mode; /* 0: max-heap on key (HnswNodeDistCloser / EvaluatedMSWNodeInt, hnsw.h:425,448)
                 1: min-heap on key (HnswNodeDistFarther, hnsw.h:404)
                 2: max-heap on (key, id) (KNNQueue's pair<dist, Object*>, knnqueue.h:73-74) */
} orc_heap;

static int heap_less(const orc_heap* h, const orc_item* a, const orc_item* b) {
    switch (h->mode) {
        case 0: return a->key < b->key;
        case 1: return a->key > b->key;
        default: return a->key < b->key || (!(b->key < a->key) && a->id < b->id);
    }
}
static void heap_init(orc_heap* h, int mode) {
    h->v = NULL;
    h->n = h->cap = 0;
    h->mode = mode;
}
static void heap_free(orc_heap* h) {
    free(h->v);
    h->v = NULL;
    h->n = h->cap = 0;
}
static void heap_sift_up(orc_heap* h, size_t hole, size_t top, orc_item val) {
    while (hole > top) {
        size_t parent = (hole - 1) / 2;
        if (!heap_less(h, &h->v[parent], &val)) break;
        h->v[hole] = h->v[parent];
        hole = parent;
    }
    h->v[hole] = val;
}
static void heap_push(orc_heap* h, double key, int32_t id) {
    if (h->n == h->cap) {
        h->cap = h->cap ? h->cap * 2 : 64;
        h->v = (orc_item*)realloc(h->v, h->cap * sizeof(orc_item));
    }
    orc_item it = {key, id};
    h->n++;
    heap_sift_up(h, h->n - 1, 0, it);
}
static void heap_pop(orc_heap* h) { /* std::pop_heap + pop_back */
    if (h->n <= 1) {
        h->n = 0;
        return;
    }
    size_t len = h->n - 1; /* heap length after removal */
    orc_item val = h->v[len];
    size_t hole = 0, child = 0;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (heap_less(h, &h->v[child], &h->v[child - 1])) child--;
        h->v[hole] = h->v[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        h->v[hole] = h->v[child - 1];
        hole = child - 1;
    }
    heap_sift_up(h, hole, 0, val);
    h->n = len;
}

/* ======================================================================== */
/* KNNQuery / KNNQueue (knnquery.cc:66-75, knnqueue.h:55-64)                */
/* ======================================================================== */
typedef struct {
    orc_heap q;
    size_t k;
} orc_knn;

static void knn_init(orc_knn* r, size_t k) {
    heap_init(&r->q, 2);
    r->k = k;
}
static void knn_check_add(orc_knn* r, double d, int32_t pos) {
    if (r->q.n < r->k || d < r->q.v[0].key) {
        if (r->q.n < r->k) {
            heap_push(&r->q, d, pos);
        } else if (r->q.v[0].key > d) {
            heap_pop(&r->q);
            heap_push(&r->q, d, pos);
        }
    }
}
/* extract_knn_results (nmslib_c.cpp:293-328): pop worst-first, reverse. */
static void knn_emit(orc_knn* r, size_t k, int32_t* out_pos, float* out_dist, int32_t* out_cnt) {
    size_t found = r->q.n;
    *out_cnt = (int32_t)found;
    for (size_t j = found; j < k; ++j) {
        out_pos[j] = -1;
        out_dist[j] = INFINITY;
    }
    for (size_t j = found; j-- > 0;) {
        out_pos[j] = r->q.v[0].id;
        out_dist[j] = (float)r->q.v[0].key;
        heap_pop(&r->q);
    }
    heap_free(&r->q);
}

static size_t row_bytes(int space, size_t dim) { return space == ORC_L2SQR_SIFT ? dim : dim * 4; }

/* seqsearch.cc:143-150 */
void orc_seq_search(int space, const void* base, size_t n, size_t dim, const void* queries,
                    size_t nq, size_t k, int32_t* out_pos, float* out_dist, int32_t* out_cnt) {
    const size_t rb = row_bytes(space, dim);
    int32_t* norms = NULL;
    if (space == ORC_L2SQR_SIFT) {
        norms = (int32_t*)malloc(n * sizeof(int32_t));
        for (size_t i = 0; i < n; ++i) norms[i] = orc_sift_norm((const uint8_t*)base + i * rb);
    }
    for (size_t q = 0; q < nq; ++q) {
        const char* qp = (const char*)queries + q * rb;
        orc_knn r;
        knn_init(&r, k);
        int32_t qn = norms ? orc_sift_norm((const uint8_t*)qp) : 0;
        for (size_t i = 0; i < n; ++i) {
            const char* bp = (const char*)base + i * rb;
            double d = norms ? (double)orc_l2sqr_sift((const uint8_t*)bp, norms[i],
                                                      (const uint8_t*)qp, qn)
                             : orc_space_distance(space, bp, qp, dim); /* DistanceObjLeft: (obj, query) */
            knn_check_add(&r, d, (int32_t)i);
        }
        knn_emit(&r, k, out_pos + q * k, out_dist + q * k, out_cnt + q);
    }
    free(norms);
}

/* ======================================================================== */
/* mt19937 + std::uniform_real_distribution<float> (libstdc++)              */
/* ======================================================================== */
typedef struct {
    uint32_t mt[624];
    int idx;
} orc_mt;
static void mt_seed(orc_mt* m, uint32_t seed) {
    m->mt[0] = seed;
    for (int i = 1; i < 624; ++i) m->mt[i] = 1812433253u * (m->mt[i - 1] ^ (m->mt[i - 1] >> 30)) + (uint32_t)i;
    m->idx = 624;
}
static uint32_t mt_next(orc_mt* m) {
    if (m->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (m->mt[i] & 0x80000000u) | (m->mt[(i + 1) % 624] & 0x7fffffffu);
            m->mt[i] = m->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        m->idx = 0;
    }
    uint32_t y = m->mt[m->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
/* utils.h:118-129 RandomReal<float>() = uniform_real_distribution<float>(0,1)
 * over mt19937: generate_canonical<float,24> takes one 32-bit draw. */
static float mt_real_float(orc_mt* m) {
    float sum = (float)mt_next(m);
    float ret = sum / 4294967296.0f;
    if (ret >= 1.0f) ret = nextafterf(1.0f, 0.0f);
    return ret;
}
/* hnsw.h:478-483 */
static int random_level(orc_mt* m, double mult, int log_variant) {
    float u = mt_real_float(m);
    float r = log_variant ? (float)(-log((double)u) * mult) : (float)((double)(-logf(u)) * mult);
    return (int)r;
}
void orc_random_levels(uint32_t seed, double mult, int log_variant, size_t n, int32_t* out) {
    orc_mt m;
    mt_seed(&m, seed);
    for (size_t i = 0; i < n; ++i) out[i] = random_level(&m, mult, log_variant);
}

/* ======================================================================== */
/* HNSW graph                                                               */
/* ======================================================================== */
struct orc_hnsw {
    int space;
    const char* base; /* original rows (index-time distances, generic search) */
    float* norm_base; /* cosine: normalised copy used by the optimized index (hnsw.cc:441-446) */
    int32_t* sift_norms;
    size_t n, dim, rb;
    int M, maxM, maxM0, efC, delaunay;
    int maxlevel, enterpoint;
    int32_t* level;   /* [n] */
    int32_t** links;  /* [n] -> per node: level-0 block (maxM0+2 ints) then per upper level (maxM+2 ints);
                         each block = count, ids...  (+1 slot for the push-then-shrink of addFriendlevel) */
    uint32_t* visited;
    uint32_t epoch;
};

static int32_t* node_links(const orc_hnsw_t* g, int i, int level) {
    return level == 0 ? g->links[i] : g->links[i] + (g->maxM0 + 2) + (size_t)(level - 1) * (g->maxM + 2);
}
static void node_alloc(orc_hnsw_t* g, int i, int level) {
    g->level[i] = level;
    size_t ints = (size_t)(g->maxM0 + 2) + (size_t)level * (g->maxM + 2);
    g->links[i] = (int32_t*)calloc(ints, sizeof(int32_t));
}
static double index_dist(const orc_hnsw_t* g, int a, int b) {
    if (g->sift_norms)
        return orc_l2sqr_sift((const uint8_t*)g->base + (size_t)a * g->rb, g->sift_norms[a],
                              (const uint8_t*)g->base + (size_t)b * g->rb, g->sift_norms[b]);
    return orc_space_distance(g->space, g->base + (size_t)a * g->rb, g->base + (size_t)b * g->rb, g->dim);
}

static orc_hnsw_t* graph_new(int space, const void* base, size_t n, size_t dim, int maxM, int maxM0) {
    orc_hnsw_t* g = (orc_hnsw_t*)calloc(1, sizeof(*g));
    g->space = space;
    g->base = (const char*)base;
    g->n = n;
    g->dim = dim;
    g->rb = row_bytes(space, dim);
    g->maxM = maxM;
    g->maxM0 = maxM0;
    g->level = (int32_t*)calloc(n, sizeof(int32_t));
    g->links = (int32_t**)calloc(n, sizeof(int32_t*));
    g->visited = (uint32_t*)calloc(n + 1, sizeof(uint32_t));
    if (space == ORC_L2SQR_SIFT) {
        g->sift_norms = (int32_t*)malloc(n * sizeof(int32_t));
        for (size_t i = 0; i < n; ++i) g->sift_norms[i] = orc_sift_norm((const uint8_t*)base + i * g->rb);
    }
    if (space == ORC_COSINE) {
        g->norm_base = (float*)malloc(n * dim * sizeof(float));
        memcpy(g->norm_base, base, n * dim * sizeof(float));
        for (size_t i = 0; i < n; ++i) orc_normalize(g->norm_base + i * dim, dim);
    }
    return g;
}

void orc_hnsw_free(orc_hnsw_t* g) {
    if (!g) return;
    for (size_t i = 0; i < g->n; ++i) free(g->links[i]);
    free(g->links);
    free(g->level);
    free(g->visited);
    free(g->sift_norms);
    free(g->norm_base);
    free(g);
}
int orc_hnsw_maxlevel(const orc_hnsw_t* g) { return g->maxlevel; }
int orc_hnsw_enterpoint(const orc_hnsw_t* g) { return g->enterpoint; }
void orc_hnsw_levels(const orc_hnsw_t* g, int32_t* out) { memcpy(out, g->level, g->n * sizeof(int32_t)); }
/* slots past the count are zeroed (the reference leaves them uninitialised, hnsw.cc:461-464) */
static void copy_links(int32_t* out, const int32_t* L, int width) {
    for (int j = 0; j < width; ++j) out[j] = (j <= L[0]) ? L[j] : 0;
}
void orc_hnsw_links0(const orc_hnsw_t* g, int32_t* out) {
    for (size_t i = 0; i < g->n; ++i) copy_links(out + i * (g->maxM0 + 1), g->links[i], g->maxM0 + 1);
}
void orc_hnsw_links_up(const orc_hnsw_t* g, int i, int level, int32_t* out) {
    copy_links(out, node_links(g, i, level), g->maxM + 1);
}

/* HnswNode::getNeighborsByHeuristic2, hnsw.h:129-169.  `rs` is the max-heap
 * result set (HnswNodeDistCloser); on return it holds the selected ones. */
static void heuristic2(orc_hnsw_t* g, orc_heap* rs, size_t NN) {
    if (rs->n < NN) return;
    orc_heap closest;
    heap_init(&closest, 1);
    while (rs->n) {
        heap_push(&closest, rs->v[0].key, rs->v[0].id);
        heap_pop(rs);
    }
    orc_item* ret = (orc_item*)malloc((closest.n + 1) * sizeof(orc_item));
    size_t nret = 0;
    while (closest.n) {
        if (nret >= NN) break;
        orc_item cur = closest.v[0];
        heap_pop(&closest);
        int good = 1;
        for (size_t j = 0; j < nret; ++j) {
            double d = index_dist(g, ret[j].id, cur.id);
            if (d < cur.key) {
                good = 0;
                break;
            }
        }
        if (good) ret[nret++] = cur;
    }
    for (size_t j = 0; j < nret; ++j) heap_push(rs, ret[j].key, ret[j].id);
    free(ret);
    heap_free(&closest);
}

/* HnswNode::addFriendlevel, hnsw.h:258-314 */
static void add_friend(orc_hnsw_t* g, int node, int level, int elem) {
    int32_t* L = node_links(g, node, level);
    for (int i = 0; i < L[0]; ++i)
        if (L[1 + i] == elem) return;
    L[1 + L[0]] = elem;
    L[0]++;
    int maxsz = level > 0 ? g->maxM : g->maxM0;
    if (L[0] <= maxsz) return;
    if (g->delaunay > 0) {
        orc_heap rs;
        heap_init(&rs, 0);
        for (int i = 0; i < L[0]; ++i) heap_push(&rs, index_dist(g, node, L[1 + i]), L[1 + i]);
        heuristic2(g, &rs, rs.n - 1);
        L[0] = 0;
        while (rs.n) { /* refilled farthest first, hnsw.h:297-300 */
            L[1 + L[0]] = rs.v[0].id;
            L[0]++;
            heap_pop(&rs);
        }
        heap_free(&rs);
    } else { /* drop the farthest, hnsw.h:301-312 */
        double mx = index_dist(g, node, L[1]);
        int maxi = 0;
        for (int i = 1; i < L[0]; ++i) {
            double d = index_dist(g, node, L[1 + i]);
            if (d > mx) {
                mx = d;
                maxi = i;
            }
        }
        memmove(&L[1 + maxi], &L[2 + maxi], (size_t)(L[0] - 1 - maxi) * sizeof(int32_t));
        L[0]--;
    }
}

/* Hnsw::kSearchElementsWithAttemptsLevel, hnsw.cc:611-708 */
static void search_level_build(orc_hnsw_t* g, int q, size_t ef, orc_heap* rs, int ep, int level) {
    g->epoch++;
    orc_heap cand;
    heap_init(&cand, 1);
    double d = index_dist(g, q, ep);
    heap_push(&cand, d, ep);
    heap_push(rs, d, ep);
    g->visited[ep] = g->epoch;
    while (cand.n) {
        orc_item cur = cand.v[0];
        double lower = rs->v[0].key;
        if (cur.key > lower) break;
        heap_pop(&cand);
        const int32_t* L = node_links(g, cur.id, level);
        for (int j = 0; j < L[0]; ++j) {
            int nb = L[1 + j];
            if (g->visited[nb] == g->epoch) continue;
            g->visited[nb] = g->epoch;
            d = index_dist(g, q, nb);
            if (rs->n < ef || rs->v[0].key > d) {
                heap_push(rs, d, nb);
                heap_push(&cand, d, nb);
                if (rs->n > ef) heap_pop(rs);
            }
        }
    }
    heap_free(&cand);
}

/* Hnsw::add, hnsw.cc:534-609 (single-threaded: no locks) */
static void hnsw_add(orc_hnsw_t* g, int id, int curlevel) {
    node_alloc(g, id, curlevel);
    int maxlevelcopy = g->maxlevel;
    int ep = g->enterpoint;
    if (curlevel < maxlevelcopy) {
        double curdist = index_dist(g, id, ep);
        int cur = ep;
        for (int level = maxlevelcopy; level > curlevel; --level) {
            int changed = 1;
            while (changed) {
                changed = 0;
                const int32_t* L = node_links(g, cur, level);
                int size = L[0];
                /* like the reference, the whole list of the node the pass started from is scanned */
                for (int i = 0; i < size; ++i) {
                    int nb = L[1 + i];
                    double d = index_dist(g, id, nb);
                    if (d < curdist) {
                        curdist = d;
                        cur = nb;
                        changed = 1;
                    }
                }
            }
        }
        ep = cur;
    }
    int top = curlevel < maxlevelcopy ? curlevel : maxlevelcopy;
    for (int level = top; level >= 0; --level) {
        orc_heap rs;
        heap_init(&rs, 0);
        search_level_build(g, id, (size_t)g->efC, &rs, ep, level);
        if (g->delaunay == 0) {
            while (rs.n > (size_t)g->M) heap_pop(&rs);
        } else {
            heuristic2(g, &rs, (size_t)g->M);
        }
        while (rs.n) {
            ep = rs.v[0].id;
            add_friend(g, rs.v[0].id, level, id); /* link(first=neighbour, second=new), hnsw.h:517-523 */
            add_friend(g, id, level, rs.v[0].id);
            heap_pop(&rs);
        }
        heap_free(&rs);
    }
    if (curlevel > g->level[g->enterpoint]) {
        g->enterpoint = id;
        g->maxlevel = curlevel;
    }
}

orc_hnsw_t* orc_hnsw_build(int space, const void* base, size_t n, size_t dim, int M, int maxM,
                           int maxM0, int efConstruction, int delaunay_type, uint32_t seed,
                           int log_variant) {
    orc_hnsw_t* g = graph_new(space, base, n, dim, maxM, maxM0);
    g->M = M;
    g->efC = efConstruction;
    g->delaunay = delaunay_type;
    if (n == 0) return g;
    orc_mt m;
    mt_seed(&m, seed);
    double mult = 1.0 / log(1.0 * M); /* hnsw.cc:203 */
    node_alloc(g, 0, random_level(&m, mult, log_variant)); /* hnsw.cc:228-232 */
    g->maxlevel = g->level[0];
    g->enterpoint = 0;
    for (size_t id = 1; id < n; ++id) hnsw_add(g, (int)id, random_level(&m, mult, log_variant));
    return g;
}

orc_hnsw_t* orc_hnsw_from_arrays(int space, const void* base, size_t n, size_t dim, int maxM,
                                 int maxM0, int maxlevel, int enterpoint, const int32_t* levels,
                                 const int32_t* links0, const int64_t* up_off,
                                 const int32_t* up_links) {
    orc_hnsw_t* g = graph_new(space, base, n, dim, maxM, maxM0);
    g->maxlevel = maxlevel;
    g->enterpoint = enterpoint;
    for (size_t i = 0; i < n; ++i) {
        node_alloc(g, (int)i, levels[i]);
        memcpy(g->links[i], links0 + i * (maxM0 + 1), (size_t)(maxM0 + 1) * sizeof(int32_t));
        for (int l = 1; l <= levels[i]; ++l)
            memcpy(node_links(g, (int)i, l), up_links + up_off[i] + (size_t)(l - 1) * (maxM + 1),
                   (size_t)(maxM + 1) * sizeof(int32_t));
    }
    return g;
}

/* ======================================================================== */
/* SortArrBI (include/sort_arr_bi.h:30-216)                                 */
/* ======================================================================== */
typedef struct {
    double key;
    int used;
    int32_t data;
} sa_item;
typedef struct {
    sa_item* v;
    size_t cap, n;
} sortarr;

/* push_or_replace_non_empty_exp, sort_arr_bi.h:159-199 */
static size_t sa_push(sortarr* s, double key, int32_t data) {
    size_t curr = s->n - 1;
    if (s->v[curr].key <= key) {
        if (s->n < s->cap) {
            s->v[s->n].used = 0;
            s->v[s->n].key = key;
            s->v[s->n].data = data;
            return s->n++;
        }
        return s->n;
    }
    size_t prev = curr, d = 1;
    while (curr > 0 && s->v[curr].key > key) {
        prev = curr;
        curr -= d;
        d *= 2;
        if (d > curr) d = curr;
    }
    if (curr < prev) { /* std::lower_bound on [curr, prev) */
        size_t lo = curr, hi = prev;
        while (lo < hi) {
            size_t mid = lo + (hi - lo) / 2;
            if (s->v[mid].key < key) lo = mid + 1; else hi = mid;
        }
        curr = lo;
    }
    if (s->n < s->cap) s->n++;
    if (s->n - (1 + curr) > 0) memmove(&s->v[curr + 1], &s->v[curr], (s->n - (1 + curr)) * sizeof(sa_item));
    s->v[curr].used = 0;
    s->v[curr].key = key;
    s->v[curr].data = data;
    return curr;
}

static void stable_merge(sa_item* v, size_t n1, size_t n2) { /* std::inplace_merge */
    sa_item* tmp = (sa_item*)malloc((n1 + n2) * sizeof(sa_item));
    size_t i = 0, j = n1, o = 0;
    while (i < n1 && j < n1 + n2) tmp[o++] = (v[j].key < v[i].key) ? v[j++] : v[i++];
    while (i < n1) tmp[o++] = v[i++];
    while (j < n1 + n2) tmp[o++] = v[j++];
    memcpy(v, tmp, (n1 + n2) * sizeof(sa_item));
    free(tmp);
}

/* merge_with_sorted_items, sort_arr_bi.h:123-155 */
static size_t sa_merge(sortarr* s, const sa_item* items, size_t qty) {
    if (!qty) return s->n;
    if (qty > s->cap) qty = s->cap;
    size_t left = s->cap - s->n;
    if (left >= qty) {
        memcpy(&s->v[s->n], items, qty * sizeof(sa_item));
        stable_merge(s->v, s->n, qty);
        s->n += qty;
    } else {
        size_t rem = 0;
        while (qty > left + rem && s->n > rem && items[left + rem].key < s->v[s->n - rem - 1].key) rem++;
        memcpy(&s->v[s->n - rem], items, (left + rem) * sizeof(sa_item));
        stable_merge(s->v, s->n - rem, s->cap - (s->n - rem));
        s->n = s->cap;
    }
    size_t ret = 0;
    while (ret < s->n && s->v[ret].used) ++ret;
    return ret;
}

static int sa_cmp(const void* a, const void* b) {
    double x = ((const sa_item*)a)->key, y = ((const sa_item*)b)->key;
    return (x > y) - (x < y);
}

/* ======================================================================== */
/* HNSW search                                                              */
/* ======================================================================== */
typedef struct {
    const orc_hnsw_t* g;
    int optimized;
    const float* qf;    /* (normalised, if cosine+optimized) query */
    const uint8_t* qu8;
    int32_t qnorm;
    int64_t ndc;
} qctx;

static double query_dist(qctx* c, int node) {
    const orc_hnsw_t* g = c->g;
    c->ndc++;
    if (g->sift_norms)
        return orc_l2sqr_sift((const uint8_t*)g->base + (size_t)node * g->rb, g->sift_norms[node], c->qu8, c->qnorm);
    if (c->optimized) {
        const float* row = g->norm_base ? g->norm_base + (size_t)node * g->dim
                                        : (const float*)(g->base + (size_t)node * g->rb);
        return orc_hnsw_opt_distance(g->space, c->qf, row, g->dim);
    }
    /* generic path: query->DistanceObjLeft(obj) = space distance (obj, query), query.cc:59-62 */
    return orc_space_distance(g->space, g->base + (size_t)node * g->rb, c->qf, g->dim);
}

/* greedy descent through the upper levels: hnsw_distfunc_opt.cc:173-198 / hnsw.cc:1191-1210 */
static int descend(qctx* c, double* curdist_out) {
    const orc_hnsw_t* g = c->g;
    int cur = g->enterpoint;
    double curdist = query_dist(c, cur);
    for (int lvl = g->maxlevel; lvl > 0; --lvl) {
        int changed = 1;
        while (changed) {
            changed = 0;
            const int32_t* L = node_links(g, cur, lvl);
            int size = L[0];
            for (int j = 1; j <= size; ++j) {
                int t = L[j];
                double d = query_dist(c, t);
                if (d < curdist) {
                    curdist = d;
                    cur = t;
                    changed = 1;
                }
            }
        }
    }
    *curdist_out = curdist;
    return cur;
}

static void search_v1merge(qctx* c, uint32_t* visited, uint32_t epoch, size_t k, size_t ef, orc_knn* res, int64_t* hops) {
    const orc_hnsw_t* g = c->g;
    double curdist;
    int cur = descend(c, &curdist);
    sortarr s;
    s.cap = ef > k ? ef : k;
    s.v = (sa_item*)calloc(s.cap + 1, sizeof(sa_item));
    s.n = 0;
    s.v[0].used = 0; /* push_unsorted_grow, sort_arr_bi.h:61-67 */
    s.v[0].key = curdist;
    s.v[0].data = cur;
    s.n = 1;
    size_t buffcap = 1 + (size_t)(g->maxM > g->maxM0 ? g->maxM : g->maxM0);
    sa_item* buff = (sa_item*)malloc(buffcap * sizeof(sa_item));
    long currElem = 0;
    visited[cur] = epoch;
    while (currElem < (long)(s.n < ef ? s.n : ef)) {
        sa_item* e = &s.v[currElem];
        e->used = 1;
        cur = e->data;
        ++currElem;
        (*hops)++;
        size_t qty = 0;
        double topKey = s.v[s.n - 1].key;
        const int32_t* L = node_links(g, cur, 0);
        int size = L[0];
        for (int j = 1; j <= size; ++j) {
            int t = L[j];
            if (visited[t] == epoch) continue;
            visited[t] = epoch;
            double d = query_dist(c, t);
            if (d < topKey || s.n < ef) {
                buff[qty].key = d;
                buff[qty].used = 0;
                buff[qty].data = t;
                qty++;
            }
        }
        if (qty) {
            qsort(buff, qty, sizeof(sa_item), sa_cmp); /* std::sort: ties unordered */
            if (qty > 100) { /* MERGE_BUFFER_ALGO_SWITCH_THRESHOLD, hnsw_distfunc_opt.cc:36 */
                size_t ins = sa_merge(&s, buff, qty);
                if ((long)ins < currElem) currElem = (long)ins;
            } else {
                for (size_t ii = 0; ii < qty; ++ii) {
                    size_t ins = sa_push(&s, buff[ii].key, buff[ii].data);
                    if ((long)ins < currElem) currElem = (long)ins;
                }
            }
        }
        while (currElem < (long)s.n && s.v[currElem].used) ++currElem;
    }
    for (size_t i = 0; i < k && i < s.n; ++i) knn_check_add(res, s.v[i].key, s.v[i].data);
    free(buff);
    free(s.v);
}

static void search_old(qctx* c, uint32_t* visited, uint32_t epoch, size_t ef, orc_knn* res, int64_t* hops) {
    const orc_hnsw_t* g = c->g;
    double curdist;
    int cur = descend(c, &curdist);
    orc_heap cand, closest;
    heap_init(&cand, 1);    /* candidateQueue: smallest distance on top */
    heap_init(&closest, 0); /* closestDistQueue: largest on top */
    heap_push(&cand, curdist, cur);
    heap_push(&closest, curdist, cur);
    knn_check_add(res, curdist, cur);
    visited[cur] = epoch;
    while (cand.n) {
        orc_item ev = cand.v[0];
        double lower = closest.v[0].key;
        if (ev.key > lower) break;
        heap_pop(&cand);
        (*hops)++;
        const int32_t* L = node_links(g, ev.id, 0);
        int size = L[0];
        for (int j = 1; j <= size; ++j) {
            int t = L[j];
            if (visited[t] == epoch) continue;
            visited[t] = epoch;
            double d = query_dist(c, t);
            if (closest.v[0].key > d || closest.n < ef) {
                heap_push(&cand, d, t);
                knn_check_add(res, d, t);
                heap_push(&closest, d, t);
                if (closest.n > ef) heap_pop(&closest);
            }
        }
    }
    heap_free(&cand);
    heap_free(&closest);
}

void orc_hnsw_search(const orc_hnsw_t* g, int optimized, int algo, const void* queries,
                     size_t nq, size_t k, size_t ef, int32_t* out_pos, float* out_dist,
                     int32_t* out_cnt, int64_t* out_ndc, int64_t* out_hops) {
    uint32_t* visited = (uint32_t*)calloc(g->n + 1, sizeof(uint32_t));
    float* qbuf = (float*)malloc((g->dim + 1) * sizeof(float));
    for (size_t q = 0; q < nq; ++q) {
        qctx c;
        memset(&c, 0, sizeof(c));
        c.g = g;
        c.optimized = optimized && !g->sift_norms;
        if (g->sift_norms) {
            c.qu8 = (const uint8_t*)queries + q * g->rb;
            c.qnorm = orc_sift_norm(c.qu8);
        } else {
            memcpy(qbuf, (const char*)queries + q * g->rb, g->rb);
            if (c.optimized && g->space == ORC_COSINE) orc_normalize(qbuf, g->dim); /* hnsw_distfunc_opt.cc:160-162 */
            c.qf = qbuf;
        }
        orc_knn res;
        knn_init(&res, k);
        int64_t hops = 0;
        if (g->n) {
            if (algo == 0) search_v1merge(&c, visited, (uint32_t)q + 1, k, ef, &res, &hops);
            else search_old(&c, visited, (uint32_t)q + 1, ef, &res, &hops);
        }
        knn_emit(&res, k, out_pos + q * k, out_dist + q * k, out_cnt + q);
        if (out_ndc) out_ndc[q] = c.ndc;
        if (out_hops) out_hops[q] = hops;
    }
    free(qbuf);
    free(visited);
}
