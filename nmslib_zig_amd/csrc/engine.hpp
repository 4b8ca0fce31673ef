// Host side of the MI355X k-NN engine: index objects whose rows and graphs live in HBM.
//
// Data model (deliberately not NMSLIB's Object* soup, include/object.h:41-104): rows are one
// row-major array in HBM (stride padded to 32 bytes), external ids a parallel int32 array, the
// HNSW graph two fixed-stride int32 arrays.  The host keeps a copy of the rows only to serve
// nmslib_get_data_point / borrow / save and to construct the graph.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "kernels/kernels.hpp"

namespace gfxknn {

// ---- errors: thrown inside the engine, mapped to nmslib_error_t at the ABI ----------------
enum class Err : int {
    InvalidArgument = 2,
    OutOfMemory = 3,
    SpaceIncompatible = 5,
    QueryTooLarge = 6,
    IndexBuildFailed = 8,
    QueryExecutionFailed = 9,
    DataIO = 10,
    Runtime = 13,
};
struct EngineError : std::runtime_error {
    Err code;
    EngineError(Err c, const std::string& m) : std::runtime_error(m), code(c) {}
};

void hip_check(hipError_t e, const char* what);  // throws EngineError(Runtime / OutOfMemory)

// ---- "name=value" parameters (include/params.h:44-74,181-251) -------------------------------
class ParamSet {
   public:
    ParamSet() = default;
    explicit ParamSet(const std::vector<std::string>& desc);  // throws on bad format / duplicates
    bool has(const std::string& name) const;
    // Typed optional getters; conversion failures throw like ConvertStrToValue (params.h:289-299).
    void get(const std::string& name, long long& v);
    void get(const std::string& name, int& v);
    void get(const std::string& name, size_t& v);
    void get(const std::string& name, double& v);
    void get(const std::string& name, bool& v);
    void get(const std::string& name, std::string& v);
    void check_unused() const;  // AnyParamManager::CheckUnused (params.h:241-251)

   private:
    const std::string* find(const std::string& name);
    std::vector<std::pair<std::string, std::string>> kv_;
    std::vector<bool> seen_;
};

// ---- device buffer ---------------------------------------------------------------------------
class DevBuf {
   public:
    DevBuf() = default;
    ~DevBuf() { release(); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    void* ensure(size_t bytes);  // grow-only
    void release();
    void* ptr() const { return p_; }
    template <typename T>
    T* as() const {
        return static_cast<T*>(p_);
    }
    size_t bytes() const { return n_; }

   private:
    void* p_ = nullptr;
    size_t n_ = 0;
};

// ---- HNSW graph on the host (flat arrays; same layout goes to HBM) ---------------------------
struct HostGraph {
    int n = 0, M = 16, maxM = 16, maxM0 = 32, efConstruction = 200, delaunay = 2;
    int maxlevel = 0, enterpoint = 0;
    std::vector<int32_t> levels;    // [n]
    std::vector<int32_t> links0;    // [n][maxM0+1] = count, ids...
    std::vector<int64_t> up_off;    // [n] offset (ints) into up_links, -1 if level 0 only
    std::vector<int32_t> up_links;  // per node: level blocks of (maxM+1) ints
    bool empty() const { return n == 0; }
};

struct HnswBuildParams {
    int M = 16, maxM = 16, maxM0 = 32, efConstruction = 200, delaunay = 2, post = 0;
    int threads = 0;  // 0 = hardware concurrency
    double mult = 0;  // 0 = 1/ln(M)
    bool skip_optimized = false;
    int gpu_build = -1;       // engine extension: 1 = batched construction on the GPU, 0 = host, -1 = auto
    int gpu_build_batch = 0;  // max nodes inserted per batch (0 = default)
    int gpu_build_div = 0;    // a batch is at most 1/div of the graph built so far (0 = default)
};


void hnsw_check_params(const HnswBuildParams& bp);
std::vector<int32_t> hnsw_random_levels(size_t n, const HnswBuildParams& bp);

// Construct the graph (restates Hnsw::add / kSearchElementsWithAttemptsLevel /
// getNeighborsByHeuristic2 / addFriendlevel, src/method/hnsw.cc:534-708, include/method/hnsw.h:
// 129-169,258-314) with `threads` workers.  space = index-time SpaceCode.  rows: f32 [n][dim]
// or u8 [n][128].
void hnsw_build_host(int space, const void* rows, size_t n, size_t dim, const HnswBuildParams& bp,
                     HostGraph& out);

// ---- the index ---------------------------------------------------------------------------------
enum class Method { Brute, Hnsw };

class Engine {
   public:
    Engine(const std::string& space, const std::string& method, int data_type, int dist_type);
    ~Engine();

    const std::string& space_name() const { return space_name_; }
    const std::string& method_name() const { return method_name_; }
    bool is_u8() const { return space_ == SP_L2SQR_SIFT; }
    size_t size() const { return parent_ ? view_n_ : ids_.size(); }
    size_t dim() const { return dim_; }
    size_t elem_bytes() const { return is_u8() ? 1 : 4; }
    size_t row_bytes() const { return dim_ * elem_bytes(); }
    size_t stored_row_bytes() const { return is_u8() ? dim_ + 4 : dim_ * 4; }  // u8 rows carry their norm

    void add_row(const void* data, size_t elem_count, int32_t id);
    const void* host_row(size_t pos) const;
    void stored_row(size_t pos, void* dst) const;  // payload as the reference stores it
    int32_t ext_id(size_t pos) const { return ids_[pos]; }
    void reset();

    void create_index(const std::vector<std::string>& params);  // nmslib_create_index
    void set_query_params(const std::vector<std::string>& params);
    bool index_created() const { return created_; }
    void finalize();  // upload + build if dirty (nmslib_initialize_pool / lazy)

    // k-NN: device-resident batch (the hot entry) and host convenience wrapper
    void knn_device(const void* d_queries, size_t nq, size_t elem_count, size_t k, int32_t* d_ids,
                    float* d_dists, int32_t* d_cnt, hipStream_t stream);
    // host buffers in, results in this engine's pinned staging block (valid until the next call; the caller holds `mu`)
    void knn_host(const void* queries, size_t nq, size_t elem_count, size_t k, const int32_t** ids, const float** dists,
                  const int32_t** cnt);
    float pair_distance(size_t p1, size_t p2);
    // RangeQuery on the brute-force index: matches in insertion order, the first `capacity`; returns how many were written
    size_t range_host(const void* query, size_t elem_count, double radius, size_t capacity, int32_t* ids, float* dists);
    bool is_brute() const { return method_ == Method::Brute; }

    void save(const std::string& path, bool save_data);
    static std::unique_ptr<Engine> load(const std::string& path, int data_type, int dist_type, bool load_data);

    size_t thread_pool_size = 0;
    size_t memory_usage() const;

    // counters of the last HNSW batch (device pointers)
    const int32_t* last_ndc() const { return have_counters_ ? ws_ndc_.as<int32_t>() : nullptr; }
    const int32_t* last_hops() const { return have_counters_ ? ws_hops_.as<int32_t>() : nullptr; }
    const int32_t* last_hops_up() const { return have_counters_ ? ws_hops_up_.as<int32_t>() : nullptr; }

    // HIP-event timing of the dominant kernel (nmslib_gpu_kernel_timing)
    void set_profiling(bool on) { prof_ = on; }
    void collect_profile(double* total_ms, uint64_t* launches);

    double upload_seconds = 0, build_seconds = 0;
    int last_path = 0;  // see nmslib_gpu_stats_t
    // flags of the last fast-path slice (device): [fast_nqt_] fallback flags, then (float rows) [fast_nqt_] precise flags
    const int* fast_flags_ = nullptr;
    int fast_nqt_ = 0;
    bool fast_has_precise_ = false;
    void fast_tile_counts(size_t* tiles, size_t* precise, size_t* fallback);
    // HNSW (LDS visited table): queries of the last batch that were re-run by the bitset kernel (waits for the batch)
    size_t hnsw_redone();
    hipStream_t last_stream_ = nullptr;  // stream of the most recent batch (the statistics wait for THAT one)
    bool hnsw_fix_valid_ = false;
    size_t hbm_bytes() const;

    std::mutex mu;  // serialises finalize + queries on one index

    // ---- row shards behind one handle (index parameter gpu_shards) ----
    size_t shard_count() const { return shards_.size(); }

   private:
    // host rows / ids of this engine: its own, or (shard child) a window of the parent's
    const float* rows_f32() const { return parent_ ? parent_->rows_f32_.data() + view_lo_ * dim_ : rows_f32_.data(); }
    const uint8_t* rows_u8() const { return parent_ ? parent_->rows_u8_.data() + view_lo_ * 128 : rows_u8_.data(); }
    int resolve_shards() const;
    void finalize_sharded(int nshards);
    void knn_sharded(const void* d_queries, size_t nq, size_t elem_count, size_t k, int32_t* d_ids, float* d_dists,
                     int32_t* d_cnt, hipStream_t stream);
    std::vector<std::unique_ptr<Engine>> shards_;
    std::vector<hipEvent_t> shard_events_;
    hipEvent_t shard_ready_ = nullptr;
    const Engine* parent_ = nullptr;  // shard child: rows [view_lo_, view_lo_ + view_n_) of the parent, no copy
    size_t view_lo_ = 0, view_n_ = 0;
    int forced_device_ = -1;
    int gpu_shards_ = -1;  // index parameter: -1 unset, 0 = all visible devices, N = that many
    DevBuf ws_sh_ids_, ws_sh_d_, ws_bigk_, ws_bigk_tmp_;
    void* pinned(size_t bytes);
    void* pinned_ = nullptr;
    size_t pinned_bytes_ = 0;
    void check_device();
    void ensure_graph();
    void upload_rows();
    void build_graph();
    bool use_gpu_build() const;
    void build_graph_gpu();
    void prepare_graph_rows();
    void upload_graph();
    void knn_brute(const void* d_queries, size_t nq, size_t k, int32_t* d_ids, float* d_dists,
                   int32_t* d_cnt, hipStream_t stream);
    void knn_hnsw(const void* d_queries, size_t nq, size_t k, int32_t* d_ids, float* d_dists,
                  int32_t* d_cnt, hipStream_t stream);
    void knn_hnsw_old(const void* d_queries, size_t nq, size_t k, int32_t* d_ids, float* d_dists,
                      int32_t* d_cnt, hipStream_t stream);

    std::string space_name_, method_name_;
    int space_ = SP_L2;
    Method method_ = Method::Brute;
    size_t dim_ = 0;
    std::vector<int32_t> ids_;
    std::vector<float> rows_f32_;
    std::vector<uint8_t> rows_u8_;

    // index-time state
    bool created_ = false;       // nmslib_create_index was called
    bool dirty_ = true;          // rows added since the last finalize (device copy stale)
    bool graph_dirty_ = true;    // rows added since the graph was last built
    bool loaded_graph_ = false;  // graph came from a file: never rebuild it
    HnswBuildParams bp_;
    HostGraph graph_;
    std::vector<float> graph_rows_;  // rows as stored inside a loaded index (cosine: normalised)
    int ef_ = 200;                   // the shim's default (nmslib_c.cpp:330)
    std::string algo_ = "hybrid";

    // device state
    int device_ = -1;
    hipStream_t stream_ = nullptr;
    DevBuf d_rows_, d_rows_i8_, d_aux_, d_ids_, d_links0_, d_up_off_, d_up_links_, d_rownorm_;
    DevBuf d_auxh_;  // uint8 brute force: aux >> 1 (fast-path scan)
    DevBuf d_f16_hi_, d_auxp16_;   // fp16 tiles of the selection rows times f16_scale_ and their start values (one-product scan)
    DevBuf d_bf_hi_, d_bf_lo_, d_auxp_, ws_f32_q_;  // f32 fast path: bf16 hi / lo tiles of the selection rows, padded aux, split queries
    bool have_bf16_ = false;
    float f16_scale_q_ = 1.f;              // ... and of the batch's fp16 queries (the rows' scale, except centred cosine)
    float f16_scale_ = 1.f, bres16_ = 0;   // one-product scan: power-of-two scale of its fp16 tiles, the rows' largest fp16 residual
    void measure_rows_f16(const float* rows, size_t n, int ld, int dim, bool relative, float* bm);
    float bmax_ = 0;  // largest norm of the selection rows
    float bres_ = 0;  // their largest bf16 rounding residual (relative to the norm for the cosine spaces)
    DevBuf ws_u8_cand_, ws_u8_cnt_, ws_u8_thr_, ws_u8_list_, ws_u8_listcnt_;
    DevBuf ws_flags_;   // verified l2 path: query tiles whose proof failed (exact tail)
    DevBuf d_rows_sel_, d_mean_;  // brute-force L2 on un-centred data: selection copy (rows - column mean) and the mean
    bool centred_ = false;
    double mu_norm_ = 0;  // |column mean| (centred cosine scoring)
    double cosc_spread2_ = 0;          // E|b - mean|^2 measured at finalize
    float cosc_lambda_ = 1.f;          // centred cosine on the fast path: scale of the two constant columns (row_aug_cosc_kernel)
    float bmax_c_ = 0, bres_c_ = 0;    // largest norm / bf16 rounding residual of the augmented rows
    DevBuf ws_qaug_;                   // augmented queries of a batch
    size_t d_n_ = 0;
    int ldb_ = 0;
    HnswDeviceGraph dg_{};

    // workspaces
    DevBuf ws_q_, ws_qpad_, ws_qsel_, ws_qaux_, ws_cand_, ws_cnt_, ws_ids_, ws_dists_, ws_outcnt_, ws_status_, ws_bitset_;
    DevBuf ws_ndc_, ws_hops_, ws_hops_up_, ws_pair_, ws_rdist_, ws_rcnt_, ws_old_a_, ws_old_r_, ws_old_heap_;
    DevBuf wb_pts_, wb_src_, wb_starts_, wb_cand_ids_, wb_cand_d_, wb_cand_n_, wb_status_, wb_req_key_, wb_req_dist_,
        wb_req_key2_, wb_req_dist2_, wb_sort_tmp_, wb_active_, wb_nactive_, wb_extra_ids_, wb_extra_d_, wb_extra_n_;  // construction workspaces (released after the build)
    DevBuf ws_fix_;       // visited-overflow list of the HNSW search (count + query ids), device only
    size_t ctr_off_ = 0;  // offset of the current slice inside the per-batch counter arrays
    bool have_counters_ = false;
    bool prof_ = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events_;
    void prof_begin(hipStream_t s);
    void prof_end(hipStream_t s);
};

}  // namespace gfxknn
