// Engine: host logic around the gfx950 kernels (see engine.hpp).
#include "engine.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <fstream>
#include <set>
#include <sstream>
#include <thread>

namespace gfxknn {

using clk = std::chrono::steady_clock;

void hip_check(hipError_t e, const char* what) {
    if (e == hipSuccess) return;
    std::string msg = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();
    throw EngineError(e == hipErrorOutOfMemory ? Err::OutOfMemory : Err::Runtime, msg);
}

// ---------------------------------------------------------------------------------------------
// ParamSet
// ---------------------------------------------------------------------------------------------
ParamSet::ParamSet(const std::vector<std::string>& desc) {
    std::set<std::string> seen;
    for (const std::string& d : desc) {
        // AnyParams(const vector<string>&), params.h:50-74: exactly one '=' separating two parts
        size_t eq = d.find('=');
        if (eq == std::string::npos || d.find('=', eq + 1) != std::string::npos || eq == 0 || eq + 1 == d.size())
            throw std::runtime_error("Wrong format of an argument: '" + d + "' should be in the format: <Name>=<Value>");
        std::string name = d.substr(0, eq), val = d.substr(eq + 1);
        if (seen.count(name)) throw std::runtime_error("Duplicate parameter: " + name);
        seen.insert(name);
        kv_.emplace_back(name, val);
    }
    seen_.assign(kv_.size(), false);
}
bool ParamSet::has(const std::string& name) const {
    for (auto& p : kv_)
        if (p.first == name) return true;
    return false;
}
const std::string* ParamSet::find(const std::string& name) {
    for (size_t i = 0; i < kv_.size(); ++i)
        if (kv_[i].first == name) {
            seen_[i] = true;
            return &kv_[i].second;
        }
    return nullptr;
}
template <typename T>
static void convert(const std::string& s, T& v) {
    std::stringstream str(s);
    if (!(str >> v) || !str.eof()) throw std::runtime_error("Failed to convert value '" + s + "'");
}
void ParamSet::get(const std::string& name, long long& v) {
    if (auto s = find(name)) convert(*s, v);
}
void ParamSet::get(const std::string& name, int& v) {
    if (auto s = find(name)) convert(*s, v);
}
void ParamSet::get(const std::string& name, size_t& v) {
    if (auto s = find(name)) convert(*s, v);
}
void ParamSet::get(const std::string& name, double& v) {
    if (auto s = find(name)) convert(*s, v);
}
void ParamSet::get(const std::string& name, bool& v) {
    if (auto s = find(name)) convert(*s, v);
}
void ParamSet::get(const std::string& name, std::string& v) {
    if (auto s = find(name)) v = *s;
}
void ParamSet::check_unused() const {
    for (size_t i = 0; i < kv_.size(); ++i)
        if (!seen_[i]) throw std::runtime_error("Unknown parameters found!");
}

// ---------------------------------------------------------------------------------------------
// DevBuf
// ---------------------------------------------------------------------------------------------
void* DevBuf::ensure(size_t bytes) {
    if (bytes <= n_ && p_) return p_;
    release();
    if (bytes == 0) bytes = 16;
    hip_check(hipMalloc(&p_, bytes), "hipMalloc");
    n_ = bytes;
    // NMSLIB_GPU_POISON=<byte>: fill every new workspace with that byte, so that a kernel reading memory nobody wrote
    // shows up in the tests instead of depending on what the allocator hands back
    static const char* poison = getenv("NMSLIB_GPU_POISON");
    if (poison) hip_check(hipMemset(p_, atoi(poison) & 0xFF, bytes), "poison");
    return p_;
}
void DevBuf::release() {
    if (p_) (void)hipFree(p_);
    p_ = nullptr;
    n_ = 0;
}

// ---------------------------------------------------------------------------------------------
// Engine: construction / data
// ---------------------------------------------------------------------------------------------
static int space_from_name(const std::string& s) {
    // registered dense names: include/factory/init_spaces.h:77-86,120
    if (s == "l2") return SP_L2;
    if (s == "l1") return SP_L1;
    if (s == "linf") return SP_LINF;
    if (s == "cosinesimil") return SP_COSINE;
    if (s == "angulardist") return SP_ANGULAR;
    if (s == "negdotprod") return SP_NEGDOT;
    if (s == "l2sqr_sift") return SP_L2SQR_SIFT;
    return -1;
}

Engine::Engine(const std::string& space, const std::string& method, int data_type, int dist_type)
    : space_name_(space), method_name_(method) {
    (void)dist_type;
    const int sp = space_from_name(space);
    if (sp < 0)
        throw EngineError(Err::SpaceIncompatible,
                          "space '" + space + "' is not served by the GPU engine (dense spaces: l2, l1, linf, "
                          "cosinesimil, angulardist, negdotprod, l2sqr_sift)");
    space_ = sp;
    // data_type: 0 dense float, 2 dense uint8 (nmslib_c.h:12-17)
    if (data_type != 0 && data_type != 2)
        throw EngineError(Err::SpaceIncompatible, "only dense float and dense uint8 data are served by the GPU engine");
    if ((sp == SP_L2SQR_SIFT) != (data_type == 2))
        throw EngineError(Err::SpaceIncompatible, "data type does not match space '" + space + "'");
    // The method name is resolved at nmslib_create_index, like the reference
    // (MethodFactoryRegistry::CreateMethod, nmslib_c.cpp:493-496).
    thread_pool_size = std::thread::hardware_concurrency();
}

Engine::~Engine() {
    for (auto& pr : prof_events_) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    for (hipEvent_t e : shard_events_) (void)hipEventDestroy(e);
    if (shard_ready_) (void)hipEventDestroy(shard_ready_);
    if (pinned_) (void)hipHostFree(pinned_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

void Engine::prof_begin(hipStream_t s) {
    if (!prof_ || prof_events_.size() >= 65536) return;
    hipEvent_t a, b;
    hip_check(hipEventCreate(&a), "hipEventCreate");
    hip_check(hipEventCreate(&b), "hipEventCreate");
    prof_events_.emplace_back(a, b);
    hip_check(hipEventRecord(a, s), "hipEventRecord");
}
void Engine::prof_end(hipStream_t s) {
    if (!prof_ || prof_events_.empty()) return;
    hip_check(hipEventRecord(prof_events_.back().second, s), "hipEventRecord");
}
void Engine::collect_profile(double* total_ms, uint64_t* launches) {
    double tot = 0;
    uint64_t n = 0;
    for (auto& pr : prof_events_) {
        float ms = 0;
        if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
            tot += ms;
            ++n;
        }
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    prof_events_.clear();
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
}

void Engine::add_row(const void* data, size_t elem_count, int32_t id) {
    if (is_u8()) {
        // CreateObjFromUint8Vect CHECKs size == SIFT_DIM (space_l2sqr_sift.cc:136-140)
        if (elem_count != 128) throw EngineError(Err::Runtime, "SIFT vectors must have 128 bytes");
        if (dim_ == 0) dim_ = 128;
        const uint8_t* p = static_cast<const uint8_t*>(data);
        rows_u8_.insert(rows_u8_.end(), p, p + 128);
    } else {
        if (dim_ == 0) dim_ = elem_count;
        if (elem_count != dim_)
            throw EngineError(Err::InvalidArgument, "all rows of a dense index must have the same dimension");
        const float* p = static_cast<const float*>(data);
        rows_f32_.insert(rows_f32_.end(), p, p + elem_count);
    }
    ids_.push_back(id);
    dirty_ = true;
    graph_dirty_ = true;
}

const void* Engine::host_row(size_t pos) const {
    return is_u8() ? static_cast<const void*>(&rows_u8_[pos * 128]) : static_cast<const void*>(&rows_f32_[pos * dim_]);
}

void Engine::stored_row(size_t pos, void* dst) const {
    if (is_u8()) {
        // payload = 128 bytes + int32 sum of squares (space_l2sqr_sift.cc:146-149)
        std::memcpy(dst, &rows_u8_[pos * 128], 128);
        int32_t s = 0;
        for (int i = 0; i < 128; ++i) s += (int32_t)rows_u8_[pos * 128 + i] * (int32_t)rows_u8_[pos * 128 + i];
        std::memcpy(static_cast<char*>(dst) + 128, &s, 4);
    } else {
        std::memcpy(dst, &rows_f32_[pos * dim_], dim_ * 4);
    }
}

void Engine::reset() {
    ids_.clear();
    rows_f32_.clear();
    rows_u8_.clear();
    graph_ = HostGraph();
    graph_rows_.clear();
    loaded_graph_ = false;
    created_ = false;
    dirty_ = true;
    graph_dirty_ = true;
    d_n_ = 0;
    shards_.clear();
    dim_ = 0;  // a reset index accepts rows of another dimension
    centred_ = false;
    have_bf16_ = false;
    last_path = 0;
    fast_flags_ = nullptr;
    fast_nqt_ = 0;
    hnsw_fix_valid_ = false;
    have_counters_ = false;
}

size_t Engine::memory_usage() const {
    // nmslib_index_memory_usage (nmslib_c.cpp:1546-1565): object buffers + n*dim*4
    if (!created_) return 0;
    size_t total = ids_.size() * (16 + stored_row_bytes());
    total += ids_.size() * dim_ * sizeof(float);
    return total;
}

size_t Engine::hbm_bytes() const {
    size_t sh = 0;
    for (const auto& c : shards_) sh += c->hbm_bytes();
    return sh + d_rows_.bytes() + d_rows_i8_.bytes() + d_aux_.bytes() + d_ids_.bytes() + d_links0_.bytes() + d_up_off_.bytes() +
           d_up_links_.bytes() + d_rownorm_.bytes() + d_rows_sel_.bytes() + d_auxh_.bytes() + d_bf_hi_.bytes() + d_bf_lo_.bytes() + d_auxp_.bytes();
}

// ---------------------------------------------------------------------------------------------
// Index / query parameters
// ---------------------------------------------------------------------------------------------
void Engine::create_index(const std::vector<std::string>& params) {
    ParamSet ps(params);
    int defer = 0;
    if (method_name_ == "hnsw") {
        method_ = Method::Hnsw;
        HnswBuildParams bp;
        // Hnsw::CreateIndex, hnsw.cc:187-208
        ps.get("M", bp.M);
        int search_method = 0;
        ps.get("searchMethod", search_method);
        ps.get("indexThreadQty", bp.threads);
        ps.get("efConstruction", bp.efConstruction);
        bp.maxM = bp.M;
        ps.get("maxM", bp.maxM);
        bp.maxM0 = bp.M * 2;
        ps.get("maxM0", bp.maxM0);
        ps.get("mult", bp.mult);
        ps.get("delaunay_type", bp.delaunay);
        ps.get("post", bp.post);
        int skip = 0;
        ps.get("skip_optimized_index", skip);
        bp.skip_optimized = skip != 0;
        ps.get("gpu_defer", defer);  // engine extension: build now, upload to HBM at first use
        ps.get("gpu_build", bp.gpu_build);  // engine extension: batched construction on the GPU (1), host (0)
        ps.get("gpu_build_batch", bp.gpu_build_batch);
        ps.get("gpu_build_div", bp.gpu_build_div);
        ps.get("gpu_shards", gpu_shards_);  // engine extension: row shards over the visible GPUs (0 = all)
        ps.check_unused();
        if (defer && bp.gpu_build < 0) bp.gpu_build = 0;  // deferred upload = no device work now
        bp_ = bp;
        ef_ = 200;  // the shim forces efSearch=200 per query (nmslib_c.cpp:330,986)
        algo_ = "hybrid";
    } else if (method_name_ == "brute_force" || method_name_ == "seq_search") {
        method_ = Method::Brute;
        bool copy_mem = false, multi = false;
        size_t thread_qty = 0;
        ps.get("copyMem", copy_mem);  // SeqSearch::CreateIndex, seqsearch.cc:63-66
        ps.get("multiThread", multi);
        ps.get("threadQty", thread_qty);
        ps.get("gpu_defer", defer);
        ps.get("gpu_shards", gpu_shards_);
        ps.check_unused();
    } else {
        throw EngineError(Err::IndexBuildFailed,
                          "method '" + method_name_ + "' is not served by the GPU engine (hnsw, brute_force, seq_search)");
    }
    created_ = true;
    dirty_ = true;
    graph_dirty_ = true;
    loaded_graph_ = false;
    if (!ids_.empty()) {
        if (defer) ensure_graph();
        else finalize();
    }
}

void Engine::set_query_params(const std::vector<std::string>& params) {
    ParamSet ps(params);
    if (method_ == Method::Hnsw) {
        // Hnsw::SetQueryTimeParams, hnsw.cc:472-507
        if (ps.has("ef") && ps.has("efSearch"))
            throw std::runtime_error("The user shouldn't specify parameters ef and efSearch at the same time");
        size_t ef = 20;
        ps.get("ef", ef);
        ps.get("efSearch", ef);
        int tmp = 0;
        ps.get("searchMethod", tmp);
        std::string algo = "hybrid";
        ps.get("algoType", algo);
        std::transform(algo.begin(), algo.end(), algo.begin(), ::tolower);
        if (algo != "v1merge" && algo != "old" && algo != "hybrid")
            throw std::runtime_error("algoType should be one of the following: old, v1merge");
        ps.check_unused();
        if (ef < 1) throw std::runtime_error("ef must be positive");
        ef_ = (int)ef;
        algo_ = algo;
    } else {
        ps.check_unused();  // SeqSearch has no query-time parameters
    }
}

// ---------------------------------------------------------------------------------------------
// Device residency
// ---------------------------------------------------------------------------------------------
void Engine::check_device() {
    if (device_ >= 0) {
        hip_check(hipSetDevice(device_), "hipSetDevice");
        return;
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        (void)hipGetLastError();
        throw EngineError(Err::Runtime,
                          "no HIP device available: the k-NN path of this library runs only on the GPU "
                          "(there is no CPU fallback)");
    }
    int dev = 0;
    if (forced_device_ >= 0) dev = forced_device_ % count;
    else if (const char* env = getenv("NMSLIB_GPU_DEVICE")) dev = atoi(env);
    else if (const char* lr = getenv("LOCAL_RANK")) dev = atoi(lr) % count;
    hip_check(hipSetDevice(dev), "hipSetDevice");
    device_ = dev;
    hip_check(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate");
}

void Engine::upload_rows() {
    const size_t n = size();
    auto t0 = clk::now();
    if (is_u8()) {
        ldb_ = 128;
        d_rows_.ensure(std::max<size_t>(n, 1) * 128);
        if (n) hip_check(hipMemcpy(d_rows_.ptr(), rows_u8(), n * 128, hipMemcpyHostToDevice), "upload rows");
    } else {
        ldb_ = f32_row_stride((int)dim_);
        d_rows_.ensure(std::max<size_t>(n, 1) * ldb_ * 4);
        const float* src = rows_f32();
        if (loaded_graph_ && !graph_rows_.empty()) src = graph_rows_.data();
        if (n) {
            hip_check(hipMemset(d_rows_.ptr(), 0, n * ldb_ * 4), "clear rows");
            hip_check(hipMemcpy2D(d_rows_.ptr(), (size_t)ldb_ * 4, src, dim_ * 4, dim_ * 4, n, hipMemcpyHostToDevice),
                      "upload rows");
        }
    }
    d_ids_.ensure(std::max<size_t>(n, 1) * 4);
    if (parent_) {
        // shard child: results carry GLOBAL positions; the parent maps them to external ids after the merge, so
        // that ties are ordered exactly as in the unsharded index (distance, position)
        std::vector<int32_t> pos(n);
        for (size_t i = 0; i < n; ++i) pos[i] = (int32_t)(view_lo_ + i);
        if (n) hip_check(hipMemcpy(d_ids_.ptr(), pos.data(), n * 4, hipMemcpyHostToDevice), "upload positions");
    } else if (n) {
        hip_check(hipMemcpy(d_ids_.ptr(), ids_.data(), n * 4, hipMemcpyHostToDevice), "upload ids");
    }
    d_n_ = n;
    upload_seconds = std::chrono::duration<double>(clk::now() - t0).count();
}

void Engine::build_graph() {
    auto t0 = clk::now();
    if (!loaded_graph_) {
        const void* rows = is_u8() ? static_cast<const void*>(rows_u8()) : static_cast<const void*>(rows_f32());
        hnsw_build_host(space_, rows, size(), dim_, bp_, graph_);
    }
    build_seconds = std::chrono::duration<double>(clk::now() - t0).count();
}

void Engine::prepare_graph_rows() {
    // search-time distance of the flat ("optimized") index, hnsw.cc:369-412
    const size_t n = size();
    int sspace = space_;
    bool normalize = false;
    const bool optimized = !bp_.skip_optimized;
    if (optimized) {
        if (space_ == SP_L2) sspace = SP_L2SQR;
        else if (space_ == SP_COSINE) {
            sspace = SP_NORMCOS;
            normalize = true;
        }
    }
    if (normalize && !(loaded_graph_ && !graph_rows_.empty()))
        hip_check(launch_normalize_rows(d_rows_.as<float>(), (int)n, ldb_, (int)dim_, stream_), "normalize rows");
    if (is_u8()) {
        d_rownorm_.ensure(std::max<size_t>(n, 1) * 4);
        std::vector<int32_t> norms(n);
        for (size_t i = 0; i < n; ++i) {
            int32_t s = 0;
            const uint8_t* r8 = rows_u8() + i * 128;
            for (int t = 0; t < 128; ++t) s += (int32_t)r8[t] * (int32_t)r8[t];
            norms[i] = s;
        }
        if (n) hip_check(hipMemcpy(d_rownorm_.ptr(), norms.data(), n * 4, hipMemcpyHostToDevice), "row norms");
    }
    dg_ = HnswDeviceGraph{};
    dg_.rows = d_rows_.ptr();
    dg_.row_norm = d_rownorm_.as<int32_t>();
    dg_.ext_ids = d_ids_.as<int32_t>();
    dg_.n = (int)n;
    dg_.dim = (int)dim_;
    dg_.ldv = is_u8() ? 128 : ldb_;
    dg_.space = sspace;
    dg_.normalize_query = normalize ? 1 : 0;
}

void Engine::upload_graph() {
    const HostGraph& g = graph_;
    const size_t n = (size_t)g.n;
    d_links0_.ensure(std::max<size_t>(g.links0.size(), 1) * 4);
    d_up_off_.ensure(std::max<size_t>(n, 1) * 8);
    d_up_links_.ensure(std::max<size_t>(g.up_links.size(), 1) * 4);
    if (n) {
        hip_check(hipMemcpy(d_links0_.ptr(), g.links0.data(), g.links0.size() * 4, hipMemcpyHostToDevice), "links0");
        hip_check(hipMemcpy(d_up_off_.ptr(), g.up_off.data(), n * 8, hipMemcpyHostToDevice), "up_off");
        if (!g.up_links.empty())
            hip_check(hipMemcpy(d_up_links_.ptr(), g.up_links.data(), g.up_links.size() * 4, hipMemcpyHostToDevice),
                      "up_links");
    }
    prepare_graph_rows();
    dg_.links0 = d_links0_.as<int32_t>();
    dg_.up_off = d_up_off_.as<int64_t>();
    dg_.up_links = d_up_links_.as<int32_t>();
    dg_.maxM = g.maxM;
    dg_.maxM0 = g.maxM0;
    dg_.maxlevel = g.maxlevel;
    dg_.enterpoint = g.enterpoint;
    hip_check(hipStreamSynchronize(stream_), "graph upload");
}

bool Engine::use_gpu_build() const {
    if (loaded_graph_ || method_ != Method::Hnsw) return false;
    // what only the host builder does (the reference's "no limits"): lists beyond two words per lane, selection
    // heuristics 1 and 3, post-processing -- also when gpu_build=1 asks for the GPU
    if (bp_.maxM > 62 || bp_.maxM0 > 126 || bp_.M > 62 || bp_.delaunay == 1 || bp_.delaunay == 3 || bp_.post != 0) return false;
    if (bp_.gpu_build >= 0) return bp_.gpu_build != 0;
    // auto: indexThreadQty=1 asks for the reference's sequential insertion order (bit-identical graph,
    // host builder); anything else is the concurrent build, whose schedule is free -> the GPU, unless the
    // parameters exceed what its kernels hold in LDS (the host builder has the reference's "no limits")
    if (bp_.efConstruction > 1024 || bp_.maxM > 62 || bp_.maxM0 > 126) return false;
    return bp_.threads != 1;
}

// Batched construction on the GPU (kernels/hnsw_build_kernels.hip).  Insertion order, level stream and
// per-node algorithm are the reference's (Hnsw::add, hnsw.cc:534-609); what differs is visibility:
// the nodes of one batch are searched against the graph as it was before the batch, then linked.
// Batches grow with the graph (size/16, at most gpu_build_batch), a node that raises the top level
// closes its batch.  All searches of a batch (every level, top down) precede all link updates.
void Engine::build_graph_gpu() {
    auto t0 = clk::now();
    const size_t n = size();
    hnsw_check_params(bp_);
    const int M = bp_.M, maxM = bp_.maxM, maxM0 = bp_.maxM0, efC = bp_.efConstruction;
    if (efC < 1 || efC > 1024)
        throw EngineError(Err::IndexBuildFailed, "HNSW: efConstruction must be in [1, 1024] for the GPU builder");
    HostGraph& g = graph_;
    g = HostGraph{};
    g.n = (int)n;
    g.M = M;
    g.maxM = maxM;
    g.maxM0 = maxM0;
    g.efConstruction = efC;
    g.delaunay = bp_.delaunay;
    g.levels = hnsw_random_levels(n, bp_);
    g.up_off.assign(n, -1);
    size_t up_ints = 0;
    for (size_t i = 0; i < n; ++i)
        if (g.levels[i] > 0) {
            g.up_off[i] = (int64_t)up_ints;
            up_ints += (size_t)g.levels[i] * (maxM + 1);
        }
    const size_t l0_ints = n * (size_t)(maxM0 + 1);
    d_links0_.ensure(std::max<size_t>(l0_ints, 1) * 4);
    d_up_off_.ensure(std::max<size_t>(n, 1) * 8);
    d_up_links_.ensure(std::max<size_t>(up_ints, 1) * 4);
    prepare_graph_rows();
    dg_.links0 = d_links0_.as<int32_t>();
    dg_.up_off = d_up_off_.as<int64_t>();
    dg_.up_links = d_up_links_.as<int32_t>();
    dg_.maxM = maxM;
    dg_.maxM0 = maxM0;
    if (n == 0) {
        build_seconds = 0;
        return;
    }
    hipStream_t s = stream_;
    hip_check(hipMemsetAsync(d_links0_.ptr(), 0, l0_ints * 4, s), "clear links0");
    hip_check(hipMemsetAsync(d_up_links_.ptr(), 0, std::max<size_t>(up_ints, 1) * 4, s), "clear up_links");
    hip_check(hipMemcpyAsync(d_up_off_.ptr(), g.up_off.data(), n * 8, hipMemcpyHostToDevice, s), "up_off");

    HnswDeviceGraph bgraph = dg_;
    bgraph.ext_ids = nullptr;  // construction works on internal positions
    int maxlevel = g.levels[0], enterpoint = 0;
    const int max_batch = bp_.gpu_build_batch > 0 ? bp_.gpu_build_batch : 4096;
    const int batch_div = bp_.gpu_build_div > 0 ? bp_.gpu_build_div : 16;
    // reverse-link requests of one level of one batch: M slots per new node, sorted on the device
    const size_t req_max = (size_t)max_batch * (size_t)M;
    wb_req_key_.ensure(req_max * 8);
    wb_req_dist_.ensure(req_max * 4);
    wb_req_key2_.ensure(req_max * 8);
    wb_req_dist2_.ensure(req_max * 4);
    const size_t sort_tmp = hnsw_build_sort_temp_bytes((int)req_max, (int)n);
    wb_sort_tmp_.ensure(sort_tmp);
    wb_active_.ensure(req_max * 4);
    wb_nactive_.ensure(64);
    wb_extra_ids_.ensure((size_t)max_batch * 64 * 4);  // batch-mates: 64 per new node of a level slice
    wb_extra_d_.ensure((size_t)max_batch * 64 * 4);
    wb_extra_n_.ensure((size_t)max_batch * 4);

    std::vector<int32_t> pts, src, status;
    std::vector<size_t> base;
    size_t next = 1;
    while (next < n) {
        const size_t sz = next;
        size_t bsz = std::min<size_t>(std::max<size_t>(sz / (size_t)batch_div, 1), (size_t)max_batch);
        size_t end = std::min(n, next + bsz);
        for (size_t i = next; i < end; ++i)
            if (g.levels[i] > maxlevel) {
                end = i + 1;
                break;
            }
        // (node, level) pairs of the batch, level-major (top level first); src = the same node's pair one level up
        const int top = maxlevel;
        pts.clear();
        src.clear();
        base.assign((size_t)top + 2, 0);
        std::vector<int32_t> prev_slot(end - next, -1), cur_slot(end - next, -1);
        for (int l = top; l >= 0; --l) {
            base[l] = pts.size();
            for (size_t i = next; i < end; ++i) {
                if (std::min(g.levels[i], top) < l) continue;
                cur_slot[i - next] = (int32_t)pts.size();
                pts.push_back((int32_t)i);
                src.push_back(prev_slot[i - next]);
            }
            prev_slot = cur_slot;
            std::fill(cur_slot.begin(), cur_slot.end(), -1);
        }
        // base[l] .. base[l-1] (or npairs for l == 0) is level l's slice
        const size_t npairs = pts.size();
        auto slice_end = [&](int l) { return l == 0 ? npairs : base[l - 1]; };
        wb_pts_.ensure(npairs * 4);
        wb_src_.ensure(npairs * 4);
        wb_starts_.ensure(npairs * 4);
        wb_cand_ids_.ensure(npairs * (size_t)efC * 4);
        wb_cand_d_.ensure(npairs * (size_t)efC * 4);
        wb_cand_n_.ensure(npairs * 4);
        wb_status_.ensure(npairs * 4);
        ws_ndc_.ensure(npairs * 4);
        ws_hops_.ensure(npairs * 4);
        ws_hops_up_.ensure(npairs * 4);
        hip_check(hipMemcpyAsync(wb_pts_.ptr(), pts.data(), npairs * 4, hipMemcpyHostToDevice, s), "batch nodes");
        hip_check(hipMemcpyAsync(wb_src_.ptr(), src.data(), npairs * 4, hipMemcpyHostToDevice, s), "batch sources");
        bgraph.maxlevel = maxlevel;
        bgraph.enterpoint = enterpoint;

        for (int attempt = 0; attempt < 2; ++attempt) {
            const bool force_bitset = attempt == 1;
            bool used_table = false;
            for (int l = top; l >= 0; --l) {
                const size_t b0 = base[l], m = slice_end(l) - b0;
                if (m == 0) continue;
                hip_check(launch_hnsw_build_starts(wb_src_.as<int32_t>() + b0, wb_cand_ids_.as<int32_t>(),
                                                   wb_cand_n_.as<int32_t>(), efC, wb_starts_.as<int32_t>() + b0, (int)m, s),
                          "build starts");
                HnswSearchPlan p = hnsw_make_plan(bgraph, (int)m, efC, efC, force_bitset);
                uint32_t* bitset = nullptr;
                if (p.table_size == 0) {
                    ws_bitset_.ensure(m * p.bitset_words * 4);
                    hip_check(hipMemsetAsync(ws_bitset_.ptr(), 0, m * p.bitset_words * 4, s), "clear visited bitset");
                    bitset = ws_bitset_.as<uint32_t>();
                } else {
                    used_table = true;
                }
                hip_check(launch_hnsw_search_ex(bgraph, p, nullptr, wb_pts_.as<int32_t>() + b0,
                                                wb_starts_.as<int32_t>() + b0, l, bitset,
                                                wb_cand_ids_.as<int32_t>() + b0 * efC, wb_cand_d_.as<float>() + b0 * efC,
                                                wb_cand_n_.as<int32_t>() + b0, ws_ndc_.as<int32_t>() + b0,
                                                ws_hops_.as<int32_t>() + b0, ws_hops_up_.as<int32_t>() + b0,
                                                wb_status_.as<int32_t>() + b0, s),
                          "build search");
            }
            if (!used_table) break;
            status.resize(npairs);
            hip_check(hipMemcpyAsync(status.data(), wb_status_.ptr(), npairs * 4, hipMemcpyDeviceToHost, s), "status");
            hip_check(hipStreamSynchronize(s), "build search");
            bool any = false;
            for (int32_t v : status) any |= (v != 0);
            if (!any) break;  // else: the LDS visited table overflowed somewhere -> redo with HBM bitsets
        }

        HnswBuildGraph bg{};
        bg.g = bgraph;
        bg.links0 = d_links0_.as<int32_t>();
        bg.up_links = d_up_links_.as<int32_t>();
        bg.M = M;
        bg.delaunay = bp_.delaunay;
        for (int l = top; l >= 0; --l) {
            const size_t b0 = base[l], m = slice_end(l) - b0;
            if (m == 0) continue;
            const int total = (int)(m * (size_t)M);
            hip_check(launch_hnsw_build_mates(bg, l, wb_pts_.as<int32_t>() + b0, (int)m, wb_cand_d_.as<float>() + b0 * efC,
                                              wb_cand_n_.as<int32_t>() + b0, efC, wb_extra_ids_.as<int32_t>(),
                                              wb_extra_d_.as<float>(), wb_extra_n_.as<int32_t>(), s),
                      "build batch-mates");
            hip_check(launch_hnsw_build_select(bg, l, wb_pts_.as<int32_t>() + b0, (int)m,
                                               wb_cand_ids_.as<int32_t>() + b0 * efC, wb_cand_d_.as<float>() + b0 * efC,
                                               wb_cand_n_.as<int32_t>() + b0, efC, wb_extra_ids_.as<int32_t>(),
                                               wb_extra_d_.as<float>(), wb_extra_n_.as<int32_t>(),
                                               wb_req_key_.as<unsigned long long>(), wb_req_dist_.as<float>(), s),
                      "build select");
            hip_check(launch_hnsw_build_sort_requests(wb_req_key_.as<unsigned long long>(), wb_req_dist_.as<float>(), total,
                                                      wb_req_key2_.as<unsigned long long>(), wb_req_dist2_.as<float>(),
                                                      wb_sort_tmp_.ptr(), sort_tmp, wb_active_.as<int32_t>(),
                                                      wb_nactive_.as<int32_t>(), s),
                      "build sort");
            const size_t max_active = std::min<size_t>(sz, (size_t)total);
            hip_check(launch_hnsw_build_link(bg, l, wb_active_.as<int32_t>(), wb_nactive_.as<int32_t>(), (int)max_active,
                                             wb_req_key2_.as<unsigned long long>(), wb_req_dist2_.as<float>(), total, s),
                      "build link");
        }
        for (size_t i = next; i < end; ++i)
            if (g.levels[i] > maxlevel) {
                maxlevel = g.levels[i];
                enterpoint = (int)i;
            }
        // pts/src are reused by the next batch: the copies above must have been consumed
        hip_check(hipStreamSynchronize(s), "build batch");
        next = end;
    }
    g.maxlevel = maxlevel;
    g.enterpoint = enterpoint;
    g.links0.resize(l0_ints);
    g.up_links.resize(up_ints);
    hip_check(hipMemcpyAsync(g.links0.data(), d_links0_.ptr(), l0_ints * 4, hipMemcpyDeviceToHost, s), "links0 D2H");
    if (up_ints)
        hip_check(hipMemcpyAsync(g.up_links.data(), d_up_links_.ptr(), up_ints * 4, hipMemcpyDeviceToHost, s),
                  "up_links D2H");
    hip_check(hipStreamSynchronize(s), "graph download");
    dg_.maxlevel = maxlevel;
    dg_.enterpoint = enterpoint;
    for (DevBuf* b : {&wb_pts_, &wb_src_, &wb_starts_, &wb_cand_ids_, &wb_cand_d_, &wb_cand_n_, &wb_status_,
                      &wb_req_key_, &wb_req_dist_, &wb_req_key2_, &wb_req_dist2_, &wb_sort_tmp_, &wb_active_, &wb_nactive_, &wb_extra_ids_, &wb_extra_d_, &wb_extra_n_})
        b->release();
    build_seconds = std::chrono::duration<double>(clk::now() - t0).count();
}

void Engine::ensure_graph() {
    if (method_ != Method::Hnsw || !graph_dirty_) return;
    if (use_gpu_build() || (!parent_ && (shards_.size() > 1 || (dirty_ && resolve_shards() > 1)))) {
        finalize();
        return;
    }
    build_graph();
    graph_dirty_ = false;
}

void Engine::finalize() {
    if (!created_) throw EngineError(Err::IndexBuildFailed, "Index not built");
    if (!dirty_) return;
    if (!parent_) {
        const int nsh = resolve_shards();
        if (nsh > 1) {
            finalize_sharded(nsh);
            dirty_ = false;
            graph_dirty_ = false;
            return;
        }
        shards_.clear();
    }
    if (method_ == Method::Hnsw && graph_dirty_ && use_gpu_build()) {
        check_device();
        upload_rows();
        build_graph_gpu();
        graph_dirty_ = false;
        dirty_ = false;
        return;
    }
    ensure_graph();  // host-side construction first: it needs no device
    check_device();
    upload_rows();
    if (method_ == Method::Brute) {
        const size_t n = size();
        if (is_u8()) {
            const size_t n_pad = (size_t)bf_u8_rows_padded((int)n);
            d_aux_.ensure(n_pad * 4);
            d_rows_i8_.ensure(n_pad * 128);
            d_auxh_.ensure(n_pad * 4);
            hip_check(launch_prepare_u8(d_rows_.as<uint8_t>(), (int)n, d_rows_i8_.as<uint8_t>(), d_aux_.as<int32_t>(),
                                        d_auxh_.as<int32_t>(), stream_),
                      "prepare u8 rows");
        } else {
            d_aux_.ensure(std::max<size_t>(n, 1) * 4);
            const float* sel_rows = d_rows_.as<float>();
            centred_ = false;
            d_rows_sel_.release();
            if ((space_ == SP_L2 || space_ == SP_COSINE || space_ == SP_ANGULAR) && n > 0) {
                // L2 is translation invariant, the Q.B^T score q.b - |b|^2/2 is not: its f32 rounding error grows with
                // |q||b|, i.e. with a common offset of the data, and can exceed the gaps between neighbours (the
                // reference's direct sum (a-b)^2, distcomp_lp.cc:304-365, has no such term).  When the column mean
                // is not small against the spread, SELECTION runs on rows - mean and queries - mean; the exact
                // re-rank keeps using the original rows.  Cosine / angular: same centred copy, and the score is
                // rebuilt as 1 - cos = (|q'-b'|^2 - (|q|-|b|)^2) / (2|q||b|) (bf_kernels.hip, BF_COSC).
                std::vector<double> st((size_t)ldb_ + 1);
                DevBuf d_stats;
                d_stats.ensure(st.size() * 8);
                hip_check(launch_col_stats(d_rows_.as<float>(), (int)n, ldb_, (int)dim_, d_stats.as<double>(), stream_),
                          "column stats");
                hip_check(hipMemcpyAsync(st.data(), d_stats.ptr(), st.size() * 8, hipMemcpyDeviceToHost, stream_), "stats D2H");
                hip_check(hipStreamSynchronize(stream_), "column stats");
                double mu2 = 0;
                std::vector<float> mean((size_t)ldb_, 0.f);
                for (size_t c = 0; c < dim_; ++c) {
                    const double m = st[c] / (double)n;
                    mean[c] = (float)m;
                    mu2 += m * m;
                }
                const double spread2 = std::max(0.0, st[(size_t)ldb_] / (double)n - mu2);
                cosc_spread2_ = spread2;
                bool centre = mu2 > 0.0625 * spread2;
                if (const char* env = getenv("NMSLIB_GPU_CENTER")) centre = atoi(env) != 0;
                if (centre) {
                    d_mean_.ensure((size_t)ldb_ * 4);
                    d_rows_sel_.ensure(n * (size_t)ldb_ * 4);
                    hip_check(hipMemcpyAsync(d_mean_.ptr(), mean.data(), (size_t)ldb_ * 4, hipMemcpyHostToDevice, stream_), "mean");
                    hip_check(launch_center_rows(d_rows_.as<float>(), d_mean_.as<float>(), (int)n, (int)n, ldb_, (int)dim_,
                                                 d_rows_sel_.as<float>(), stream_),
                              "centre rows");
                    hip_check(hipStreamSynchronize(stream_), "centre rows");  // `mean` is read by the copy above
                    sel_rows = d_rows_sel_.as<float>();
                    centred_ = true;
                    mu_norm_ = 0;
                    for (size_t c = 0; c < dim_; ++c) mu_norm_ += (double)mean[c] * (double)mean[c];
                    mu_norm_ = std::sqrt(mu_norm_);
                }
            }
            if (centred_ && space_ != SP_L2) {
                d_aux_.ensure(std::max<size_t>(n, 1) * 12);
                hip_check(launch_row_aux_cosc(d_rows_.as<float>(), sel_rows, (int)n, ldb_, (int)dim_, mu_norm_,
                                              d_aux_.as<float>(), stream_),
                          "row aux");
            } else {
                hip_check(launch_row_aux_f32(sel_rows, (int)n, ldb_, (int)dim_, space_, d_aux_.as<float>(), stream_),
                          "row aux");
            }
            // largest norm (and bf16 rounding residual) of the selection rows: the error bounds of the selection scores
            // (proofs in bf_rerank_kernel and bf_rerank_f32_list_kernel)
            {
                const bool cosine = space_ == SP_COSINE || space_ == SP_ANGULAR;
                float bm[4] = {0.f, 0.f, 0.f, 0.f};
                measure_rows_f16(sel_rows, n, ldb_, (int)dim_, cosine, bm);
                bmax_ = bm[0];
                bres_ = bm[1];
            }
            // fast path (large batches, D <= 128): bf16 hi / lo tiles of the selection rows + padded aux
            have_bf16_ = false;
            d_bf_hi_.release();
            d_bf_lo_.release();
            d_f16_hi_.release();
            d_auxp_.release();
            d_auxp16_.release();
            const bool cosine_space = space_ == SP_COSINE || space_ == SP_ANGULAR;
            const bool fast_space = space_ == SP_L2 || space_ == SP_NEGDOT || (cosine_space && !centred_);
            if (cosine_space && centred_ && dim_ + 3 <= 1024 && n >= 65536) {
                // centred cosine / angular (round 3): the score -(1 - cos)|q| as an inner product of rows and queries with
                // three more columns (bf_kernels.hip, row_aug_cosc_kernel); bf16 tiles of those rows, scanned in the
                // inner-product mode.  A zero-norm row has no score of this form: the index then stays on the adaptive kernel.
                const BfF32Fast f0 = bf_f32_fast_plan((int)n, (int)dim_, 1024, 10, space_, true);
                if (f0.use) {
                    const size_t dp = (size_t)f0.dp;
                    const size_t n_pad = (size_t)bf_f32_rows_padded((int)n);
                    DevBuf d_aug, d_flag;
                    d_aug.ensure(n * dp * 4);
                    d_flag.ensure(16);
                    hip_check(hipMemsetAsync(d_flag.ptr(), 0, 16, stream_), "clear");
                    // (the two constant columns balanced at the typical (|b'|^2 - db^2) / 2 <= spread^2 / 2)
                    cosc_lambda_ = (float)std::sqrt(std::max(0.5 * cosc_spread2_, 1e-30));
                    hip_check(launch_row_aug_cosc(d_rows_.as<float>(), sel_rows, (int)n, ldb_, (int)dim_, mu_norm_, cosc_lambda_,
                                                  d_aug.as<float>(), (int)dp, d_flag.as<int>(), stream_),
                              "augmented rows");
                    int fl[1] = {0};
                    hip_check(hipMemcpyAsync(fl, d_flag.ptr(), 4, hipMemcpyDeviceToHost, stream_), "flags");
                    hip_check(hipStreamSynchronize(stream_), "augmented rows");
                    if (fl[0] == 0) {
                        float bm[4] = {0.f, 0.f, 0.f, 0.f};
                        measure_rows_f16(d_aug.as<float>(), n, (int)dp, (int)dim_ + 3, false, bm);   // (sets f16_scale_, bres16_)
                        // (the augmented rows are divided by their norm ~ |mean|, the augmented queries are not)
                        if (mu_norm_ > 0) {
                            const int eq = std::max(-40, std::min(40, (int)std::lround(std::log2((double)f16_scale_) - std::log2(mu_norm_))));
                            f16_scale_q_ = std::ldexp(1.f, eq);
                        }
                        bmax_c_ = bm[0];
                        bres_c_ = bm[1];
                        if (dp == 128) {
                            d_bf_hi_.ensure(n_pad * dp * 2);
                            d_bf_lo_.ensure(n_pad * 128 * 2);
                        }
                        d_f16_hi_.ensure(n_pad * dp * 2);
                        d_auxp_.ensure(n_pad * 4);
                        d_auxp16_.ensure(n_pad * 4);
                        hip_check(launch_split_bf16(d_aug.as<float>(), (int)n, (int)n_pad, (int)dp, (int)dim_ + 3,
                                                    dp == 128 ? d_bf_hi_.ptr() : nullptr, dp == 128 ? d_bf_lo_.ptr() : nullptr,
                                                    nullptr, 0.f, d_auxp_.as<float>(), stream_, (int)dp, d_f16_hi_.ptr(), f16_scale_,
                                                    d_auxp16_.as<float>(), 1.f),
                                  "split rows");
                        hip_check(hipStreamSynchronize(stream_), "split rows");   // (d_aug goes out of scope)
                        have_bf16_ = true;
                    }
                }
            }
            if (fast_space && dim_ <= 1024 && n >= 65536) {
                // (rows up to 128 dimensions: hi and lo tiles; longer rows, round 3: hi tiles of 128 * kch columns
                //  for the K-chunked one-product scan -- the plan's kch, which depends on the dimension only)
                const BfF32Fast f0 = bf_f32_fast_plan((int)n, (int)dim_, 1024, 10, space_, centred_);
                const size_t dp = f0.use ? (size_t)f0.dp : 128;
                const size_t n_pad = (size_t)bf_f32_rows_padded((int)n);
                // (bf16 hi / lo tiles: the split-product scan, rows of up to 128 dimensions only; fp16 tiles of the rows times
                //  f16_scale_: the one-product scan -- its start values in units of scale^2 for l2, the plain 1/|b| for cosine)
                if (dp == 128) {
                    d_bf_hi_.ensure(n_pad * dp * 2);
                    d_bf_lo_.ensure(n_pad * 128 * 2);
                }
                d_f16_hi_.ensure(n_pad * dp * 2);
                d_auxp_.ensure(n_pad * 4);
                d_auxp16_.ensure(n_pad * 4);
                const float pad = space_ == SP_L2 ? -INFINITY : 0.f;
                hip_check(launch_split_bf16(sel_rows, (int)n, (int)n_pad, ldb_, (int)dim_, dp == 128 ? d_bf_hi_.ptr() : nullptr,
                                            dp == 128 ? d_bf_lo_.ptr() : nullptr,
                                            space_ == SP_NEGDOT ? nullptr : d_aux_.as<float>(), pad, d_auxp_.as<float>(),
                                            stream_, (int)dp, d_f16_hi_.ptr(), f16_scale_, d_auxp16_.as<float>(),
                                            space_ == SP_L2 ? f16_scale_ * f16_scale_ : 1.f),
                          "split rows");
                have_bf16_ = true;
            }
        }
        hip_check(hipStreamSynchronize(stream_), "finalize");
    } else {
        upload_graph();
    }
    dirty_ = false;
}

// bm[0] largest row norm, bm[1] largest bf16 residual; chooses the fp16 scale of the one-product scan from the largest
// |element| (a power of two that puts it into [2^13, 2^14): a factor 4 of headroom below fp16's 65504 for the queries) and
// measures the rows' largest fp16 residual at that scale (f16_scale_, bres16_)
void Engine::measure_rows_f16(const float* rows, size_t n, int ld, int dim, bool relative, float* bm) {
    DevBuf d_bm;
    d_bm.ensure(16);
    hip_check(launch_row_maxnorm(rows, (int)n, ld, dim, relative, d_bm.as<float>(), stream_), "row norms");
    hip_check(hipMemcpyAsync(bm, d_bm.ptr(), 16, hipMemcpyDeviceToHost, stream_), "bmax");
    hip_check(hipStreamSynchronize(stream_), "row norms");
    f16_scale_ = 1.f;
    if (bm[2] > 0.f && std::isfinite(bm[2])) {
        // (|exponent| <= 40: scale^2, the unit of the scan's scores, must stay a float; rows smaller than 2^-27 simply lose
        //  fp16 precision, which the measured residual reports)
        const int e = std::max(-40, std::min(40, 13 - (int)std::floor(std::log2(bm[2]))));
        f16_scale_ = std::ldexp(1.f, e);
    }
    float bm2[4] = {0.f, 0.f, 0.f, 0.f};
    hip_check(launch_row_maxnorm(rows, (int)n, ld, dim, relative, d_bm.as<float>(), stream_, f16_scale_), "row residuals");
    hip_check(hipMemcpyAsync(bm2, d_bm.ptr(), 16, hipMemcpyDeviceToHost, stream_), "bres16");
    hip_check(hipStreamSynchronize(stream_), "row residuals");
    bres16_ = bm2[3];
    f16_scale_q_ = f16_scale_;
}

// restores the calling thread's current device on every way out of a scope that visits other devices
struct DeviceGuard {
    int dev;
    explicit DeviceGuard(int d) : dev(d) {}
    ~DeviceGuard() { (void)hipSetDevice(dev); }
};

// ---------------------------------------------------------------------------------------------
// Row shards behind one handle: shard s owns rows [s*n/S, (s+1)*n/S) on device (base + s) % count,
// with its own workspaces, stream and (HNSW) graph.  A query batch goes to every shard; per-shard top-k lists are
// copied peer-to-peer to the primary device and merged there by (distance, global position) -- for the exact scan
// that is bit for bit the unsharded result.  Nothing here needs torch or RCCL: the caller of nmslib_knn_query_batch
// gets all GPUs of the node.  Shards hold windows of this engine's host rows, not copies.
// ---------------------------------------------------------------------------------------------
int Engine::resolve_shards() const {
    if (loaded_graph_ || parent_ || ids_.empty()) return 1;
    int want = gpu_shards_;
    if (want < 0) {
        if (const char* env = getenv("NMSLIB_GPU_SHARDS")) want = atoi(env);
        // a process that was given its device (one rank per GPU, bench.py) keeps to it
        else if (getenv("NMSLIB_GPU_DEVICE") || getenv("LOCAL_RANK")) return 1;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return 1;  // finalize() reports the missing device
    }
    const size_t n = ids_.size();
    if (want < 0) {
        // automatic: the exact scan only -- sharded it returns the unsharded result bit for bit.  One HNSW graph per
        // GPU changes ids and recall with the number of visible devices and cannot be saved in the reference's
        // single-graph file, so HNSW is sharded only when gpu_shards / NMSLIB_GPU_SHARDS ask for it.
        if (method_ != Method::Brute) return 1;
        want = (int)std::min<size_t>((size_t)count, std::max<size_t>(1, n / 1000000));  // 1M rows per shard
    } else if (want == 0) want = count;
    if ((size_t)want > n) want = (int)n;
    return std::max(1, want);
}

void Engine::finalize_sharded(int nshards) {
    check_device();  // primary device: queries arrive there, results are merged there
    int count = 1;
    hip_check(hipGetDeviceCount(&count), "hipGetDeviceCount");
    const size_t n = ids_.size();
    shards_.clear();
    for (int s = 0; s < nshards; ++s) {
        std::unique_ptr<Engine> c(new Engine(space_name_, method_name_, is_u8() ? 2 : 0, 0));
        c->parent_ = this;
        c->view_lo_ = n * (size_t)s / (size_t)nshards;
        c->view_n_ = n * (size_t)(s + 1) / (size_t)nshards - c->view_lo_;
        c->dim_ = dim_;
        c->method_ = method_;
        c->bp_ = bp_;
        c->created_ = true;
        c->forced_device_ = (device_ + s) % count;
        shards_.push_back(std::move(c));
    }
    // build all shards at once: every shard has its own device (or at least its own stream)
    std::vector<std::string> errors((size_t)nshards);
    std::vector<std::thread> th;
    auto t0 = clk::now();
    for (int s = 0; s < nshards; ++s)
        th.emplace_back([&, s] {
            try {
                shards_[(size_t)s]->finalize();
            } catch (const std::exception& e) {
                errors[(size_t)s] = e.what();
                if (errors[(size_t)s].empty()) errors[(size_t)s] = "failed";
            }
        });
    for (auto& t : th) t.join();
    for (int s = 0; s < nshards; ++s)
        if (!errors[(size_t)s].empty()) {
            shards_.clear();
            throw EngineError(Err::IndexBuildFailed, "shard " + std::to_string(s) + ": " + errors[(size_t)s]);
        }
    build_seconds = std::chrono::duration<double>(clk::now() - t0).count();
    hip_check(hipSetDevice(device_), "hipSetDevice");
    d_ids_.ensure(std::max<size_t>(n, 1) * 4);  // position -> external id, applied after the merge
    hip_check(hipMemcpy(d_ids_.ptr(), ids_.data(), n * 4, hipMemcpyHostToDevice), "upload ids");
    d_n_ = n;
    for (hipEvent_t e : shard_events_) (void)hipEventDestroy(e);
    shard_events_.clear();
    for (int s = 0; s < nshards; ++s) {
        hip_check(hipSetDevice(shards_[(size_t)s]->device_), "hipSetDevice");
        hipEvent_t e;
        hip_check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
        shard_events_.push_back(e);
    }
    hip_check(hipSetDevice(device_), "hipSetDevice");
    if (!shard_ready_) hip_check(hipEventCreateWithFlags(&shard_ready_, hipEventDisableTiming), "hipEventCreate");
}

void Engine::knn_sharded(const void* d_queries, size_t nq, size_t elem_count, size_t k, int32_t* d_ids, float* d_dists,
                         int32_t* d_cnt, hipStream_t stream) {
    const size_t S = shards_.size();
    const size_t qbytes = nq * elem_count * elem_bytes();
    have_counters_ = false;
    DeviceGuard guard(device_);  // an exception thrown by a shard must not leave this thread on the shard's device
    ws_sh_ids_.ensure(S * nq * k * 4);
    ws_sh_d_.ensure(S * nq * k * 4);
    hip_check(hipEventRecord(shard_ready_, stream), "hipEventRecord");  // the queries are ready on the caller's stream
    for (size_t s = 0; s < S; ++s) {
        Engine& c = *shards_[s];
        hip_check(hipSetDevice(c.device_), "hipSetDevice");
        c.ef_ = ef_;
        c.algo_ = algo_;
        hip_check(hipStreamWaitEvent(c.stream_, shard_ready_, 0), "hipStreamWaitEvent");
        const void* q = d_queries;
        if (c.device_ != device_) {
            c.ws_q_.ensure(qbytes);
            hip_check(hipMemcpyPeerAsync(c.ws_q_.ptr(), c.device_, d_queries, device_, qbytes, c.stream_), "queries P2P");
            q = c.ws_q_.ptr();
        }
        c.ws_ids_.ensure(nq * k * 4);
        c.ws_dists_.ensure(nq * k * 4);
        c.knn_device(q, nq, elem_count, k, c.ws_ids_.as<int32_t>(), c.ws_dists_.as<float>(), nullptr, c.stream_);
        hip_check(hipMemcpyPeerAsync(ws_sh_ids_.as<int32_t>() + s * nq * k, device_, c.ws_ids_.ptr(), c.device_, nq * k * 4,
                                     c.stream_),
                  "ids P2P");
        hip_check(hipMemcpyPeerAsync(ws_sh_d_.as<float>() + s * nq * k, device_, c.ws_dists_.ptr(), c.device_, nq * k * 4,
                                     c.stream_),
                  "dists P2P");
        hip_check(hipEventRecord(shard_events_[s], c.stream_), "hipEventRecord");
        // NMSLIB_GPU_SHARD_SERIAL=1 (diagnostics): one shard at a time instead of all shards' kernels side by side
        static const bool serial = getenv("NMSLIB_GPU_SHARD_SERIAL") && atoi(getenv("NMSLIB_GPU_SHARD_SERIAL"));
        if (serial) hip_check(hipStreamSynchronize(c.stream_), "shard (serial)");
    }
    hip_check(hipSetDevice(device_), "hipSetDevice");
    last_path = shards_[0]->last_path;  // every shard takes the same path for the same batch shape
    for (size_t s = 0; s < S; ++s) hip_check(hipStreamWaitEvent(stream, shard_events_[s], 0), "hipStreamWaitEvent");
    hip_check(launch_merge_topk_ex(ws_sh_d_.as<float>(), ws_sh_ids_.as<int32_t>(), nq * k, (int)S, (int)nq, (int)k, d_dists,
                                   d_ids, d_cnt, d_ids_.as<int32_t>(), stream),
              "merge_topk");
}

// ---------------------------------------------------------------------------------------------
// Queries
// ---------------------------------------------------------------------------------------------
void Engine::knn_device(const void* d_queries, size_t nq, size_t elem_count, size_t k, int32_t* d_ids,
                        float* d_dists, int32_t* d_cnt, hipStream_t stream) {
    if (!created_) throw EngineError(Err::IndexBuildFailed, "Index not built");
    if (dirty_) finalize();
    check_device();
    if (nq == 0) return;
    if (k == 0) throw EngineError(Err::InvalidArgument, "k must be positive");
    last_stream_ = stream;
    if (!shards_.empty()) {
        if (size() > 0 && elem_count != dim_)
            throw EngineError(Err::QueryExecutionFailed, "query dimension does not match the index");
        knn_sharded(d_queries, nq, elem_count, k, d_ids, d_dists, d_cnt, stream);
        return;
    }
    if (d_n_ > 0 && elem_count != dim_)
        // SpaceLp::HiddenDistance CHECKs equal lengths (space_lp.cc:27-35) -> the query fails
        throw EngineError(Err::QueryExecutionFailed, "query dimension does not match the index");
    // very large batches go through in slices: the per-query workspaces (candidate buffers, visited bitsets)
    // stay bounded and every slice still fills the chip (the work counters then describe the last slice)
    const size_t slice = method_ == Method::Brute ? 32768 : 65536;
    const size_t qbytes = elem_count * elem_bytes();
    if (method_ == Method::Hnsw) {  // work counters for the WHOLE batch (slices write at their offset)
        ws_ndc_.ensure(nq * 4);
        ws_hops_.ensure(nq * 4);
        ws_hops_up_.ensure(nq * 4);
        ws_status_.ensure(nq * 4);
    }
    for (size_t q0 = 0; q0 < nq; q0 += slice) {
        const size_t m = std::min(slice, nq - q0);
        ctr_off_ = q0;
        const void* qs = static_cast<const char*>(d_queries) + q0 * qbytes;
        int32_t* cnt = d_cnt ? d_cnt + q0 : nullptr;
        if (method_ == Method::Brute) knn_brute(qs, m, k, d_ids + q0 * k, d_dists + q0 * k, cnt, stream);
        else knn_hnsw(qs, m, k, d_ids + q0 * k, d_dists + q0 * k, cnt, stream);
    }
}

void Engine::fast_tile_counts(size_t* tiles, size_t* precise, size_t* fallback) {
    *tiles = *precise = *fallback = 0;
    if (!shards_.empty()) {   // row shards: every shard sees the same batch; the first one's counters stand for the handle
        shards_[0]->fast_tile_counts(tiles, precise, fallback);
        return;
    }
    if ((last_path != 1 && last_path != 3) || !fast_flags_ || fast_nqt_ <= 0) return;
    std::vector<int> h((size_t)fast_nqt_ * 2, 0);
    hip_check(hipSetDevice(device_), "hipSetDevice");
    hip_check(hipStreamSynchronize(last_stream_ ? last_stream_ : stream_), "stats");
    hip_check(hipMemcpy(h.data(), fast_flags_, (size_t)fast_nqt_ * (fast_has_precise_ ? 2 : 1) * sizeof(int), hipMemcpyDeviceToHost),
              "stats flags");
    *tiles = (size_t)fast_nqt_;
    for (int i = 0; i < fast_nqt_; ++i) {
        *fallback += h[i] != 0;
        if (fast_has_precise_) *precise += h[(size_t)fast_nqt_ + i] != 0;
    }
}

size_t Engine::hnsw_redone() {
    if (!shards_.empty()) return shards_[0]->hnsw_redone();
    if (last_path != 4 || !hnsw_fix_valid_) return 0;
    int32_t v = 0;
    hip_check(hipSetDevice(device_), "hipSetDevice");
    hip_check(hipStreamSynchronize(last_stream_ ? last_stream_ : stream_), "stats");
    hip_check(hipMemcpy(&v, ws_fix_.ptr(), 4, hipMemcpyDeviceToHost), "stats redo count");
    return (size_t)v;
}

void Engine::knn_brute(const void* d_queries, size_t nq, size_t k, int32_t* d_ids, float* d_dists, int32_t* d_cnt,
                       hipStream_t stream) {
    have_counters_ = false;
    const int dim_eff = d_n_ ? (int)dim_ : 1;
    if (k > (size_t)BF_MAX_K) {
        // beyond the selection kernels' capacity: per query one pass with the reference formula + one stable radix sort
        const int ld = is_u8() ? 128 : ldb_;
        const int elem = is_u8() ? 1 : 4;
        const size_t n = d_n_;
        ws_qpad_.ensure(std::max<size_t>(nq, 1) * ld * elem);
        hip_check(launch_pad_rows(d_queries, (int)nq, dim_eff, ws_qpad_.ptr(), (int)nq, ld, elem, stream), "pad queries");
        ws_rdist_.ensure(std::max<size_t>(n, 1) * 4);
        ws_bigk_.ensure(std::max<size_t>(n, 1) * 16);
        const size_t tb = bf_bigk_temp_bytes((int)n);
        ws_bigk_tmp_.ensure(tb);
        hip_check(launch_bf_bigk(space_, d_rows_.ptr(), ld, (int)n, ws_qpad_.ptr(), (size_t)ld * elem, (int)nq, dim_eff, (int)k,
                                 d_ids_.as<int32_t>(), ws_rdist_.as<float>(), ws_bigk_.as<uint32_t>(), ws_bigk_tmp_.ptr(), tb,
                                 d_ids, d_dists, d_cnt, stream),
                  "bf_bigk");
        last_path = 5;
        return;
    }
    if (is_u8()) {
        // large batches: thresholds fixed by a sample pass, then one streaming scan (bf_kernels.hip, bf_scan_u8_kernel)
        const BfU8Fast f = bf_u8_fast_plan((int)d_n_, (int)nq, (int)k);
        if (f.use) {
            ws_qpad_.ensure((size_t)f.qpad * 128);
            ws_cand_.ensure(bf_cand_elems(f.fallback) * 8);
            ws_cnt_.ensure(bf_cnt_elems(f.fallback) * 4);
            ws_u8_cand_.ensure(bf_u8_top8_elems(f) * 4);
            ws_u8_thr_.ensure((size_t)f.qpad * 4 + (size_t)f.nqt * 4 + 64);
            ws_u8_list_.ensure(bf_u8_list_elems(f) * 4);
            ws_u8_listcnt_.ensure(bf_u8_listcnt_elems(f) * 4);
            int* thr = ws_u8_thr_.as<int>();
            int* tile_fail = thr + f.qpad;
            // (padding happens inside the fast path's one preparation kernel)
            hipEvent_t eb = nullptr, ee = nullptr;
            if (prof_ && prof_events_.size() < 65536) {
                hip_check(hipEventCreate(&eb), "hipEventCreate");
                hip_check(hipEventCreate(&ee), "hipEventCreate");
                prof_events_.emplace_back(eb, ee);
            }
            hip_check(launch_bf_u8_fast(f, (int)d_n_, (int)nq, (int)k, d_rows_.as<uint8_t>(), d_rows_i8_.as<uint8_t>(),
                                        d_aux_.as<int32_t>(), d_auxh_.as<int32_t>(), ws_qpad_.as<uint8_t>(),
                                        ws_u8_cand_.as<int>(),
                                        ws_cand_.as<unsigned long long>(), ws_cnt_.as<int>(), thr,
                                        ws_u8_list_.as<uint32_t>(), ws_u8_listcnt_.as<int>(), tile_fail,
                                        d_ids_.as<int32_t>(), d_ids, d_dists, d_cnt, eb, ee, stream,
                                        static_cast<const uint8_t*>(d_queries)),
                      "bf_u8_fast");
            last_path = 3;
            fast_flags_ = tile_fail;
            fast_nqt_ = f.nqt;
            fast_has_precise_ = false;
            return;
        }
    }
    if (!is_u8() && have_bf16_) {
        // large batches at D <= 128: split-bf16 MFMA selection with sample-fixed thresholds (bf_scan_f32_kernel)
        const BfF32Fast f = bf_f32_fast_plan((int)d_n_, dim_eff, (int)nq, (int)k, space_, centred_);
        if (f.use) {
            const int ldb = ldb_;
            ws_qpad_.ensure((size_t)f.qpad * ldb * 4);
            // (uncentred rows: padding happens inside the fast path's one preparation kernel)
            if (centred_) hip_check(launch_pad_rows(d_queries, (int)nq, dim_eff, ws_qpad_.ptr(), f.qpad, ldb, 4, stream), "pad queries");
            const float* qsel = ws_qpad_.as<float>();
            if (centred_) {
                ws_qsel_.ensure((size_t)f.qpad * ldb * 4);
                hip_check(launch_center_rows(ws_qpad_.as<float>(), d_mean_.as<float>(), f.qpad, (int)nq, ldb, dim_eff,
                                             ws_qsel_.as<float>(), stream),
                          "centre queries");
                qsel = ws_qsel_.as<float>();
            }
            if (f.cosc) {   // centred cosine / angular: the augmented queries (query_aug_cosc_kernel) are what the scans see
                ws_qaux_.ensure((size_t)f.qpad * 16);
                ws_qaug_.ensure((size_t)f.qpad * f.dp * 4);
                hip_check(launch_query_aux_cosc(ws_qpad_.as<float>(), ws_qsel_.as<float>(), f.qpad, ldb, dim_eff, mu_norm_,
                                                ws_qaux_.as<float>(), stream),
                          "query aux");
                hip_check(launch_query_aug_cosc(ws_qsel_.as<float>(), ws_qaux_.as<float>(), (int)nq, f.qpad, ldb, dim_eff,
                                                cosc_lambda_, ws_qaug_.as<float>(), f.dp, stream),
                          "augmented queries");
                qsel = ws_qaug_.as<float>();
            }
            ws_cand_.ensure(bf_cand_elems(f.fallback) * 8);
            ws_cnt_.ensure(bf_cnt_elems(f.fallback) * 4);
            ws_f32_q_.ensure((size_t)f.qpad * f.dp * 2 * 3);   // bf16 hi, bf16 lo, fp16
            ws_u8_cand_.ensure(bf_f32_top8_elems(f) * 4);
            ws_u8_thr_.ensure(bf_f32_thr_bytes(f));
            ws_flags_.ensure((size_t)f.fallback.nqt * 4 + 64);
            ws_u8_list_.ensure(bf_f32_list_elems(f) * 4);
            ws_u8_listcnt_.ensure(bf_f32_listcnt_elems(f) * 4);
            float* thr = ws_u8_thr_.as<float>();
            int* tile_fail = reinterpret_cast<int*>(thr + 2 * (size_t)f.qpad);
            char* qh = ws_f32_q_.as<char>();
            char* ql = qh + (size_t)f.qpad * f.dp * 2;
            BfF16Side h16{d_f16_hi_.ptr(), d_auxp16_.as<float>(), ql + (size_t)f.qpad * f.dp * 2, f16_scale_, bres16_, f16_scale_q_};
            hipEvent_t eb = nullptr, ee = nullptr;
            if (prof_ && prof_events_.size() < 65536) {
                hip_check(hipEventCreate(&eb), "hipEventCreate");
                hip_check(hipEventCreate(&ee), "hipEventCreate");
                prof_events_.emplace_back(eb, ee);
            }
            hip_check(launch_bf_f32_fast(f, space_, (int)d_n_, dim_eff, ldb, (int)nq, (int)k, d_rows_.as<float>(),
                                         centred_ ? d_rows_sel_.as<float>() : d_rows_.as<float>(), d_aux_.as<float>(),
                                         d_bf_hi_.ptr(), d_bf_lo_.ptr(), d_auxp_.as<float>(), f.cosc ? bmax_c_ : bmax_,
                                         f.cosc ? bres_c_ : bres_, ws_qpad_.as<float>(), qsel, qh, ql,
                                         ws_u8_cand_.as<float>(), ws_cand_.as<unsigned long long>(), ws_cnt_.as<int>(), thr,
                                         ws_u8_list_.as<uint32_t>(), ws_u8_listcnt_.as<int>(), tile_fail, ws_flags_.as<int>(),
                                         d_ids_.as<int32_t>(), d_ids, d_dists, d_cnt, eb, ee, stream,
                                         centred_ ? nullptr : static_cast<const float*>(d_queries), centred_ ? nullptr : ws_qpad_.as<float>(),
                                         f.cosc ? ws_qaux_.as<float>() : nullptr, f.cosc ? ws_qsel_.as<float>() : nullptr, f.dp, h16),
                      "bf_f32_fast");
            last_path = 1;
            fast_flags_ = tile_fail;
            fast_nqt_ = f.nqt;
            fast_has_precise_ = true;
            return;
        }
    }
    last_path = is_u8() ? 2 : 0;
    BfPlan p = bf_make_plan((int)d_n_, dim_eff, (int)nq, (int)k, is_u8());
    if (d_n_ == 0) p.ldb = is_u8() ? 128 : f32_row_stride(dim_eff);
    const int elem = is_u8() ? 1 : 4;
    ws_qpad_.ensure((size_t)p.qpad * p.ldb * elem);
    ws_cand_.ensure(bf_cand_elems(p) * 8);
    ws_cnt_.ensure(bf_cnt_elems(p) * 4);
    hip_check(launch_pad_rows(d_queries, (int)nq, dim_eff, ws_qpad_.ptr(), p.qpad, p.ldb, elem, stream), "pad queries");
    if (centred_) {
        ws_qsel_.ensure((size_t)p.qpad * p.ldb * 4);
        hip_check(launch_center_rows(ws_qpad_.as<float>(), d_mean_.as<float>(), p.qpad, (int)nq, p.ldb, dim_eff,
                                     ws_qsel_.as<float>(), stream),
                  "centre queries");
        if (space_ != SP_L2) {
            ws_qaux_.ensure((size_t)p.qpad * 16);
            hip_check(launch_query_aux_cosc(ws_qpad_.as<float>(), ws_qsel_.as<float>(), p.qpad, p.ldb, dim_eff, mu_norm_,
                                            ws_qaux_.as<float>(), stream),
                      "query aux");
        }
    }
    prof_begin(stream);
    if (is_u8()) {
        hip_check(launch_bf_select_u8(p, d_rows_i8_.as<uint8_t>(), d_aux_.as<int32_t>(), ws_qpad_.as<uint8_t>(),
                                      ws_cand_.as<unsigned long long>(), ws_cnt_.as<int>(), stream),
                  "bf_select_u8");
    } else if (space_ == SP_L1 || space_ == SP_LINF) {
        hip_check(launch_bf_select_direct_f32(p, space_, d_rows_.as<float>(), ws_qpad_.as<float>(),
                                              ws_cand_.as<unsigned long long>(), ws_cnt_.as<int>(), stream),
                  "bf_select_direct");
    } else {
        // selection + re-rank in one chain (l2: verified, with the exact tail for tiles of near-duplicates)
        ws_flags_.ensure((size_t)p.nqt * 4 + 64);
        hip_check(launch_bf_adaptive_f32(p, space_, dim_eff, (int)k, d_rows_.as<float>(),
                                         centred_ ? d_rows_sel_.as<float>() : d_rows_.as<float>(), d_aux_.as<float>(),
                                         ws_qpad_.as<float>(), centred_ ? ws_qsel_.as<float>() : ws_qpad_.as<float>(),
                                         (centred_ && space_ != SP_L2) ? ws_qaux_.as<float>() : nullptr, bmax_,
                                         ws_cand_.as<unsigned long long>(), ws_cnt_.as<int>(), ws_flags_.as<int>(),
                                         d_ids_.as<int32_t>(), d_ids, d_dists, d_cnt, nullptr, 1, stream),
                  "bf_adaptive_f32");
        prof_end(stream);
        return;
    }
    prof_end(stream);
    hip_check(launch_bf_rerank(p, space_, dim_eff, (int)k, d_rows_.ptr(), ws_qpad_.ptr(),
                               ws_cand_.as<unsigned long long>(), ws_cnt_.as<int>(), d_ids_.as<int32_t>(), d_ids,
                               d_dists, d_cnt, stream),
              "bf_rerank");
}

void Engine::knn_hnsw(const void* d_queries, size_t nq, size_t k, int32_t* d_ids, float* d_dists, int32_t* d_cnt,
                      hipStream_t stream) {
    const int ef = ef_;
    last_path = 4;
    hnsw_fix_valid_ = false;
    // Hnsw::Search, hnsw.cc:724: algoType=old, or hybrid with ef >= 1000, runs SearchOld
    if (algo_ == "old" || (algo_ == "hybrid" && ef >= 1000)) {
        knn_hnsw_old(d_queries, nq, k, d_ids, d_dists, d_cnt, stream);
        return;
    }
    if (std::max<size_t>(ef, k) > 1024 || hnsw_nbcap(dg_) > HNSW_NBCAP_LDS) {
        // beyond the LDS kernels' sorted array, or adjacency lists longer than their frontier arrays (maxM0 > 254): the
        // same algorithm with the array in HBM and lists walked in chunks (slices of bounded workspace)
        int32_t* cnt2 = d_cnt;
        if (!cnt2) {
            ws_outcnt_.ensure(nq * 4);
            cnt2 = ws_outcnt_.as<int32_t>();
        }
        const size_t cap = std::max<size_t>(ef, k), words = (d_n_ + 31) / 32;
        const size_t per_q = cap * 8 + words * 4;
        const size_t slice = std::max<size_t>(1, std::min<size_t>(nq, ((size_t)4 << 30) / per_q));
        const size_t qbytes = dim_ * elem_bytes();
        have_counters_ = true;
        for (size_t q0 = 0; q0 < nq; q0 += slice) {
            const size_t m = std::min(slice, nq - q0);
            ws_old_a_.ensure(m * cap * 4);
            ws_old_r_.ensure(m * cap * 4);
            ws_bitset_.ensure(m * words * 4);
            hip_check(hipMemsetAsync(ws_bitset_.ptr(), 0, m * words * 4, stream), "clear visited bitset");
            if (q0 == 0) prof_begin(stream);
            hip_check(launch_hnsw_search_big(dg_, (int)m, (int)k, ef, static_cast<const char*>(d_queries) + q0 * qbytes,
                                             ws_bitset_.as<uint32_t>(), ws_old_a_.as<float>(), ws_old_r_.as<int32_t>(),
                                             d_ids + q0 * k, d_dists + q0 * k, cnt2 + q0, ws_ndc_.as<int32_t>() + ctr_off_ + q0,
                                             ws_hops_.as<int32_t>() + ctr_off_ + q0, ws_hops_up_.as<int32_t>() + ctr_off_ + q0,
                                             ws_status_.as<int32_t>() + ctr_off_ + q0, stream),
                      "hnsw_search(big)");
            if (q0 + slice >= nq) prof_end(stream);
        }
        return;
    }
    int32_t* cnt = d_cnt;
    if (!cnt) {
        ws_outcnt_.ensure(nq * 4);
        cnt = ws_outcnt_.as<int32_t>();
    }
    int32_t* ndc = ws_ndc_.as<int32_t>() + ctr_off_;
    int32_t* hops = ws_hops_.as<int32_t>() + ctr_off_;
    int32_t* hops_up = ws_hops_up_.as<int32_t>() + ctr_off_;
    int32_t* status = ws_status_.as<int32_t>() + ctr_off_;
    HnswSearchPlan p = hnsw_make_plan(dg_, (int)nq, (int)k, ef, false);
    have_counters_ = true;
    if (p.table_size == 0) {
        ws_bitset_.ensure(nq * p.bitset_words * 4);
        hip_check(hipMemsetAsync(ws_bitset_.ptr(), 0, nq * p.bitset_words * 4, stream), "clear visited bitset");
        prof_begin(stream);
        hip_check(launch_hnsw_search(dg_, p, d_queries, ws_bitset_.as<uint32_t>(), d_ids, d_dists, cnt, ndc, hops, hops_up,
                                     status, stream),
                  "hnsw_search");
        prof_end(stream);
        return;
    }
    // The LDS visited table is exact but finite.  Queries that fill it append themselves to a list on the device and
    // are re-run by a second, small launch of the bitset variant that walks that list -- the host never looks at it,
    // so the call only enqueues work (include/nmslib_gpu.h).
    const int fix_slots = (int)std::min<size_t>(nq, 128);
    HnswSearchPlan pb = hnsw_make_plan(dg_, (int)nq, (int)k, ef, true);
    ws_fix_.ensure((nq + 16) * 4);
    ws_bitset_.ensure((size_t)fix_slots * pb.bitset_words * 4);
    int32_t* fix_count = ws_fix_.as<int32_t>();
    int32_t* fix_list = fix_count + 16;
    hip_check(hipMemsetAsync(fix_count, 0, 4, stream), "clear overflow count");
    hnsw_fix_valid_ = true;
    prof_begin(stream);
    hip_check(launch_hnsw_search_fix(dg_, p, d_queries, nullptr, 0, fix_list, fix_count, d_ids, d_dists, cnt, ndc, hops,
                                     hops_up, status, stream),
              "hnsw_search");
    prof_end(stream);
    hip_check(launch_hnsw_search_fix(dg_, pb, d_queries, ws_bitset_.as<uint32_t>(), fix_slots, fix_list, fix_count, d_ids,
                                     d_dists, cnt, ndc, hops, hops_up, status, stream),
              "hnsw_search(bitset)");
}

// Hnsw::SearchOld (hnsw_distfunc_opt.cc:46-150) on the GPU: no limit on ef or k.  Queues that outgrow LDS live in
// per-query HBM workspaces, so very large batches go through in slices of bounded workspace.
void Engine::knn_hnsw_old(const void* d_queries, size_t nq, size_t k, int32_t* d_ids, float* d_dists, int32_t* d_cnt,
                          hipStream_t stream) {
    const int ef = ef_;
    int32_t* cnt = d_cnt;
    if (!cnt) {
        ws_outcnt_.ensure(nq * 4);
        cnt = ws_outcnt_.as<int32_t>();
    }
    const size_t qbytes = dim_ * elem_bytes();
    const size_t co = ctr_off_;
    bool force_bitset = false;
    int heap_cap = 0;
    std::vector<int32_t> status(nq);
    for (int attempt = 0; attempt < 3; ++attempt) {
        HnswSearchPlan p0 = hnsw_make_plan_old(dg_, 1, (int)k, ef, force_bitset, heap_cap);
        const size_t per_q = hnsw_old_ws_a(p0) + hnsw_old_ws_r(p0) + hnsw_old_ws_heap(p0) + p0.bitset_words * 4;
        const size_t budget = (size_t)4 << 30;
        const size_t slice = std::max<size_t>(1, std::min<size_t>(nq, per_q ? budget / per_q : nq));
        for (size_t q0 = 0; q0 < nq; q0 += slice) {
            const size_t m = std::min(slice, nq - q0);
            HnswSearchPlan p = hnsw_make_plan_old(dg_, (int)m, (int)k, ef, force_bitset, heap_cap);
            uint32_t* bitset = nullptr;
            if (p.table_size == 0) {
                ws_bitset_.ensure(m * p.bitset_words * 4);
                hip_check(hipMemsetAsync(ws_bitset_.ptr(), 0, m * p.bitset_words * 4, stream), "clear visited bitset");
                bitset = ws_bitset_.as<uint32_t>();
            }
            ws_old_a_.ensure(std::max<size_t>(16, m * hnsw_old_ws_a(p)));
            ws_old_r_.ensure(std::max<size_t>(16, m * hnsw_old_ws_r(p)));
            ws_old_heap_.ensure(std::max<size_t>(16, m * hnsw_old_ws_heap(p)));
            if (q0 == 0) prof_begin(stream);  // all slices of one attempt are one timed interval
            hip_check(launch_hnsw_search_old(dg_, p, static_cast<const char*>(d_queries) + q0 * qbytes, bitset,
                                             ws_old_a_.ptr(), ws_old_r_.ptr(), ws_old_heap_.ptr(), d_ids + q0 * k,
                                             d_dists + q0 * k, cnt + q0, ws_ndc_.as<int32_t>() + co + q0,
                                             ws_hops_.as<int32_t>() + co + q0, ws_hops_up_.as<int32_t>() + co + q0,
                                             ws_status_.as<int32_t>() + co + q0, stream),
                      "hnsw_search_old");
            if (q0 + slice >= nq) prof_end(stream);
        }
        have_counters_ = true;
        // status 1: the LDS visited table filled up -> HBM bitsets; status 2: the candidate heap outgrew its bound
        hip_check(hipMemcpyAsync(status.data(), ws_status_.as<int32_t>() + co, nq * 4, hipMemcpyDeviceToHost, stream), "status");
        hip_check(hipStreamSynchronize(stream), "hnsw_search_old");
        bool any1 = false, any2 = false;
        for (int32_t v : status) {
            any1 |= v == 1;
            any2 |= v == 2;
        }
        if (!any1 && !any2) return;
        if (any1) force_bitset = true;
        if (any2) heap_cap = (int)std::min<size_t>(d_n_ + 1, (size_t)INT32_MAX);
    }
    throw EngineError(Err::QueryExecutionFailed, "SearchOld: queue workspaces exhausted");
}

// Pinned host staging buffer (grow-only): PCIe copies from/to pageable memory are staged by the runtime in small
// chunks and block the calling thread; one pinned block makes the batch's H2D and D2H a single DMA each.
void* Engine::pinned(size_t bytes) {
    if (bytes <= pinned_bytes_ && pinned_) return pinned_;
    if (pinned_) (void)hipHostFree(pinned_);
    pinned_ = nullptr;
    pinned_bytes_ = 0;
    hip_check(hipHostMalloc(&pinned_, bytes, hipHostMallocDefault), "hipHostMalloc");
    pinned_bytes_ = bytes;
    return pinned_;
}

void Engine::knn_host(const void* queries, size_t nq, size_t elem_count, size_t k, const int32_t** ids, const float** dists,
                      const int32_t** cnt) {
    if (!created_) throw EngineError(Err::IndexBuildFailed, "Index not built");
    if (dirty_) finalize();
    check_device();
    const size_t qbytes = nq * elem_count * elem_bytes();
    const size_t rbytes = nq * k * 4;
    // device: queries | ids | dists | counts in ONE block each way; host: one pinned block
    ws_q_.ensure(qbytes);
    ws_ids_.ensure(2 * rbytes + nq * 4);
    int32_t* d_ids = ws_ids_.as<int32_t>();
    float* d_dists = reinterpret_cast<float*>(d_ids + nq * k);
    int32_t* d_cnt = d_ids + 2 * nq * k;
    char* hp = static_cast<char*>(pinned(std::max(qbytes, 2 * rbytes + nq * 4)));
    std::memcpy(hp, queries, qbytes);
    hip_check(hipMemcpyAsync(ws_q_.ptr(), hp, qbytes, hipMemcpyHostToDevice, stream_), "queries H2D");
    knn_device(ws_q_.ptr(), nq, elem_count, k, d_ids, d_dists, d_cnt, stream_);
    hip_check(hipMemcpyAsync(hp, d_ids, 2 * rbytes + nq * 4, hipMemcpyDeviceToHost, stream_), "results D2H");
    hip_check(hipStreamSynchronize(stream_), "knn");
    *ids = reinterpret_cast<const int32_t*>(hp);
    *dists = reinterpret_cast<const float*>(hp + rbytes);
    *cnt = reinterpret_cast<const int32_t*>(hp + 2 * rbytes);
}

size_t Engine::range_host(const void* query, size_t elem_count, double radius, size_t capacity, int32_t* ids,
                          float* dists) {
    if (!created_) throw EngineError(Err::IndexBuildFailed, "Index not built");
    if (dirty_) finalize();
    check_device();
    const size_t n = d_n_;
    if (n == 0 || capacity == 0) return 0;
    if (elem_count != dim_) throw EngineError(Err::QueryExecutionFailed, "query dimension does not match the index");
    if (!shards_.empty()) {
        // insertion order = shard order; shards report global positions
        DeviceGuard guard(device_);   // (a shard that throws must not leave this thread on its device)
        size_t got = 0;
        for (auto& c : shards_) {
            if (got >= capacity) break;
            got += c->range_host(query, elem_count, radius, capacity - got, ids + got, dists + got);
        }
        for (size_t i = 0; i < got; ++i) ids[i] = ids_[(size_t)ids[i]];
        hip_check(hipSetDevice(device_), "hipSetDevice");
        return got;
    }
    // RangeQuery<dist_t>(space, obj, static_cast<dist_t>(radius)), nmslib_c.cpp:1092-1093
    const float r = is_u8() ? (float)(int)radius : (float)radius;
    const int ld = is_u8() ? 128 : ldb_;
    const int elem = is_u8() ? 1 : 4;
    ws_qpad_.ensure((size_t)ld * elem);
    ws_q_.ensure(elem_count * elem);
    ws_rdist_.ensure(n * 4);
    ws_rcnt_.ensure(range_count_elems((int)n) * 4);
    ws_ids_.ensure(capacity * 4);
    ws_dists_.ensure(capacity * 4);
    hip_check(hipMemcpyAsync(ws_q_.ptr(), query, elem_count * elem, hipMemcpyHostToDevice, stream_), "query H2D");
    hip_check(launch_pad_rows(ws_q_.ptr(), 1, (int)dim_, ws_qpad_.ptr(), 1, ld, elem, stream_), "pad query");
    hip_check(launch_range_search(space_, d_rows_.ptr(), ld, (int)n, ws_qpad_.ptr(), (int)dim_, r, d_ids_.as<int32_t>(),
                                  ws_rdist_.as<float>(), ws_rcnt_.as<int>(), (int)std::min<size_t>(capacity, INT32_MAX),
                                  ws_ids_.as<int32_t>(), ws_dists_.as<float>(), stream_),
              "range search");
    int total = 0;
    hip_check(hipMemcpyAsync(&total, ws_rcnt_.as<int>() + (range_count_elems((int)n) - 1), 4, hipMemcpyDeviceToHost, stream_),
              "range count");
    hip_check(hipStreamSynchronize(stream_), "range search");
    const size_t m = std::min<size_t>((size_t)total, capacity);
    if (m) {
        hip_check(hipMemcpyAsync(ids, ws_ids_.ptr(), m * 4, hipMemcpyDeviceToHost, stream_), "range ids");
        hip_check(hipMemcpyAsync(dists, ws_dists_.ptr(), m * 4, hipMemcpyDeviceToHost, stream_), "range dists");
        hip_check(hipStreamSynchronize(stream_), "range search");
    }
    return m;
}

float Engine::pair_distance(size_t p1, size_t p2) {
    // Space::IndexTimeDistance on the ORIGINAL rows (nmslib_c.cpp:1166), one wave on the GPU
    check_device();
    const size_t rb = is_u8() ? 128 : (size_t)f32_row_stride((int)dim_) * 4;
    ws_pair_.ensure(2 * rb + 16);
    hip_check(hipMemsetAsync(ws_pair_.ptr(), 0, 2 * rb + 16, stream_), "pair clear");
    char* base = ws_pair_.as<char>();
    hip_check(hipMemcpyAsync(base, host_row(p1), row_bytes(), hipMemcpyHostToDevice, stream_), "pair H2D");
    hip_check(hipMemcpyAsync(base + rb, host_row(p2), row_bytes(), hipMemcpyHostToDevice, stream_), "pair H2D");
    float* out = reinterpret_cast<float*>(base + 2 * rb);
    hip_check(launch_pair_distance(space_, base, base + rb, (int)dim_, out, stream_), "pair_distance");
    float v = 0;
    hip_check(hipMemcpyAsync(&v, out, 4, hipMemcpyDeviceToHost, stream_), "pair D2H");
    hip_check(hipStreamSynchronize(stream_), "pair_distance");
    return v;
}

// ---------------------------------------------------------------------------------------------
// Persistence: the reference's own formats
//   <path>      optimized HNSW index, hnsw.cc:774-806 / 1025-1074
//   <path>.dat  object vector, space.cc:88-105
// ---------------------------------------------------------------------------------------------
template <typename T>
static void wr(std::ostream& o, const T& v) {
    o.write(reinterpret_cast<const char*>(&v), sizeof(T));
}
template <typename T>
static void rd(std::istream& i, T& v) {
    i.read(reinterpret_cast<char*>(&v), sizeof(T));
    if (!i) throw EngineError(Err::DataIO, "unexpected end of index file");
}

void Engine::save(const std::string& path, bool save_data) {
    if (!created_) throw EngineError(Err::InvalidArgument, "Index not built");
    if (method_ == Method::Hnsw && !loaded_graph_ && (shards_.size() > 1 || (dirty_ && resolve_shards() > 1)))
        throw EngineError(Err::DataIO, "a sharded HNSW index holds one graph per GPU and has no single-file form; "
                                       "build with gpu_shards=1 to save it in the reference's format");
    ensure_graph();  // persistence is host-side: no device needed
    const size_t n = ids_.size();
    if (save_data) {
        std::ofstream out(path + ".dat", std::ios::binary);
        if (!out) throw EngineError(Err::DataIO, "Cannot open file '" + path + ".dat' for writing");
        wr(out, (uint64_t)n);
        std::vector<char> buf(16 + stored_row_bytes());
        for (size_t i = 0; i < n; ++i) {
            const uint64_t len = 16 + stored_row_bytes();
            wr(out, len);
            const int32_t id = ids_[i], label = -1;
            const uint64_t dl = stored_row_bytes();
            std::memcpy(&buf[0], &id, 4);
            std::memcpy(&buf[4], &label, 4);
            std::memcpy(&buf[8], &dl, 8);
            stored_row(i, &buf[16]);
            out.write(buf.data(), (std::streamsize)buf.size());
        }
        if (!out) throw EngineError(Err::DataIO, "write failed: " + path + ".dat");
    }
    if (method_ != Method::Hnsw)
        // Index::SaveIndex default throws for SeqSearch (include/index.h:56-58)
        throw EngineError(Err::DataIO, "SaveIndex is not implemented for method: Sequential search");
    int dist_func = -1;
    if (!bp_.skip_optimized) {
        if (space_ == SP_L2) dist_func = (dim_ % 16 == 0) ? 1 : 2;
        else if (space_ == SP_COSINE) dist_func = 3;
        else if (space_ == SP_NEGDOT) dist_func = 4;
        else if (space_ == SP_L1) dist_func = 5;
        else if (space_ == SP_LINF) dist_func = 6;
    }
    std::ofstream out(path, std::ios::binary);
    if (!out) throw EngineError(Err::DataIO, "Cannot open file '" + path + "' for writing");
    const HostGraph& g = graph_;
    if (dist_func < 0) {
        // regular (non-optimized) index, SaveRegularIndexBin hnsw.cc:808-840
        wr(out, (uint32_t)0);
        wr(out, (uint32_t)n);
        wr(out, (int32_t)g.maxlevel);
        wr(out, (uint32_t)g.enterpoint);
        wr(out, (uint64_t)g.M);
        wr(out, (uint64_t)g.maxM);
        wr(out, (uint64_t)g.maxM0);
        for (size_t i = 0; i < n; ++i) {
            wr(out, (uint32_t)g.levels[i]);
            for (int l = 0; l <= g.levels[i]; ++l) {
                const int32_t* L = l == 0 ? &g.links0[i * (g.maxM0 + 1)]
                                          : &g.up_links[g.up_off[i] + (size_t)(l - 1) * (g.maxM + 1)];
                wr(out, (uint32_t)L[0]);
                out.write(reinterpret_cast<const char*>(L + 1), (std::streamsize)L[0] * 4);
            }
        }
    } else {
        const uint64_t data_section = 16 + dim_ * 4;
        const uint64_t mem_per_obj = data_section + (uint64_t)(g.maxM0 + 1) * 4;
        wr(out, (uint32_t)1);
        wr(out, (uint32_t)n);
        wr(out, mem_per_obj);
        wr(out, data_section);   // offsetLevel0_
        wr(out, (uint64_t)0);    // offsetData_
        wr(out, (int32_t)g.maxlevel);
        wr(out, (uint32_t)g.enterpoint);
        wr(out, (uint64_t)g.maxM);
        wr(out, (uint64_t)g.maxM0);
        wr(out, (int32_t)dist_func);
        wr(out, (uint64_t)3);    // searchMethod_
        std::vector<char> rec(mem_per_obj);
        std::vector<float> row(dim_);
        for (size_t i = 0; i < n; ++i) {
            const int32_t id = ids_[i], label = -1;
            const uint64_t dl = dim_ * 4;
            std::memcpy(&rec[0], &id, 4);
            std::memcpy(&rec[4], &label, 4);
            std::memcpy(&rec[8], &dl, 8);
            const float* src = (loaded_graph_ && !graph_rows_.empty()) ? &graph_rows_[i * dim_] : &rows_f32_[i * dim_];
            std::memcpy(row.data(), src, dim_ * 4);
            if (space_ == SP_COSINE && !(loaded_graph_ && !graph_rows_.empty())) {
                float s = 0;  // NormalizeVect, hnsw.h:486-497
                for (size_t d = 0; d < dim_; ++d) s += row[d] * row[d];
                if (s != 0.0f) {
                    s = 1 / std::sqrt(s);
                    for (size_t d = 0; d < dim_; ++d) row[d] *= s;
                }
            }
            std::memcpy(&rec[16], row.data(), dim_ * 4);
            std::memcpy(&rec[data_section], &g.links0[i * (g.maxM0 + 1)], (size_t)(g.maxM0 + 1) * 4);
            out.write(rec.data(), (std::streamsize)rec.size());
        }
        for (size_t i = 0; i < n; ++i) {
            const uint32_t bytes = (uint32_t)((size_t)g.levels[i] * (g.maxM + 1) * 4);
            wr(out, bytes);
            if (bytes) out.write(reinterpret_cast<const char*>(&g.up_links[g.up_off[i]]), bytes);
        }
    }
    // Trailer (ignored by the reference's readers, which stop after the last node): lets this
    // library restore the space of an index it saved itself; the reference hard-codes "l2"
    // on load (nmslib_c.cpp:1421-1429).
    char trailer[32] = {0};
    std::memcpy(trailer, "GFXKNNv1", 8);
    std::strncpy(trailer + 8, space_name_.c_str(), 23);
    out.write(trailer, 32);
    if (!out) throw EngineError(Err::DataIO, "write failed: " + path);
}

std::unique_ptr<Engine> Engine::load(const std::string& path, int data_type, int dist_type, bool load_data) {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw EngineError(Err::DataIO, "Cannot open file '" + path + "' for reading");
    std::string saved_space;
    {
        in.seekg(0, std::ios::end);
        const std::streamoff sz = in.tellg();
        if (sz >= 36) {
            char trailer[32];
            in.seekg(sz - 32);
            in.read(trailer, 32);
            if (std::memcmp(trailer, "GFXKNNv1", 8) == 0) {
                trailer[31] = 0;
                saved_space = trailer + 8;
            }
        }
        in.clear();
        in.seekg(0);
    }
    uint32_t flag = 0;
    rd(in, flag);
    std::unique_ptr<Engine> e;
    // objects first (.dat), as nmslib_load_index does when load_data is set (nmslib_c.cpp:1430-1434)
    std::vector<int32_t> dat_ids;
    std::vector<char> dat_payload;
    size_t dat_len = 0;
    if (load_data) {
        std::ifstream d(path + ".dat", std::ios::binary);
        if (!d) throw EngineError(Err::DataIO, "Cannot open file '" + path + ".dat' for reading");
        uint64_t qty = 0;
        rd(d, qty);
        for (uint64_t i = 0; i < qty; ++i) {
            uint64_t osz = 0;
            rd(d, osz);
            std::vector<char> buf(osz);
            d.read(buf.data(), (std::streamsize)osz);
            if (!d || osz < 16) throw EngineError(Err::DataIO, "corrupt .dat file");
            int32_t id;
            uint64_t dl;
            std::memcpy(&id, &buf[0], 4);
            std::memcpy(&dl, &buf[8], 8);
            if (i == 0) dat_len = dl;
            if (dl != dat_len || dl + 16 != osz) throw EngineError(Err::DataIO, "ragged .dat file");
            dat_ids.push_back(id);
            dat_payload.insert(dat_payload.end(), buf.begin() + 16, buf.end());
        }
    }
    if (flag == 1) {
        uint32_t n;
        uint64_t mem_per_obj, off_l0, off_data, maxM, maxM0, search_method;
        int32_t maxlevel, dist_func;
        uint32_t enterpoint;
        rd(in, n); rd(in, mem_per_obj); rd(in, off_l0); rd(in, off_data); rd(in, maxlevel); rd(in, enterpoint);
        rd(in, maxM); rd(in, maxM0); rd(in, dist_func); rd(in, search_method);
        const char* space = nullptr;
        switch (dist_func) {  // DistFuncType, hnsw.h:50-58
            case 1: case 2: space = "l2"; break;
            case 3: space = "cosinesimil"; break;
            case 4: space = "negdotprod"; break;
            case 5: space = "l1"; break;
            case 6: space = "linf"; break;
            default: throw EngineError(Err::DataIO, "Unknown distance function code in index file");
        }
        if (data_type != 0) throw EngineError(Err::DataIO, "an optimized HNSW index holds dense float vectors");
        if (off_l0 < off_data + 16 || mem_per_obj < off_l0 + (maxM0 + 1) * 4 || maxM0 > 8192 || maxM > 4096)
            throw EngineError(Err::DataIO, "unsupported optimized index geometry");
        e.reset(new Engine(space, "hnsw", data_type, dist_type));
        const size_t dim = (off_l0 - off_data - 16) / 4;
        e->dim_ = dim;
        HostGraph& g = e->graph_;
        g.n = (int)n;
        g.maxM = (int)maxM;
        g.M = (int)maxM;
        g.maxM0 = (int)maxM0;
        g.maxlevel = maxlevel;
        g.enterpoint = (int)enterpoint;
        g.levels.assign(n, 0);
        g.links0.assign((size_t)n * (maxM0 + 1), 0);
        g.up_off.assign(n, -1);
        e->graph_rows_.resize((size_t)n * dim);
        e->ids_.resize(n);
        std::vector<char> rec(mem_per_obj);
        for (uint32_t i = 0; i < n; ++i) {
            in.read(rec.data(), (std::streamsize)mem_per_obj);
            if (!in) throw EngineError(Err::DataIO, "truncated index file");
            std::memcpy(&e->ids_[i], &rec[off_data], 4);  // Object header id (object.h:61-77)
            std::memcpy(&e->graph_rows_[(size_t)i * dim], &rec[off_data + 16], dim * 4);
            int32_t cnt;
            std::memcpy(&cnt, &rec[off_l0], 4);
            if (cnt < 0 || (uint64_t)cnt > maxM0) throw EngineError(Err::DataIO, "corrupt level-0 list");
            std::memcpy(&g.links0[(size_t)i * (maxM0 + 1)], &rec[off_l0], (size_t)(cnt + 1) * 4);
        }
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t bytes;
            rd(in, bytes);
            if (!bytes) continue;
            const size_t ints = bytes / 4;
            g.levels[i] = (int)(ints / (maxM + 1));  // the level is not stored: SURVEY.md 8f N1
            g.up_off[i] = (int64_t)g.up_links.size();
            const size_t at = g.up_links.size();
            g.up_links.resize(at + ints);
            in.read(reinterpret_cast<char*>(&g.up_links[at]), bytes);
            if (!in) throw EngineError(Err::DataIO, "truncated index file");
            for (size_t l = 0; l < ints / (maxM + 1); ++l) {  // zero the uninitialised tail slots
                int32_t* L = &g.up_links[at + l * (maxM + 1)];
                for (uint64_t j = (uint64_t)L[0] + 1; j <= maxM; ++j) L[j] = 0;
            }
        }
        e->bp_.M = e->bp_.maxM = (int)maxM;
        e->bp_.maxM0 = (int)maxM0;
        e->method_ = Method::Hnsw;
        if (load_data && !dat_ids.empty()) {
            if (dat_ids.size() != n || dat_len != dim * 4) throw EngineError(Err::DataIO, ".dat does not match the index");
            e->rows_f32_.resize((size_t)n * dim);
            std::memcpy(e->rows_f32_.data(), dat_payload.data(), dat_payload.size());
        } else {
            e->rows_f32_ = e->graph_rows_;  // rows as stored in the index (cosine: normalised)
        }
    } else {
        // regular index: graph only, rows must come from .dat (LoadRegularIndexBin, hnsw.cc:941-991)
        if (!load_data || dat_ids.empty()) throw EngineError(Err::DataIO, "a regular HNSW index needs its .dat file");
        uint32_t n, enterpoint;
        int32_t maxlevel;
        uint64_t M, maxM, maxM0;
        rd(in, n); rd(in, maxlevel); rd(in, enterpoint); rd(in, M); rd(in, maxM); rd(in, maxM0);
        if (n != dat_ids.size()) throw EngineError(Err::DataIO, "The number of stored elements doesn't match the data");
        const bool u8 = data_type == 2;
        // like the reference, a float index without further information is taken to be "l2"
        e.reset(new Engine(u8 ? "l2sqr_sift" : (saved_space.empty() ? "l2" : saved_space.c_str()), "hnsw", data_type,
                           dist_type));
        if (u8) {
            if (dat_len != 132) throw EngineError(Err::DataIO, "unexpected SIFT payload size");
            e->dim_ = 128;
            e->rows_u8_.resize((size_t)n * 128);
            for (uint32_t i = 0; i < n; ++i) std::memcpy(&e->rows_u8_[(size_t)i * 128], &dat_payload[(size_t)i * 132], 128);
        } else {
            e->dim_ = dat_len / 4;
            e->rows_f32_.resize((size_t)n * e->dim_);
            std::memcpy(e->rows_f32_.data(), dat_payload.data(), dat_payload.size());
        }
        e->ids_ = dat_ids;
        HostGraph& g = e->graph_;
        g.n = (int)n;
        g.M = (int)M;
        g.maxM = (int)maxM;
        g.maxM0 = (int)maxM0;
        g.maxlevel = maxlevel;
        g.enterpoint = (int)enterpoint;
        g.levels.assign(n, 0);
        g.links0.assign((size_t)n * (maxM0 + 1), 0);
        g.up_off.assign(n, -1);
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t lvl;
            rd(in, lvl);
            g.levels[i] = (int)lvl;
            if (lvl) {
                g.up_off[i] = (int64_t)g.up_links.size();
                g.up_links.resize(g.up_links.size() + (size_t)lvl * (maxM + 1), 0);
            }
            for (uint32_t l = 0; l <= lvl; ++l) {
                uint32_t qty;
                rd(in, qty);
                int32_t* L = l == 0 ? &g.links0[(size_t)i * (maxM0 + 1)]
                                    : &g.up_links[g.up_off[i] + (size_t)(l - 1) * (maxM + 1)];
                if (qty > (l == 0 ? maxM0 : maxM)) throw EngineError(Err::DataIO, "corrupt adjacency list");
                L[0] = (int32_t)qty;
                in.read(reinterpret_cast<char*>(L + 1), (std::streamsize)qty * 4);
                if (!in) throw EngineError(Err::DataIO, "truncated index file");
            }
        }
        e->bp_.M = (int)M;
        e->bp_.maxM = (int)maxM;
        e->bp_.maxM0 = (int)maxM0;
        e->bp_.skip_optimized = true;
        e->method_ = Method::Hnsw;
        e->graph_rows_.clear();
    }
    e->loaded_graph_ = true;
    e->created_ = true;
    e->dirty_ = true;
    e->graph_dirty_ = false;
    e->ef_ = 200;
    return e;
}

}  // namespace gfxknn
