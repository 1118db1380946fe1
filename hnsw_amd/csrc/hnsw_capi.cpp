// extern "C" layer of libhnsw_amd.so: include/hnsw_amd.h implemented over HostIndex
// (construction, files) and DeviceIndex (HBM copy + HIP search kernels).
#include <hip/hip_runtime.h>

#include <cstring>
#include <ios>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/hnsw_amd.h"
#include "hnsw_device.h"
#include "hnsw_host.h"

using namespace hnsw_amd;

namespace hnsw_amd {
void hdf_get_dimensions(const char* filename, const char* dataset, uint64_t dims[2]);
void hdf_read(const char* filename, const char* dataset, void* data, bool as_float);
float bruteforce_knn(const float* d_base, uint32_t n, const float* d_queries, uint32_t nq, uint32_t d, uint32_t k,
                     uint32_t id_offset, uint32_t* d_out_ids, float* d_out_dist, void* stream_v);
void merge_topk(const float* d_dist, const uint32_t* d_ids, const uint32_t* shard_offset, uint32_t S, uint32_t nq, uint32_t k,
                uint32_t out_k, size_t shard_stride, uint32_t* d_out_ids, float* d_out_dist, void* stream);
}

struct hnsw_amd_index {
    HostIndex host;
    DeviceIndex dev;
    DeviceBuild build;       // layer 0 of a device-linked build, while one runs
    int device = 0;
    FlatGraph exported;
    // grow-only staging of the host-pointer search entry points: queries and the packed results
    // (ids | distances | counts | counters) on the device, and pinned host mirrors of both, so that a call is
    // two asynchronous copies around the kernel and one wait
    void* d_q = nullptr; void* h_q = nullptr; size_t q_bytes = 0;
    void* d_out = nullptr; void* h_out = nullptr; size_t out_bytes = 0;
    hnsw_amd_index(int M, int Mmax, int efC) : host(M, Mmax, efC) {}
};

static thread_local std::string g_err;

template <class F>
static int guarded(F&& f)
{
    try {
        f();
        return HNSW_AMD_OK;
    } catch (const std::ios_base::failure& e) { g_err = e.what(); return HNSW_AMD_E_IO;
    } catch (const std::domain_error& e) { g_err = e.what(); return HNSW_AMD_E_GRAPH;
    } catch (const std::length_error& e) { g_err = e.what(); return HNSW_AMD_E_SCRATCH;
    } catch (const std::invalid_argument& e) { g_err = e.what(); return HNSW_AMD_E_ARG;
    } catch (const std::logic_error& e) { g_err = e.what(); return HNSW_AMD_E_ARG;
    } catch (const std::runtime_error& e) {
        g_err = e.what();
        if (g_err.rfind("HIP:", 0) == 0) return HNSW_AMD_E_HIP;
        if (g_err.rfind("HNSW: exceeded", 0) == 0) return HNSW_AMD_E_FULL;
        return HNSW_AMD_E_ARG;
    } catch (const std::bad_alloc&) { g_err = "out of host memory"; return HNSW_AMD_E_ARG;
    } catch (...) { g_err = "unknown error"; return HNSW_AMD_E_ARG; }
}

#define REQUIRE(cond, msg) do { if (!(cond)) throw std::invalid_argument(msg); } while (0)
#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) throw std::runtime_error(std::string("HIP: ") + #expr + ": " + hipGetErrorString(e_)); } while (0)

// The device holds this index's own rows (same table contents, same count): only the graph can have changed.
static bool device_has_rows(const hnsw_amd_index* h)
{
    return h->dev.device == h->device && h->dev.g.vectors && h->dev.rows_epoch == h->host.rows_epoch_ &&
           h->dev.g.n == h->host.size() && h->dev.g.d == h->host.vector_size_;
}

static void ensure_device_copy(hnsw_amd_index* h, uint32_t flags = 0)
{
    if (h->host.build_failed_) throw std::logic_error("search: an earlier concurrent build failed half way; init() the index again");
    if (h->dev.device != h->device || h->dev.version != h->host.version_) h->dev.upload(h->host, h->device);
    if (flags & HNSW_AMD_ROWS_Q8) h->dev.ensure_quantized(h->host);
}

extern "C" {

const char* hnsw_amd_last_error(void) { return g_err.c_str(); }

int hnsw_amd_create(int M, int Mmax, int efConstruction, hnsw_amd_index** out)
{
    return guarded([&] {
        REQUIRE(out != nullptr, "create: out is NULL");
        REQUIRE(M >= 2 && Mmax >= M && Mmax <= 32, "create: need 2 <= M <= Mmax <= 32");
        REQUIRE(efConstruction >= 1, "create: efConstruction must be positive");
        *out = new hnsw_amd_index(M, Mmax, efConstruction);
    });
}

void hnsw_amd_destroy(hnsw_amd_index* h)
{
    if (!h) return;
    if (h->dev.device >= 0) (void)hipSetDevice(h->dev.device);
    h->build.release();
    for (void* p : {h->d_q, h->d_out}) if (p) (void)hipFree(p);
    for (void* p : {h->h_q, h->h_out}) if (p) (void)hipHostFree(p);
    delete h;
}

int hnsw_amd_init(hnsw_amd_index* h, uint32_t vector_size, uint32_t max_node_count)
{
    return guarded([&] {
        REQUIRE(h, "init: index is NULL");
        REQUIRE(vector_size >= 1 && vector_size <= 65536, "init: vector_size out of range");
        REQUIRE(max_node_count >= 1 && max_node_count < 0x7FFFFFFFu, "init: max_node_count out of range");
        h->host.init(vector_size, max_node_count);
    });
}

int hnsw_amd_insert(hnsw_amd_index* h, const float* q)
{
    return guarded([&] { REQUIRE(h && q, "insert: NULL argument"); h->host.insert(q); });
}

int hnsw_amd_insert_batch(hnsw_amd_index* h, const float* rows, uint32_t n, int threads)
{
    return guarded([&] { REQUIRE(h && (rows || n == 0), "insert_batch: NULL argument"); h->host.insert_batch(rows, n, threads); });
}

// Candidate lists for a sub-batch of a GPU-assisted build: bring the device's copy of the graph up to date (all of it
// when the index grew or is not there yet, lists and upper layers otherwise) and run the batched search on it.
static void assisted_candidates(void* ctx, const float* queries, uint32_t nq, int ef, uint32_t* ids, float* dist, uint32_t* cnt)
{
    hnsw_amd_index* h = static_cast<hnsw_amd_index*>(ctx);
    if (device_has_rows(h)) h->dev.refresh_graph(h->host);
    else h->dev.upload(h->host, h->device);
    const uint32_t flags = (h->host.dist_scalar_ ? HNSW_AMD_DIST_SCALAR : HNSW_AMD_DIST_SSE) | (h->host.build_float_keys_ ? HNSW_AMD_KEYS_FLOAT : HNSW_AMD_KEYS_INT32);
    const int rc = hnsw_amd_search_batch(h, queries, nq, ef, (uint32_t)ef, flags, ids, dist, cnt, nullptr);
    if (rc != HNSW_AMD_OK) throw std::runtime_error(g_err);
}

int hnsw_amd_insert_batch_assisted(hnsw_amd_index* h, const float* rows, uint32_t n, int threads, int device)
{
    return guarded([&] {
        REQUIRE(h && (rows || n == 0), "insert_batch_assisted: NULL argument");
        int count = 0;
        HIP_OK(hipGetDeviceCount(&count));
        REQUIRE(device >= 0 && device < count, "insert_batch_assisted: no such GPU");
        h->device = device;
        h->host.insert_batch(rows, n, threads, assisted_candidates, h);
        // the device holds every row already: bring its graph to the final state instead of leaving a full upload to
        // the first search
        if (device_has_rows(h) && h->dev.version != h->host.version_) h->dev.refresh_graph(h->host);
    });
}

// Layer 0 of the build on the device (hnsw_build.hip): the three callbacks of HostIndex::Layer0Linker.
static uint32_t build_flags(const hnsw_amd_index* h)
{
    uint32_t flags = (h->host.dist_scalar_ ? HNSW_AMD_DIST_SCALAR : HNSW_AMD_DIST_SSE) | (h->host.build_float_keys_ ? HNSW_AMD_KEYS_FLOAT : HNSW_AMD_KEYS_INT32);
    // integer-valued rows (an exact byte table exists): the construction searches read it — the same distances bit for
    // bit, so the same graph, from a quarter of the bytes
    if (!h->host.dist_scalar_ && h->dev.g.vectors_u8 != nullptr) flags |= HNSW_AMD_ROWS_U8;
    return flags;
}

static void linked_from_host(void* ctx)
{
    hnsw_amd_index* h = static_cast<hnsw_amd_index*>(ctx);
    if (device_has_rows(h)) h->dev.refresh_graph(h->host);
    else h->dev.upload(h->host, h->device);
    h->build.from_host(h->dev, h->host);
}

static void linked_upper_layers(void* ctx)
{
    hnsw_amd_index* h = static_cast<hnsw_amd_index*>(ctx);
    FlatGraph upper;
    h->dev.refresh_graph(h->host, false, &upper);
    h->build.upper_view(h->dev, upper);
}

static void linked_step(void* ctx, uint32_t first_id, uint32_t cnt, const uint32_t* upper_ids, uint32_t n_upper,
                        uint32_t* cand_ids, float* cand_dist, uint32_t* cand_cnt)
{
    hnsw_amd_index* h = static_cast<hnsw_amd_index*>(ctx);
    h->build.step(h->dev, h->host, first_id, cnt, build_flags(h), upper_ids, n_upper, cand_ids, cand_dist, cand_cnt);
}

static void linked_to_host(void* ctx)
{
    hnsw_amd_index* h = static_cast<hnsw_amd_index*>(ctx);
    h->build.to_host(h->dev, h->host);
}

int hnsw_amd_insert_batch_linked(hnsw_amd_index* h, const float* rows, uint32_t n, int threads, int device)
{
    if (h && !DeviceBuild::supports(h->host)) return hnsw_amd_insert_batch_assisted(h, rows, n, threads, device);
    return guarded([&] {
        REQUIRE(h && (rows || n == 0), "insert_batch_linked: NULL argument");
        int count = 0;
        HIP_OK(hipGetDeviceCount(&count));
        REQUIRE(device >= 0 && device < count, "insert_batch_linked: no such GPU");
        h->device = device;
        const HostIndex::Layer0Linker linker{h, linked_from_host, linked_upper_layers, linked_step, linked_to_host};
        try {
            h->host.insert_batch(rows, n, threads, nullptr, nullptr, &linker);
        } catch (...) {
            h->build.release();
            throw;
        }
        h->build.release();
        // the device holds every row and layer 0 already: upper layers and entry points to their final state
        if (device_has_rows(h) && h->dev.version != h->host.version_) h->dev.refresh_graph(h->host);
    });
}

int hnsw_amd_set_build_float_keys(hnsw_amd_index* h, int on)
{
    return guarded([&] { REQUIRE(h, "set_build_float_keys: index is NULL"); h->host.build_float_keys_ = on != 0; });
}

int hnsw_amd_set_build_distance(hnsw_amd_index* h, uint32_t dist_flag)
{
    return guarded([&] {
        REQUIRE(h, "set_build_distance: index is NULL");
        REQUIRE(dist_flag == HNSW_AMD_DIST_SSE || dist_flag == HNSW_AMD_DIST_SCALAR, "set_build_distance: HNSW_AMD_DIST_SSE or HNSW_AMD_DIST_SCALAR");
        h->host.dist_scalar_ = dist_flag == HNSW_AMD_DIST_SCALAR;
    });
}

int hnsw_amd_sync_device(hnsw_amd_index* h, int device)
{
    return guarded([&] {
        REQUIRE(h, "sync_device: index is NULL");
        int count = 0;
        HIP_OK(hipGetDeviceCount(&count));
        REQUIRE(device >= 0 && device < count, "sync_device: no such GPU");
        h->device = device;
        ensure_device_copy(h);
    });
}

int hnsw_amd_search_batch_device(hnsw_amd_index* h, const float* d_queries, uint32_t nq, int ef, uint32_t out_k,
                                 uint32_t flags, uint32_t* d_out_ids, float* d_out_dist, uint32_t* d_out_count,
                                 hnsw_amd_query_stats* d_stats, void* stream, uint32_t* retried)
{
    return guarded([&] {
        REQUIRE(h && d_queries && d_out_ids && d_out_dist, "search_batch_device: NULL argument");
        ensure_device_copy(h, flags);
        uint32_t r = h->dev.search(d_queries, nq, ef, out_k, flags, d_out_ids, d_out_dist, d_out_count, d_stats, stream);
        if (retried) *retried = r;
    });
}

int hnsw_amd_search_batch_device_begin(hnsw_amd_index* h, const float* d_queries, uint32_t nq, int ef, uint32_t out_k,
                                       uint32_t flags, uint32_t* d_out_ids, float* d_out_dist, uint32_t* d_out_count,
                                       hnsw_amd_query_stats* d_stats, void* stream, hnsw_amd_search_call** call)
{
    return guarded([&] {
        REQUIRE(h && d_queries && d_out_ids && d_out_dist && call, "search_batch_device_begin: NULL argument");
        *call = nullptr;
        ensure_device_copy(h, flags);
        *call = reinterpret_cast<hnsw_amd_search_call*>(h->dev.begin(d_queries, nq, ef, out_k, flags, d_out_ids, d_out_dist, d_out_count, d_stats, stream));
    });
}

int hnsw_amd_search_batch_device_finish(hnsw_amd_index* h, hnsw_amd_search_call* call, uint32_t* retried)
{
    return guarded([&] {
        REQUIRE(h && call, "search_batch_device_finish: NULL argument");
        uint32_t r = h->dev.finish(reinterpret_cast<SearchCall*>(call));
        if (retried) *retried = r;
    });
}

int hnsw_amd_reload_env(hnsw_amd_index* h)
{
    return guarded([&] { REQUIRE(h, "reload_env: index is NULL"); h->dev.tune.read(); });
}

int hnsw_amd_search_batch_device_noalloc(hnsw_amd_index* h, const float* d_queries, uint32_t nq, int ef, uint32_t out_k,
                                         uint32_t flags, uint32_t* d_out_ids, float* d_out_dist, uint32_t* d_out_count,
                                         hnsw_amd_query_stats* d_stats, void* stream, uint32_t* retried)
{
    return guarded([&] {
        REQUIRE(h && d_queries && d_out_ids && d_out_dist, "search_batch_device_noalloc: NULL argument");
        // everything that allocates (the HBM copy of the index, the quantised table) must have happened before
        REQUIRE(h->dev.device == h->device && h->dev.version == h->host.version_,
                "search_batch_device_noalloc: the index is not on the device (or changed since): call hnsw_amd_sync_device first");
        REQUIRE(!(flags & HNSW_AMD_ROWS_Q8) || h->dev.g.vectors_q8 != nullptr,
                "search_batch_device_noalloc: HNSW_AMD_ROWS_Q8 needs the quantised table, which an allocating search call builds on first use");
        uint32_t r = h->dev.search(d_queries, nq, ef, out_k, flags, d_out_ids, d_out_dist, d_out_count, d_stats, stream, true);
        if (retried) *retried = r;
    });
}

int hnsw_amd_search_batch(hnsw_amd_index* h, const float* queries, uint32_t nq, int ef, uint32_t out_k, uint32_t flags,
                          uint32_t* out_ids, float* out_dist, uint32_t* out_count, hnsw_amd_query_stats* stats)
{
    return guarded([&] {
        REQUIRE(h && (queries || nq == 0) && out_ids && out_dist, "search_batch: NULL argument");
        if (nq == 0) return;
        ensure_device_copy(h, flags);
        HIP_OK(hipSetDevice(h->device));
        const size_t qb = (size_t)nq * h->host.vector_size_ * sizeof(float);
        const size_t oe = (size_t)nq * out_k;
        const size_t ob = (2 * oe + (size_t)nq) * 4 + (size_t)nq * sizeof(hnsw_amd_query_stats);
        if (qb > h->q_bytes) {
            if (h->d_q) HIP_OK(hipFree(h->d_q));
            if (h->h_q) HIP_OK(hipHostFree(h->h_q));
            h->d_q = h->h_q = nullptr; h->q_bytes = 0;
            HIP_OK(hipMalloc(&h->d_q, qb)); HIP_OK(hipHostMalloc(&h->h_q, qb, hipHostMallocDefault)); h->q_bytes = qb;
        }
        if (ob > h->out_bytes) {
            if (h->d_out) HIP_OK(hipFree(h->d_out));
            if (h->h_out) HIP_OK(hipHostFree(h->h_out));
            h->d_out = h->h_out = nullptr; h->out_bytes = 0;
            HIP_OK(hipMalloc(&h->d_out, ob)); HIP_OK(hipHostMalloc(&h->h_out, ob, hipHostMallocDefault)); h->out_bytes = ob;
        }
        std::memcpy(h->h_q, queries, qb);
        // A handful of queries (the reference's own loop is one query per call, src/main.cpp:291-294): the kernel reads them
        // from, and writes its results to, the pinned host buffers themselves — two copy-engine transfers less per call,
        // which is most of what such a call costs besides the search.  Larger batches are staged through device memory.
        const bool direct = nq <= 16;
        void* q_src = h->d_q;
        void* out_dst = h->d_out;
        if (direct) {
            HIP_OK(hipHostGetDevicePointer(&q_src, h->h_q, 0));
            HIP_OK(hipHostGetDevicePointer(&out_dst, h->h_out, 0));
        } else {
            HIP_OK(hipMemcpyAsync(h->d_q, h->h_q, qb, hipMemcpyHostToDevice, nullptr));
        }
        uint32_t* d_ids = static_cast<uint32_t*>(out_dst);
        float* d_dist = reinterpret_cast<float*>(d_ids + oe);
        uint32_t* d_cnt = d_ids + 2 * oe;
        hnsw_amd_query_stats* d_stats = reinterpret_cast<hnsw_amd_query_stats*>(d_cnt + nq);
        // the results travel back behind every launch of the call, ahead of the wait that learns whether a launch
        // has to be repeated for some queries (rare); the last copy is the one that counts
        const DeviceIndex::CopyBack back{h->h_out, h->d_out, stats ? ob : ob - (size_t)nq * sizeof(hnsw_amd_query_stats)};
        h->dev.search((const float*)q_src, nq, ef, out_k, flags, d_ids, d_dist, d_cnt, stats ? d_stats : nullptr, nullptr, false,
                      direct ? nullptr : &back);
        const uint32_t* r = static_cast<const uint32_t*>(h->h_out);
        std::memcpy(out_ids, r, oe * 4);
        std::memcpy(out_dist, r + oe, oe * 4);
        if (out_count) std::memcpy(out_count, r + 2 * oe, (size_t)nq * 4);
        if (stats) std::memcpy(stats, r + 2 * oe + nq, (size_t)nq * sizeof(hnsw_amd_query_stats));
    });
}

int hnsw_amd_aproximate_knn(hnsw_amd_index* h, const float* q, int k, int ef, uint32_t flags, uint32_t* out_ids,
                            float* out_dist, uint32_t* out_count)
{
    (void)k;   // ignored by the reference as well (src/hnsw.cpp:373: W_ keeps up to ef entries)
    return hnsw_amd_search_batch(h, q, 1, ef, (uint32_t)(ef > 0 ? ef : 1), flags, out_ids, out_dist, out_count, nullptr);
}

int hnsw_amd_save_graph(hnsw_amd_index* h, const char* filename)
{
    return guarded([&] { REQUIRE(h && filename, "save_graph: NULL argument"); h->host.save_graph(filename); });
}

int hnsw_amd_load_graph(hnsw_amd_index* h, const char* filename)
{
    return guarded([&] { REQUIRE(h && filename, "load_graph: NULL argument"); h->host.load_graph(filename); });
}

int hnsw_amd_import_flat(hnsw_amd_index* h, const hnsw_amd_flat_graph* g, const float* vectors)
{
    return guarded([&] {
        REQUIRE(h && g && vectors, "import_flat: NULL argument");
        REQUIRE(g->nlayers >= 1 && g->nlayers <= 64 && g->n >= 1 && g->d >= 1, "import_flat: empty graph");
        REQUIRE(g->deg0 >= 1 && g->deg0 <= 64 && g->degu >= 1 && g->degu <= 64, "import_flat: degree out of range");
        FlatGraph f;
        f.n = g->n; f.d = g->d; f.nlayers = g->nlayers; f.deg0 = g->deg0; f.degu = g->degu;
        f.up_off.assign(g->up_off, g->up_off + g->nlayers + 1);
        REQUIRE(f.up_off[0] == 0 && f.up_off[1] == 0, "import_flat: up_off must start 0,0");
        for (uint32_t l = 1; l < g->nlayers; l++) REQUIRE(f.up_off[l + 1] > f.up_off[l], "import_flat: up_off not increasing");
        const uint32_t U = f.up_off[g->nlayers];
        f.adj0.assign(g->adj0, g->adj0 + (size_t)g->n * g->deg0);
        f.cnt0.assign(g->cnt0, g->cnt0 + g->n);
        f.up_order.assign(g->up_order, g->up_order + U);
        f.up_down.assign(g->up_down, g->up_down + U);
        f.up_cnt.assign(g->up_cnt, g->up_cnt + U);
        f.up_adj_slot.assign(g->up_adj_slot, g->up_adj_slot + (size_t)U * g->degu);
        f.up_adj_order.assign(g->up_adj_order, g->up_adj_order + (size_t)U * g->degu);
        f.ep_slot.assign(g->ep_slot, g->ep_slot + g->nlayers);
        f.ep_order.assign(g->ep_order, g->ep_order + g->nlayers);
        try { h->host.import_flat(f, vectors); }
        catch (const std::invalid_argument& e) { g_err = e.what(); throw std::domain_error(e.what()); }
    });
}

int hnsw_amd_export_flat(hnsw_amd_index* h, hnsw_amd_flat_graph* g, const float** vectors)
{
    return guarded([&] {
        REQUIRE(h && g, "export_flat: NULL argument");
        h->host.export_flat(h->exported);
        const FlatGraph& f = h->exported;
        g->n = f.n; g->d = f.d; g->nlayers = f.nlayers; g->deg0 = f.deg0; g->degu = f.degu;
        g->adj0 = f.adj0.data(); g->cnt0 = f.cnt0.data(); g->up_off = f.up_off.data(); g->up_order = f.up_order.data();
        g->up_down = f.up_down.data(); g->up_cnt = f.up_cnt.data(); g->up_adj_slot = f.up_adj_slot.data();
        g->up_adj_order = f.up_adj_order.data(); g->ep_slot = f.ep_slot.data(); g->ep_order = f.ep_order.data();
        if (vectors) *vectors = h->host.vector_data_.data();
    });
}

int hnsw_amd_hdf_get_dimensions(const char* filename, const char* dataset, uint64_t dims[2])
{
    return guarded([&] { REQUIRE(filename && dataset && dims, "hdf_get_dimensions: NULL argument"); hdf_get_dimensions(filename, dataset, dims); });
}

int hnsw_amd_hdf_read_float(const char* filename, const char* dataset, float* data)
{
    return guarded([&] { REQUIRE(filename && dataset && data, "hdf_read_float: NULL argument"); hdf_read(filename, dataset, data, true); });
}

int hnsw_amd_hdf_read_int(const char* filename, const char* dataset, int* data)
{
    return guarded([&] { REQUIRE(filename && dataset && data, "hdf_read_int: NULL argument"); hdf_read(filename, dataset, data, false); });
}

int hnsw_amd_bruteforce_knn_device(const float* d_base, uint32_t n, const float* d_queries, uint32_t nq, uint32_t d, uint32_t k,
                                   uint32_t id_offset, uint32_t* d_out_ids, float* d_out_dist, void* stream, float* mfma_ms)
{
    return guarded([&] {
        float ms = bruteforce_knn(d_base, n, d_queries, nq, d, k, id_offset, d_out_ids, d_out_dist, stream);
        if (mfma_ms) *mfma_ms = ms;
    });
}

int hnsw_amd_merge_topk_device(const float* d_dist, const uint32_t* d_ids, const uint32_t* shard_offset, uint32_t n_shards,
                               uint32_t nq, uint32_t k, uint32_t out_k, uint64_t shard_stride, uint32_t* d_out_ids, float* d_out_dist,
                               void* stream)
{
    return guarded([&] {
        REQUIRE(d_dist && d_ids && shard_offset && d_out_ids && d_out_dist, "merge_topk_device: NULL argument");
        merge_topk(d_dist, d_ids, shard_offset, n_shards, nq, k, out_k, (size_t)shard_stride, d_out_ids, d_out_dist, stream);
    });
}

int hnsw_amd_gather_ceiling_device(hnsw_amd_index* h, uint32_t n_waves, uint32_t rows_per_wave, uint32_t lds_bytes_per_wave, uint32_t dist_flag,
                                   void* stream, float* ms, uint64_t* bytes)
{
    return guarded([&] {
        REQUIRE(h && ms && bytes, "gather_ceiling_device: NULL argument");
        REQUIRE(dist_flag == HNSW_AMD_DIST_SSE || dist_flag == HNSW_AMD_DIST_SCALAR, "gather_ceiling_device: HNSW_AMD_DIST_SSE or HNSW_AMD_DIST_SCALAR");
        ensure_device_copy(h);
        *ms = h->dev.gather_ceiling(n_waves, rows_per_wave, lds_bytes_per_wave, stream, *bytes, dist_flag == HNSW_AMD_DIST_SCALAR);
    });
}

int hnsw_amd_last_launch_info(hnsw_amd_index* h, uint32_t* lds_bytes_per_query, uint32_t* resident_queries_per_cu, uint32_t* visited_form)
{
    return guarded([&] {
        REQUIRE(h, "last_launch_info: index is NULL");
        uint32_t lds = 0, occ = 0, form = 0;
        h->dev.last_launch(lds, occ, form);
        if (lds_bytes_per_query) *lds_bytes_per_query = lds;
        if (resident_queries_per_cu) *resident_queries_per_cu = occ;
        if (visited_form) *visited_form = form;
    });
}

int hnsw_amd_vector_data(hnsw_amd_index* h, const float** out)
{
    return guarded([&] {
        REQUIRE(h && out, "vector_data: NULL argument");
        *out = h->host.vector_data_.empty() ? nullptr : h->host.vector_data_.data();
    });
}

int hnsw_amd_get_info(hnsw_amd_index* h, int32_t* M, int32_t* Mmax, int32_t* Mmax0, int32_t* efConstruction,
                      uint32_t* vector_size, int32_t* max_node_count, int32_t* actual_node_count, uint32_t* nlayers)
{
    return guarded([&] {
        REQUIRE(h, "get_info: index is NULL");
        if (M) *M = h->host.M_;
        if (Mmax) *Mmax = h->host.Mmax_;
        if (Mmax0) *Mmax0 = h->host.Mmax0_;
        if (efConstruction) *efConstruction = h->host.efConstruction_;
        if (vector_size) *vector_size = h->host.vector_size_;
        if (max_node_count) *max_node_count = h->host.max_node_count_;
        if (actual_node_count) *actual_node_count = h->host.actual_node_count_;
        if (nlayers) *nlayers = (uint32_t)h->host.layers_.size();
    });
}

}  // extern "C"
