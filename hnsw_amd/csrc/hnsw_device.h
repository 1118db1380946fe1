// Device-side (HBM) form of the index and the launch interface of the search kernels.
#pragma once

#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

#include "../../include/hnsw_amd.h"
#include "hnsw_host.h"

namespace hnsw_amd {

// What a kernel needs to walk the graph; passed by value.  All pointers are device memory.
struct DeviceGraph {
    uint32_t n, d;
    uint32_t row_stride;        // floats per stored row: d rounded up to a multiple of 32 (128-byte lines), zero filled
    uint32_t nlayers, deg0, degu;
    uint32_t top_ep_slot, top_ep_order;
    const float* vectors;       // n x row_stride floats, the dims of every 128-byte line regrouped by lane sum
    const uint8_t* vectors_u8;  // n x row_stride_u8 bytes, one per dim, or nullptr when some value is not an integer in 0..255
    uint32_t row_stride_u8;     // d rounded up to a multiple of 128
    const uint8_t* vectors_q8;  // same shape: byte = clamp(round((v - q8_lo) * q8_inv_step), 0, 255); nullptr until asked for
    float q8_lo, q8_inv_step;
    float max_row_norm;         // an upper bound of the Euclidean length of every row
    const uint32_t* adj0;       // n x deg0, 0xFFFFFFFF padded (the live entries are a prefix)
    const uint32_t* up_off;     // nlayers+1
    const uint32_t* up_down;    // U
    const uint32_t* up_adj_slot;   // U x degu
    const uint32_t* up_adj_order;  // U x degu
};

struct SearchArgs {
    DeviceGraph g;
    const float* queries;     // nq x d (unpadded rows)
    const uint32_t* qmap;     // optional: block -> query index (retry launches); nullptr = identity
    uint32_t nq_launch;       // blocks in this launch
    uint32_t ef, out_k, flags;
    uint32_t hash_size, cap_c, cap_w;
    uint32_t vis_rb;          // 0xFFFFFFFF: visited slots of 32 bits; else 16-bit slots in 2^vis_rb regions   // cap_w: ef+1 rounded up to even
    uint32_t vis_chunks;      // 16-bit bucket form: 16-byte chunks (8 slots) per region; 0: a probing form
    uint32_t ovf_size;        // bucket form: 32-bit slots of the overflow table behind the primary one
    uint32_t vis_regions, vis_magic;   // bucket form: number of regions (any), floor(2^32 / regions)
    uint32_t* spill_pool;     // bucket form: spill tables in device memory for queries that outgrow the overflow table
    uint32_t spill_tables, spill_slots;   // how many, and 32-bit slots each (0 tables: such queries are re-run instead)
    uint32_t qpad;            // floats of the query copy in LDS (multiple of 64)
    uint32_t c_words;         // dwords per entry of the candidate heap: 2, or 1 (key and node_order packed)
    uint32_t q_overlay;       // 1: the query's copy is dead once the kernel's registers hold it, and W / C / visited start at its place
    uint32_t* out_ids;        // nq x out_k
    float* out_dist;          // nq x out_k
    uint32_t* out_count;      // nq or nullptr
    hnsw_amd_query_stats* stats;   // nq or nullptr
    uint32_t mail_off;        // two-wavefront form: dword offset of the wavefronts' mailbox in the query's LDS
    uint32_t known_visit, known_c;   // largest visited set / heap room the launcher already knows of: only queries beyond report
    uint32_t* fail_count;     // device words: [0] number of queries whose scratch overflowed in this launch,
                              // [1] largest visited set, [2] most room the candidate heap was asked for, among the queries
                              // that finished, [3] spill tables handed out, [4] sum of the visited counts / 16 and [5] number of a sample of them
    uint32_t* fail_list;      // the overflowed queries' indices
};

size_t search_lds_bytes(const SearchArgs& a, bool w_in_lds);

// Developer switches read from the environment (HNSW_AMD_*; see Tunables::read for each), once per upload of an index
// and on hnsw_amd_reload_env — never inside a search call.
struct Tunables {
    bool generic_rows = false, no_deep_rows = false, visited32 = false, visited16 = false, visitedb = false, no_spill = false, verbose = false;
    bool no_pack_c = false, no_dot4 = false;
    float bulk_frac = 0.30f;
    long want_visit = -1, want_bulk = -1, spill_tables = -1, rb_min = 0, lds_budget = -1, lds_pad = 0;
    void read();
};

// A search call between its begin and its finish: everything the call owns (the retry protocol's two lists of
// overflowed queries, the pinned words its launches report into, its pool of spill tables) and what finish needs to
// know about it.  An index keeps a few of these, so that as many calls can be in flight on it at once.
struct SearchCall {
    bool busy = false;
    uint32_t* d_lists = nullptr;      // two lists of list_cap words each: kFailHeader words of header, then query indices
    uint32_t list_cap = 0;
    uint32_t* h_report = nullptr;     // pinned host mirror of a list's header
    uint32_t* h_report_dev = nullptr; // the same memory as the GPU addresses it
    bool headers_zero = false;        // both list headers are known to be zero (every launch leaves them so)
    bool* headers_zero_flag = nullptr; // that flag, or the one of the library's static array (the entry point that allocates nothing)
    uint32_t* d_spill = nullptr;      // pool of spill tables (bucket form of the visited set), allocated on first need
    // the call
    SearchArgs a{};
    uint32_t nq = 0, flags = 0, retried = 0;
    int ef = 0, tier = 0;
    bool no_alloc = false, launched = false, timed = false, has_copy_back = false;
    bool dot4 = false;                // the current launch reads byte rows through integer dot products
    bool packc = false;               // the current launch keeps candidate-heap entries of four bytes
    void* stream = nullptr;
    uint32_t *cur = nullptr, *other = nullptr;        // the list the current launch appends to, and the other one
    uint32_t want_visit0 = 0, want_c0 = 0, want_bulk0 = 0;
    uint32_t seen_visit = 0, seen_c = 0;
    uint64_t seen_sum16 = 0, seen_done = 0;
    struct { void* dst; const void* src; size_t bytes; } copy_back{};
    void *ev0 = nullptr, *ev1 = nullptr;              // (verbose) events around the kernel
};

struct BucketShape {
    uint32_t n = 0, want = 0, rb_min = 0;     // what it was computed for
    uint32_t regions = 0, chunks = 0, ovf = 0;
    bool ok = false;
};

class DeviceIndex {
public:
    int device = -1;
    uint64_t version = ~0ull;     // HostIndex::version_ this copy was made from
    uint64_t rows_epoch = 0;      // HostIndex::rows_epoch_ of the vector table that was uploaded
    DeviceGraph g{};
    static constexpr int kMaxCallsInFlight = 4;
    bool pack_off = false;            // most queries of an earlier batch could not promise small candidate keys: eight-byte entries
    bool dot4_off = false;            // most queries of an earlier batch were not integer-valued: byte rows go through the float arithmetic
    Tunables tune;
    struct OccupancyNote { const void* kernel; size_t bytes; int blocks; };
    std::vector<OccupancyNote> occupancy_notes;   // resident workgroups per CU the runtime reported, by kernel and LDS size
    uint32_t learn_age = 0;           // batches since the sizes were learned
    int dev_max_lds = 0, dev_num_cus = 0;   // device attributes, looked up once
    // largest visited set and candidate heap the queries of earlier batches reached, and the ef / flags they were
    // observed at: later batches are sized from them
    uint32_t learn_flags = 0;
    int learn_ef = 0;
    uint32_t learn_visit = 0, learn_c = 0;
    uint32_t learn_mean = 0;          // mean visited count of those batches
    struct LearnNote { uint32_t flags; int ef; uint32_t visit, c, mean, age; };
    std::vector<LearnNote> learn_notes;      // what batches with other flags / ef left (at most 16 kinds)
    uint32_t learn_notes_next = 0;
    // last shapes computed for the bucket form of the visited set (the search over shapes is not free): sized for
    // the farthest-reaching query, and for the bulk of the queries
    BucketShape shape_full, shape_bulk;

    ~DeviceIndex();
    void release();
    void upload(const HostIndex& h, int device);
    // Builds the quantised byte table (HNSW_AMD_ROWS_Q8) on first use.
    // lists and upper layers again (or the upper layers only); vectors stay (same rows).  `exported`: the flat form of the
    // upper layers it was made from, for a caller that needs it as well
    void refresh_graph(const HostIndex& h, bool with_layer0 = true, FlatGraph* exported = nullptr);
    // The blocking search on another graph over the same vectors (a build's view of an upper layer as a graph of its own):
    // what batches on the view teach about scratch sizes is kept apart from the index's own (kFlagView in the flags).
    static constexpr uint32_t kFlagView = 0x40000000u;
    uint32_t search_on(const DeviceGraph& view, const float* d_queries, uint32_t nq, int ef, uint32_t out_k, uint32_t flags,
                       uint32_t* d_ids, float* d_dist, uint32_t* d_count, void* stream);
    void ensure_quantized(const HostIndex& h);
    // Runs the batch (device pointers), re-running overflowed queries on larger tiers.
    // Returns the number of re-run queries; throws on HIP errors / scratch exhaustion.
    // no_alloc: the overflow lists live in the library's static device array instead of a buffer of this object
    // (no allocation of any kind inside the call; the array is shared by all calls on the GPU).
    // The first launch of the last search call: dynamic LDS bytes per query, queries a CU holds with that, and the
    // form of the visited set (0: 32-bit slots, 1: 16-bit probing regions, 2: 16-bit bucket regions).
    void last_launch(uint32_t& lds_bytes, uint32_t& resident_per_cu, uint32_t& visited_form) const;
    const void* last_kernel = nullptr;
    uint32_t last_lds, last_block = 0;
    int last_form = 0, last_wps = 4;
    // Diagnostic: milliseconds of one launch in which n_waves wavefronts each read rows_per_wave random rows of the
    // vector table with the search kernel's loads and distance arithmetic and nothing else; `bytes` = what the
    // search's counters would charge for those rows.  lds_bytes per wavefront sets the residency.
    float gather_ceiling(uint32_t n_waves, uint32_t rows_per_wave, uint32_t lds_bytes, void* stream, uint64_t& bytes, bool scalar = false);
    // copy_back: a device-to-host copy (pinned destination) queued behind every launch, before the report.
    typedef decltype(SearchCall::copy_back) CopyBack;
    // begin: sizes the batch's scratch from what is known, launches it on `stream` and queues the launch's report; returns
    // without waiting.  finish: waits for the report, re-runs the queries that outgrew their scratch on larger tables
    // (waiting for each such launch), remembers what the batch needed; returns the number of re-run queries.  Up to
    // kMaxCallsInFlight calls may be between begin and finish on one index at a time; they finish in any order.
    SearchCall* begin(const float* d_queries, uint32_t nq, int ef, uint32_t out_k, uint32_t flags, uint32_t* d_ids,
                      float* d_dist, uint32_t* d_count, hnsw_amd_query_stats* d_stats, void* stream,
                      bool no_alloc = false, const CopyBack* copy_back = nullptr);
    uint32_t finish(SearchCall* call);
    uint32_t search(const float* d_queries, uint32_t nq, int ef, uint32_t out_k, uint32_t flags, uint32_t* d_ids,
                    float* d_dist, uint32_t* d_count, hnsw_amd_query_stats* d_stats, void* stream,
                    bool no_alloc = false, const CopyBack* copy_back = nullptr);
    bool calls_in_flight() const;

private:
    void launch_tier(SearchCall& c);
    void wait_report(SearchCall& c);
    void set_max_dynamic_lds(const void* kernel, size_t lds);
    SearchCall calls_[kMaxCallsInFlight];
    SearchCall noalloc_call_;          // the entry point that allocates nothing: lists in the library's static device array
    void* bufs_[8] = {};
    size_t up_lists_cap_ = 0;          // upper-layer slots the two list tables (bufs_[4], bufs_[5]) have room for
};

// Layer 0 of a GPU-assisted build that is selected and linked on the device (hnsw_build.hip): the edge distances next
// to the index's own device copy of the lists, and the buffers a sub-batch passes through.
struct DeviceBuild {
    float* dist0 = nullptr; size_t dist0_words = 0;          // n x deg0, beside DeviceGraph::adj0
    float* d_q = nullptr; size_t q_words = 0;                // the sub-batch's rows as queries
    uint32_t* d_cids = nullptr; size_t cand_words = 0;       // their candidate lists (ids, distances, lengths)
    float* d_cdist = nullptr; size_t cdist_words = 0;
    uint32_t* d_ccnt = nullptr; size_t ccnt_words = 0;
    uint64_t *d_keys = nullptr, *d_keys2 = nullptr; size_t keys_words = 0, keys2_words = 0;   // (neighbour << 32 | new vector), unsorted / sorted
    float *d_vals = nullptr, *d_vals2 = nullptr; size_t vals_words = 0, vals2_words = 0;      // the distance of each such link
    void* d_tmp = nullptr; size_t tmp_bytes = 0;             // the sort's work space
    // Layer 1 as a graph of its own over the same vectors (lists by node_order), with the layers above it: what the
    // construction searches of vectors that live in layer 1 run on.  Rebuilt from the host's upper layers before every step.
    uint32_t* adj1 = nullptr; size_t adj1_words = 0;         // n x degu, 0xFFFFFFFF where a vector has no layer-1 list
    uint32_t* d_order1 = nullptr; size_t order1_words = 0;   // node_order of every layer-1 slot
    uint32_t* v_up_off = nullptr; size_t v_up_off_words = 0;
    uint32_t* v_up_down = nullptr; size_t v_up_down_words = 0;
    DeviceGraph view{};
    bool view_ok = false;
    static bool supports(const HostIndex& h);                // list lengths and efConstruction the kernels are built for
    void release();
    void from_host(DeviceIndex& dev, const HostIndex& h);    // dist0 <- the host's edge distances
    void upper_view(DeviceIndex& dev, const FlatGraph& upper);   // the view of layer 1, from the upper layers just put on the device
    // vectors first_id .. first_id + cnt - 1 (added to the layers, not linked): search, select, link in layer 0; for those
    // among them that live in layer 1 (upper_ids, n_upper of them) the candidate lists of their layer-1 search
    // (ef = efConstruction entries each at cand_ids / cand_dist, lengths in cand_cnt; 0 = none, search on the host)
    void step(DeviceIndex& dev, const HostIndex& h, uint32_t first_id, uint32_t cnt, uint32_t flags,
              const uint32_t* upper_ids, uint32_t n_upper, uint32_t* cand_ids, float* cand_dist, uint32_t* cand_cnt);
    void to_host(const DeviceIndex& dev, HostIndex& h);      // layer 0's lists, distances and counts into the host's arrays
};

}  // namespace hnsw_amd
