/* hnsw_amd.h — C ABI of the MI355X HNSW query engine (libhnsw_amd.so).
 *
 * The reference (RadimBaca/HNSW) has no FFI: its boundary is the public surface of
 * `class HNSW` (src/hnsw.h:217-603) as used by its driver (src/main.cpp:197-293).
 * Each entry point below is that surface flattened to C — plain pointers and sizes,
 * status codes instead of exceptions — and names the reference member it replaces.
 * include/hnsw_amd.hpp wraps these calls back into a `class HNSW` with the
 * reference's member names.
 *
 * The search entry points run ONLY on the GPU (hand-written HIP kernels for gfx950);
 * there is no CPU search path in this library.  Index construction (insert) runs on
 * the host, as in the reference.
 *
 * Threading: calls on one index must be serialised by the caller (the reference
 * object is not re-entrant either, src/hnsw.cpp:344,380).  Different indexes are
 * independent.  All functions return HNSW_AMD_OK (0) or a negative error code;
 * hnsw_amd_last_error() gives the message of the calling thread's last failure.
 */
#ifndef HNSW_AMD_H
#define HNSW_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hnsw_amd_index hnsw_amd_index;
typedef struct hnsw_amd_search_call hnsw_amd_search_call;   /* a batched search between its begin and its finish */

enum {
    HNSW_AMD_OK = 0,
    HNSW_AMD_E_ARG = -1,      /* bad argument / index state */
    HNSW_AMD_E_FULL = -2,     /* insert beyond max_node_count (reference: std::runtime_error, src/hnsw.cpp:176-179) */
    HNSW_AMD_E_IO = -3,       /* file could not be opened / is malformed */
    HNSW_AMD_E_HIP = -4,      /* a HIP runtime call failed */
    HNSW_AMD_E_SCRATCH = -5,  /* a query outgrew the largest on-chip scratch configuration */
    HNSW_AMD_E_GRAPH = -6     /* imported graph failed validation */
};

/* flags of the search entry points */
enum {
    HNSW_AMD_DIST_SSE = 0,        /* distance of the shipped build: first 16*floor(d/16) dims, 4 lane sums, fused
                                     multiply-add (src/hnsw.h:334-383) */
    HNSW_AMD_DIST_SCALAR = 1,     /* distance of a -DNO_MANUAL_VECTORIZATION build: all d dims, one running sum
                                     (src/hnsw.h:387-400) */
    HNSW_AMD_KEYS_INT32 = 0,      /* candidate heap keyed by int32-truncated distances, as the reference
                                     (src/hnsw.cpp:727,767) */
    HNSW_AMD_KEYS_FLOAT = 2,      /* candidate heap keyed by the float distance (textbook HNSW; not the reference) */
    HNSW_AMD_ROWS_U8 = 8,         /* read the base vectors from a one-byte-per-dimension copy (4x fewer bytes gathered).
                                     Only for an index whose stored values are all integers in 0..255 (SIFT, image
                                     pixels): float(byte) is then the stored float exactly and every result is
                                     unchanged.  HNSW_AMD_E_ARG otherwise, and with HNSW_AMD_DIST_SCALAR.  The
                                     reference sketched byte vectors under COMPUTE_APPROXIMATE_VECTOR
                                     (src/hnsw.cpp:451-467, aprDistance src/hnsw.h:455-467), lossy and disabled. */
    HNSW_AMD_ROWS_Q8 = 16,        /* search a quantised one-byte-per-dimension copy of ANY base (one grid of 256 levels
                                     between the 0.1 % and 99.9 % quantiles of the stored values), the query mapped
                                     to the same grid; the ef results are then re-scored with the exact float
                                     distance and sorted by it.  4x fewer bytes gathered; an approximation of the
                                     reference's search, not a reproduction (the idea the reference left disabled,
                                     see HNSW_AMD_ROWS_U8).  With HNSW_AMD_DIST_SSE only. */
    HNSW_AMD_METRIC_IP = 32,      /* No counterpart in the reference, which has squared L2 only (src/hnsw.h:334-400).  Distance of
                                     the search = max(0, 1 - sum q_i x_i) over the first 16*floor(d/16) dims (same four fused
                                     multiply-add chains and final addition order as HNSW_AMD_DIST_SSE): the inner-product /
                                     angular metric for unit-length vectors, where it is half the squared L2 distance and ranks
                                     the same.  The index is built with squared L2 as always.  With float rows and
                                     HNSW_AMD_DIST_SSE only.  Under HNSW_AMD_KEYS_INT32 every key is 0 or 1: use
                                     HNSW_AMD_KEYS_FLOAT.  Checked against the CPU restatement; parity with the reference is
                                     unpinned because the reference cannot compute it. */
    HNSW_AMD_W_STD_HEAP = 4       /* keep the result list W_ as the libstdc++ binary heap the reference keeps
                                     (src/hnsw.cpp:770-775), so that when several entries tie for the worst
                                     distance the one dropped is the reference's choice as well.  Default (flag
                                     clear): W_ is a sorted list held in registers, faster; same distances, same ids
                                     except possibly among entries exactly tied with the worst kept distance. */
};

/* Per-query counters, identical between this library and the CPU restatement used to
 * test it.  They define the algorithmic HBM traffic of a query:
 *   bytes = n_dist*4*d_used + 4*(sum_deg0 + sum_degu) + 4*d + 8*k                      */
typedef struct {
    uint32_t n_dist;        /* distance computations (one base-vector row fetched each) */
    uint32_t n_expand;      /* layer-0 candidates expanded (one adjacency row each) */
    uint32_t n_sweeps;      /* upper-layer neighbour-list sweeps */
    uint32_t sum_deg0;      /* sum of degrees of the expanded layer-0 nodes */
    uint32_t sum_degu;      /* sum of degrees of the swept upper-layer nodes */
    uint32_t max_c;         /* largest size the candidate heap reached */
    uint32_t n_visited0;    /* visited-set size after the layer-0 search */
    uint32_t n_visited_up;  /* visited-set size after the upper-layer descent */
} hnsw_amd_query_stats;

/* Flat (pointer-free) graph: the reference's Layer/Node/Neighbors objects
 * (src/layer.h:9-38, src/node.h:17-135) as arrays.  Neighbour order is preserved.
 * Layer 0: slot == node_order.  Upper layers 1..nlayers-1 are concatenated, layer l
 * owning slots [up_off[l], up_off[l+1]).  Padding value 0xFFFFFFFF. */
typedef struct {
    uint32_t n, d, nlayers, deg0, degu;
    const uint32_t* adj0;         /* n x deg0 node_orders */
    const uint32_t* cnt0;         /* n */
    const uint32_t* up_off;       /* nlayers+1, up_off[0]=up_off[1]=0 */
    const uint32_t* up_order;     /* node_order of each upper slot */
    const uint32_t* up_down;      /* slot in the layer below (layer 1: node_order) */
    const uint32_t* up_cnt;
    const uint32_t* up_adj_slot;  /* U x degu: neighbour's slot inside the same layer */
    const uint32_t* up_adj_order; /* U x degu: neighbour's node_order */
    const uint32_t* ep_slot;      /* nlayers: entry point of each layer */
    const uint32_t* ep_order;     /* nlayers */
} hnsw_amd_flat_graph;

const char* hnsw_amd_last_error(void);

/* HNSW::HNSW(int M, int Mmax, int efConstruction) — src/hnsw.h:269-280 */
int hnsw_amd_create(int M, int Mmax, int efConstruction, hnsw_amd_index** out);
/* HNSW::~HNSW / HNSW::clean — src/hnsw.h:282-288, src/hnsw.cpp:22-32 */
void hnsw_amd_destroy(hnsw_amd_index* h);
/* HNSW::init(uint32_t vector_size, uint32_t max_node_count) — src/hnsw.cpp:3-20 */
int hnsw_amd_init(hnsw_amd_index* h, uint32_t vector_size, uint32_t max_node_count);
/* HNSW::insert(float* q) — src/hnsw.cpp:163-334.  Sequential, reproduces the reference's graph. */
int hnsw_amd_insert(hnsw_amd_index* h, const float* q);
/* The insert loop of the driver (src/main.cpp:207-210) for n rows.  threads <= 1: the same
 * sequential inserts; threads > 1: concurrent inserts (a valid HNSW graph of the same
 * construction rule, but not the reference's exact adjacency). */
int hnsw_amd_insert_batch(hnsw_amd_index* h, const float* rows, uint32_t n, int threads);
/* The same loop with the construction searches of layer-0-only vectors (19 in 20) run on the GPU: HNSW::insert's
 * searchLayer(q, efConstruction) (src/hnsw.cpp:246-262) is the query path, so the batched search kernel supplies the
 * candidate lists of a sub-batch at a time on the graph as it stands (a sub-batch is a sixteenth of the nodes linked
 * so far, 16384 rows at most — beyond a million linked nodes a sixty-fourth, 65536 at most; while the graph is small, a quarter of it up to 1024 rows), and the host threads run selectNeighbors and the linking (src/hnsw.cpp:263-323,
 * 1078-1126) as in the concurrent build.  Vectors of one sub-batch do not see each other: a valid HNSW graph of the
 * same rule, not the reference's exact adjacency.  Needs threads > 1 and n >= 2048 (else: hnsw_amd_insert_batch).
 * `device`: the GPU that searches (and holds the index afterwards).  If the build fails half way (a HIP error in a
 * candidate search) the index is left unusable: every later insert or search returns HNSW_AMD_E_ARG until
 * hnsw_amd_init / hnsw_amd_load_graph / hnsw_amd_import_flat starts it over. */
int hnsw_amd_insert_batch_assisted(hnsw_amd_index* h, const float* rows, uint32_t n, int threads, int device);
/* The same loop with all of layer 0 on the GPU: after the search of a sub-batch (as above), selectNeighbors for the new
 * vectors (src/hnsw.cpp:263, 1078-1126), their own lists, the links back from every selected neighbour and the
 * re-selection of a neighbour's list that is full (src/hnsw.cpp:266-312) run as kernels on the device's copy of layer 0;
 * a sub-batch here is a sixty-fourth of the nodes linked so far (65536 at most);
 * the links of a sub-batch are sorted by neighbour and every neighbour's list is updated by one wavefront, in the order
 * of the new vectors' ids, so the graph does not depend on timing.  The host threads only insert the one vector in
 * sixteen that lives in upper layers into those layers (its layer-1 candidate list comes from a device search on that layer too); layer 0 is copied back into the host's arrays when the call
 * ends (and around a vector that opens a new top layer, which is inserted on the host).  Distances between two stored
 * vectors are summed in another order than insert() does, so comparisons that are near-ties may fall differently: a
 * valid HNSW graph of the same rule, neither the reference's exact adjacency nor the assisted build's.  For M <= Mmax0
 * <= 32 and efConstruction <= 4096; anything else is built by hnsw_amd_insert_batch_assisted.  Same conditions, same
 * behaviour after a failure. */
int hnsw_amd_insert_batch_linked(hnsw_amd_index* h, const float* rows, uint32_t n, int threads, int device);
/* Construction option with no counterpart in the reference: key the candidate heap of the searches that insert()
 * runs by the float distance instead of the reference's int32 truncation (src/hnsw.cpp:727,767).  Off by default
 * (the sequential build then reproduces the reference's graph).  On unit-norm data the truncated keys are all 0..3,
 * the construction search never stops early, and turning this on makes the build several times faster; the result
 * is a regular HNSW graph, not the reference's adjacency. */
int hnsw_amd_set_build_float_keys(hnsw_amd_index* h, int on);
/* Which of the reference's two builds insert() reproduces: HNSW_AMD_DIST_SSE (default) = the shipped one, whose distance
 * (src/hnsw.h:334-383) uses the first 16*floor(d/16) dims; HNSW_AMD_DIST_SCALAR = a -DNO_MANUAL_VECTORIZATION build
 * (src/hnsw.h:2, 387-400), whose plain loop uses all d dims — the reference build to compare with when d is not a
 * multiple of 16 (GloVe-100).  insert() and the searches of one reference binary use the same distance, so an index
 * built this way is searched with the same flag.  Sequential inserts reproduce that build's adjacency. */
int hnsw_amd_set_build_distance(hnsw_amd_index* h, uint32_t dist_flag);

/* HNSW::aproximateKnn(float* q, int k, int ef) — src/hnsw.cpp:373-449.  The reference leaves
 * its result in the public member W_ (up to ef entries, ascending squared-L2 distance; k is
 * ignored, src/hnsw.cpp:373): here W_ is returned through out_ids/out_dist (room for ef
 * entries each) and *out_count.  Runs on the GPU. */
int hnsw_amd_aproximate_knn(hnsw_amd_index* h, const float* q, int k, int ef, uint32_t flags,
                            uint32_t* out_ids, float* out_dist, uint32_t* out_count);
/* The query loop of the driver (src/main.cpp:237-240, 291-294) as one batched call: nq
 * queries (row-major nq x d, host memory), the first out_k (<= ef) entries of each query's
 * W_ into out_ids/out_dist (nq x out_k; short rows padded with 0xFFFFFFFF / +inf),
 * |W_| into out_count (may be NULL), counters into stats (nq entries, may be NULL). */
int hnsw_amd_search_batch(hnsw_amd_index* h, const float* queries, uint32_t nq, int ef, uint32_t out_k,
                          uint32_t flags, uint32_t* out_ids, float* out_dist, uint32_t* out_count,
                          hnsw_amd_query_stats* stats);
/* Same with every buffer already in device memory of the index's GPU.  The kernel runs on
 * `stream` (a hipStream_t, NULL = default stream); the call then waits for that stream to
 * learn whether any query outgrew its on-chip scratch and re-runs those on larger tables,
 * so every row of the outputs is final when it returns.  *retried (host, may be NULL)
 * receives the number of queries that were re-run. */
int hnsw_amd_search_batch_device(hnsw_amd_index* h, const float* d_queries, uint32_t nq, int ef, uint32_t out_k,
                                 uint32_t flags, uint32_t* d_out_ids, float* d_out_dist, uint32_t* d_out_count,
                                 hnsw_amd_query_stats* d_stats, void* stream, uint32_t* retried);

/* No counterpart in the reference (whose driver runs one blocking query at a time).  hnsw_amd_search_batch_device in
 * two halves, so that a caller can keep several batches in flight on one index and use the time a batch takes on the
 * GPU: begin sizes the batch's on-chip scratch from what earlier batches needed, launches it on `stream` and returns
 * at once with a handle in *call; finish waits for that launch, re-runs the queries that outgrew their scratch on
 * larger tables (on the same stream, waiting for each such launch) and only then are all output rows final;
 * *retried (host, may be NULL) as above.  Every successful begin needs exactly one finish, which also releases the
 * handle, whatever it returns.  At most 4 calls may be in flight per index (HNSW_AMD_E_ARG beyond); they may use
 * different streams and be finished in any order; each has its own scratch lists and spill tables, so calls that
 * overlap on the GPU do not interfere.  Between a begin and its finish the index must not be modified, re-uploaded or
 * destroyed, and the output buffers of calls in flight must be distinct.  begin and finish themselves are ordinary
 * calls on the index: one thread at a time. */
int hnsw_amd_search_batch_device_begin(hnsw_amd_index* h, const float* d_queries, uint32_t nq, int ef, uint32_t out_k,
                                       uint32_t flags, uint32_t* d_out_ids, float* d_out_dist, uint32_t* d_out_count,
                                       hnsw_amd_query_stats* d_stats, void* stream, hnsw_amd_search_call** call);
int hnsw_amd_search_batch_device_finish(hnsw_amd_index* h, hnsw_amd_search_call* call, uint32_t* retried);

/* No counterpart in the reference.  hnsw_amd_search_batch_device for callers that cannot allocate device memory
 * (tight serving loops, harnesses that forbid hipMalloc): same arguments, same results, but the call itself
 * allocates nothing, neither on the device nor pinned host memory.  The lists of the retry protocol (queries that
 * outgrew their on-chip scratch) live in a static device array of the library, 512 KB per GPU, held until the
 * process ends.  Limits: nq <= 65535 per call (HNSW_AMD_E_ARG beyond); the index must already be on its device
 * (hnsw_amd_sync_device after the last insert/load; HNSW_AMD_E_ARG otherwise), and with HNSW_AMD_ROWS_Q8 the
 * quantised table must exist already.  All calls a process makes on one GPU share that array (whatever the index):
 * calls that overlap in time on the same GPU race and must be serialised by the caller.  Like the entry point
 * above it waits for `stream` before it returns. */
int hnsw_amd_search_batch_device_noalloc(hnsw_amd_index* h, const float* d_queries, uint32_t nq, int ef, uint32_t out_k,
                                         uint32_t flags, uint32_t* d_out_ids, float* d_out_dist, uint32_t* d_out_count,
                                         hnsw_amd_query_stats* d_stats, void* stream, uint32_t* retried);

/* HNSW::saveGraph / HNSW::loadGraph — src/hnsw.cpp:1178-1240, 1244-1311; same file format. */
int hnsw_amd_save_graph(hnsw_amd_index* h, const char* filename);
int hnsw_amd_load_graph(hnsw_amd_index* h, const char* filename);

/* "Identical adjacency exported from the reference build": adopt a flat graph and its
 * vectors (n x d row-major) instead of building; and the reverse.  The exported pointers
 * stay owned by the index and valid until it is modified or destroyed. */
int hnsw_amd_import_flat(hnsw_amd_index* h, const hnsw_amd_flat_graph* g, const float* vectors);
int hnsw_amd_export_flat(hnsw_amd_index* h, hnsw_amd_flat_graph* g, const float** vectors);

/* Copy (or refresh) the flat index into the HBM of `device`.  Search calls do this
 * themselves when the index changed; calling it first keeps the upload out of timings. */
int hnsw_amd_sync_device(hnsw_amd_index* h, int device);

/* HDF5 input path.  hdfReader::getDimensions — src/hdfReader.cpp:9-23 (dims of a 2-D
 * dataset; a 1-D dataset reports dims[1] = 1); hdfReader::readData(float*) — :26-42 and
 * readData(int*) — :45-61 (the whole dataset, converted to float / int, into a buffer the
 * caller sized from the dimensions).  libhdf5 is loaded at run time; without it these
 * return HNSW_AMD_E_IO and everything else still works. */
int hnsw_amd_hdf_get_dimensions(const char* filename, const char* dataset, uint64_t dims[2]);
int hnsw_amd_hdf_read_float(const char* filename, const char* dataset, float* data);
int hnsw_amd_hdf_read_int(const char* filename, const char* dataset, int* data);

/* Exact k nearest base rows of every query by squared L2 (brute force), all buffers in
 * device memory of the current GPU: ground truth for recall figures.  The reference has no
 * counterpart (it reads ground truth from the dataset's "neighbors" array,
 * src/main.cpp:190-195).  Output rows ascending by (distance, id), ids offset by id_offset,
 * k <= 128.  The q.x products run on the matrix cores (f32-input MFMA); *mfma_ms (host, may be
 * NULL) receives the device time of that kernel: 2*nq*n*d flop / mfma_ms is its rate.
 * Synchronises the stream. */
int hnsw_amd_bruteforce_knn_device(const float* d_base, uint32_t n, const float* d_queries, uint32_t nq, uint32_t d, uint32_t k,
                                   uint32_t id_offset, uint32_t* d_out_ids, float* d_out_dist, void* stream, float* mfma_ms);

/* Sharded multi-GPU search, combine step (no counterpart in the reference, which is one
 * in-memory index): d_dist / d_ids hold n_shards per-shard result blocks of nq x k each
 * (ids local to the shard, 0xFFFFFFFF = padding), as an all-gather leaves them;
 * shard_offset[s] (host array) is the global id of shard s's first vector; shard s's block
 * starts shard_stride elements after shard s-1's (0 = nq*k, i.e. blocks back to back; a
 * larger stride lets distances and ids travel in one gathered buffer).  Writes the
 * out_k nearest of each query's n_shards*k candidates (global ids, ascending distance,
 * ties by id) on the current device, asynchronously on `stream`. */
int hnsw_amd_merge_topk_device(const float* d_dist, const uint32_t* d_ids, const uint32_t* shard_offset, uint32_t n_shards,
                               uint32_t nq, uint32_t k, uint32_t out_k, uint64_t shard_stride, uint32_t* d_out_ids, float* d_out_dist,
                               void* stream);

/* Development aid, no counterpart in the reference: the library's diagnostic switches (environment variables
 * HNSW_AMD_VERBOSE, HNSW_AMD_VISITED32, ...; listed where they are read, hnsw_kernels.hip) are read when an index is
 * copied to its GPU, not in every search call; this reads them again for `h`. */
int hnsw_amd_reload_env(hnsw_amd_index* h);

/* Public data members of the reference object that callers read (src/hnsw.h:221-236). */
int hnsw_amd_get_info(hnsw_amd_index* h, int32_t* M, int32_t* Mmax, int32_t* Mmax0, int32_t* efConstruction,
                      uint32_t* vector_size, int32_t* max_node_count, int32_t* actual_node_count, uint32_t* nlayers);

/* Measurement aid, no counterpart in the reference: the rate at which this GPU delivers the search kernel's row
 * reads when nothing else is in the way.  n_waves wavefronts each compute distances to rows_per_wave (a multiple of
 * 32) uniformly random rows of the index's own vector table with the search kernel's loads and arithmetic, but
 * without heaps, visited set, neighbour lists or any dependence between batches; lds_bytes_per_wave sets how many
 * wavefronts a CU holds, as the search's scratch does; dist_flag (HNSW_AMD_DIST_SSE / HNSW_AMD_DIST_SCALAR) picks the
 * arithmetic.  *ms = time of one launch, *bytes = what the search's counters would charge for those rows
 * (4 * 16*floor(d/16) each, or 4 * d in scalar order); bytes / ms is the ceiling a search on this index can be
 * compared with.  Waits for the stream. */
int hnsw_amd_gather_ceiling_device(hnsw_amd_index* h, uint32_t n_waves, uint32_t rows_per_wave, uint32_t lds_bytes_per_wave, uint32_t dist_flag,
                                   void* stream, float* ms, uint64_t* bytes);

/* Measurement aid, no counterpart in the reference: how the last search call's (first) launch was configured — on-chip
 * scratch (LDS) bytes per query, how many queries a CU holds at once with that (what bounds a latency-bound
 * search), and the form of the visited set (0: 32-bit slots, 1: 16-bit slots probed inside large regions,
 * 2: 16-bit slots in small regions read whole, with overflow and spill tables).  Any pointer may be NULL. */
int hnsw_amd_last_launch_info(hnsw_amd_index* h, uint32_t* lds_bytes_per_query, uint32_t* resident_queries_per_cu, uint32_t* visited_form);

/* HNSW::vector_data_ — src/hnsw.h:231 (getNodeVector, :317): the index's host copy of the vectors,
 * max_node_count x vector_size floats, row = node_order; NULL before init.  Owned by the index, read-only,
 * valid until the next init / load_graph / import_flat / destroy. */
int hnsw_amd_vector_data(hnsw_amd_index* h, const float** out);

#ifdef __cplusplus
}
#endif
#endif /* HNSW_AMD_H */
