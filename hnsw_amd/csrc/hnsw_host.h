// Host-side HNSW index: construction, (de)serialisation and the flat export the GPU reads.
//
// The reference keeps one heap-allocated Node per (vector, layer) with a
// std::vector<Neighbors> inside and raw pointers between nodes (src/node.h:69-135,
// src/layer.h:9-38).  Here every layer is a structure of fixed-degree arrays indexed by
// slot; a neighbour is a (slot, node_order, distance) triple in three parallel arrays.
// The construction rule is the reference's (HNSW::insert, src/hnsw.cpp:163-334;
// selectNeighbors, :1078-1126) and, run sequentially, yields the same adjacency in the
// same order.
#pragma once

#include <atomic>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace hnsw_amd {

constexpr uint32_t kPad = 0xFFFFFFFFu;

// Element of the result list W_ (reference: struct Neighbors, src/node.h:17-40), with the
// node pointer replaced by the node's slot in the layer being searched.
struct Neighbor {
    float distance;
    uint32_t node_order;
    uint32_t slot;
};

struct LayerStore {
    uint32_t deg = 0;       // capacity of a neighbour list: 2*Mmax on layer 0, Mmax above
    uint32_t count = 0;     // nodes in this layer
    std::vector<uint32_t> order;   // node_order of slot s (layer 0: s itself)
    std::vector<uint32_t> down;    // slot of the same vector in the layer below
    std::vector<uint32_t> cnt;     // neighbour count
    std::vector<uint8_t> sorted;   // reference: Node::neighbors_sorted
    std::vector<uint32_t> nb_slot, nb_order;  // [slot*deg + j]
    std::vector<float> nb_dist;               // distance stored on the edge (feeds re-selection)
    uint32_t ep_slot = 0, ep_order = 0;       // entry point: the node that created the layer

    void reserve_nodes(size_t n);
    uint32_t add_node(uint32_t node_order, uint32_t down_slot);
};

// Flat arrays in the layout of hnsw_amd_flat_graph (include/hnsw_amd.h).
struct FlatGraph {
    uint32_t n = 0, d = 0, nlayers = 0, deg0 = 0, degu = 0;
    std::vector<uint32_t> adj0, cnt0, up_off, up_order, up_down, up_cnt, up_adj_slot, up_adj_order, ep_slot, ep_order;
};

class GlibcRand;  // level generator

class HostIndex {
public:
    int M_, Mmax_, Mmax0_, efConstruction_;
    float ml_;
    int max_node_count_ = 0;
    int actual_node_count_ = -1;   // id of the last inserted vector (reference member of the same name)
    uint32_t vector_size_ = 0;
    std::vector<LayerStore> layers_;
    std::vector<float> vector_data_;   // max_node_count_ x vector_size_, row = node_order
    bool build_float_keys_ = false;    // construction searches keyed by the float distance instead of its int32 truncation
    bool dist_scalar_ = false;         // insert() computes distances as the reference's -DNO_MANUAL_VECTORIZATION build does
                                       // (src/hnsw.h:387-400: all d dims, one running sum) instead of the shipped SSE loop
    float dist(const float* a, const float* b) const;   // the distance insert() uses
    uint64_t version_ = 0;             // bumped by every mutation (device copy refresh)
    uint64_t rows_epoch_ = 0;          // identifies the contents of vector_data_'s table: a new value (unique in the process) whenever
                                       // init / load_graph / import_flat replaces it, so a device copy of the rows knows whose rows it holds
    bool build_failed_ = false;        // a concurrent build stopped half way (nodes added, not all linked): searches and inserts refuse until
                                       // init / load_graph / import_flat starts over
    double assisted_search_s = 0.0, assisted_link_s = 0.0;   // where an assisted build's time went (HNSW_AMD_VERBOSE prints them):
                                                             // candidate searches (device-linked build: the whole device step), host selecting and linking
    uint32_t assisted_batches = 0;

    HostIndex(int M, int Mmax, int efConstruction);
    ~HostIndex();

    void init(uint32_t vector_size, uint32_t max_node_count);
    void insert(const float* q);                                   // sequential, reference-exact
    // Results of layer-0 searches with ef = efConstruction run by someone else (the GPU) on the graph as it stands:
    // for query j, cnt[j] entries (node_order ascending by distance, and the distances) at ids/dist + j * ef.
    typedef void (*CandidateSearchFn)(void* ctx, const float* queries, uint32_t nq, int ef, uint32_t* ids, float* dist, uint32_t* cnt);
    // threads > 1: concurrent.  With `candidates`, vectors that live in layer 0 only take their candidate lists from it
    // (in sub-batches, between which it sees the graph again) instead of searching on the host.
    // With `layer0`, layer 0 of the concurrent build lives somewhere else (the GPU) between from_host and to_host: each
    // sub-batch is searched, selected and linked there by step(), and the host threads only insert the sub-batch's
    // vectors into the layers above.  from_host: the graph as the host has it now is the starting point; upper_layers:
    // the layers above as the host has them now are what the next step descends through (called while no host thread
    // changes them); step: called from a thread of its own while the host threads link the previous sub-batch into the
    // upper layers — it may read vector_data_ and nothing else of the index; to_host: layer 0's lists, edge distances
    // and counts are written back into layers_[0].
    struct Layer0Linker {
        void* ctx;
        void (*from_host)(void* ctx);
        void (*upper_layers)(void* ctx);
        // upper_ids: those of the step's vectors that also live in layer 1; their layer-1 candidate lists come back in
        // cand_ids / cand_dist (efConstruction entries each) and cand_cnt (0: none, the host searches itself)
        void (*step)(void* ctx, uint32_t first_id, uint32_t cnt, const uint32_t* upper_ids, uint32_t n_upper,
                     uint32_t* cand_ids, float* cand_dist, uint32_t* cand_cnt);
        void (*to_host)(void* ctx);
    };
    void insert_batch(const float* rows, uint32_t n, int threads, CandidateSearchFn candidates = nullptr, void* ctx = nullptr,
                      const Layer0Linker* layer0 = nullptr);
    void save_graph(const char* filename) const;                   // reference file format
    void load_graph(const char* filename);
    void import_flat(const FlatGraph& g, const float* vectors);
    // (without layer 0: upper layers and entry points only; without the upper lists: not even up_adj_* / up_cnt — a caller
    // that reads the lists from layers_[l] itself)
    void export_flat(FlatGraph& g, bool with_layer0 = true, bool with_upper_lists = true) const;
    uint32_t size() const { return (uint32_t)(actual_node_count_ + 1); }
    const float* row(uint32_t node_order) const { return vector_data_.data() + (size_t)node_order * vector_size_; }

private:
    struct Scratch;
    std::unique_ptr<GlibcRand> rng_;
    std::unique_ptr<Scratch> scratch_;                 // for the sequential path
    std::vector<std::unique_ptr<std::atomic_flag[]>> locks_;  // per layer, per slot (concurrent path)
    std::vector<size_t> lock_cap_;

    void parse_graph_file(const char* filename);
    void validate_loaded(const char* filename) const;
    int draw_level();
    template <bool MT> void insert_one(uint32_t id, const float* q, int level, const uint32_t* new_slots, Scratch& s,
                                       const uint32_t* cand_ids = nullptr, const float* cand_dist = nullptr, uint32_t cand_cnt = 0,
                                       bool upper_layers_only = false, const uint32_t* slot1_of = nullptr);
    template <bool MT> void search_layer_one(int layer, const float* q, Scratch& s);
    template <bool MT> void search_layer(int layer, const float* q, int ef, Scratch& s);
    void select_neighbors(std::vector<Neighbor>& W, std::vector<Neighbor>& R, int M, int layer) const;
    template <bool MT> void link(int layer, uint32_t new_slot, uint32_t id, const std::vector<Neighbor>& R, Scratch& s);
    void change_layer(int layer, std::vector<Neighbor>& W) const;
    void ensure_locks();
};

// Squared L2 exactly as the reference's shipped build computes it (src/hnsw.h:334-383).
float distance_sse_order(const float* a, const float* b, uint32_t d);
// ... and as its -DNO_MANUAL_VECTORIZATION build computes it (src/hnsw.h:387-400 compiled by g++ 11 -O3 -march=native).
float distance_scalar_order(const float* a, const float* b, uint32_t d);

}  // namespace hnsw_amd
