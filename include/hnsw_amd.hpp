// hnsw_amd.hpp — the reference's `class HNSW` (RadimBaca/HNSW src/hnsw.h:217-603) as a thin
// header-only wrapper over the C ABI of libhnsw_amd.so.  Same constructor, member functions
// and public data member names, so the reference's driver (src/main.cpp, default mode:
// sift_test(), :167-319) compiles against it unmodified when a file named hnsw.h that includes
// this header and the reference's settings.h stands in for the reference's hnsw.h
// (tests/test_reference_driver_compiles.py does exactly that):
//
//     HNSW hnsw(16, 16, 200);  hnsw.init(d, n);  hnsw.insert(v);  hnsw.loadGraph(f);
//     hnsw.stat_.clear();  hnsw.aproximateKnn(q, k, ef);  for (auto item : hnsw.W_) use(item.node_order);
//
// Searches run on the GPU (MI355X); there is no CPU search behind this class.  Errors that the
// reference reports by exception (insert beyond capacity: std::runtime_error,
// src/hnsw.cpp:176-179) are exceptions here too; saveGraph/loadGraph print a message and
// return on an unopenable file, and print and exit(0) on any other failure, like the reference
// (src/hnsw.cpp:1183-1187,1236-1239,1248-1251,1305-1308).
//
// Public data members of the reference object and what they hold here:
//   M_ Mmax_ Mmax0_ efConstruction_ ml_ max_node_count_ min_M_ actual_node_count_ vector_size_ vector_parts
//                    same values (actual_node_count_ is not bumped by queries: upstream's bump, src/hnsw.cpp:380,
//                    corrupts later inserts and saves)
//   W_               result of the last aproximateKnn, as upstream; the Node* of an entry is null
//   vector_data_     the index's host copy of the vectors (read-only here), valid after init / loadGraph
//   stat_            precise_distance_computations grows by the query's distance computations when the scalar
//                    distance is selected (upstream counts in the scalar distance only, src/hnsw.h:390; its SSE
//                    build prints zeros and so does this); the other counters are touched upstream only by the
//                    disabled uint8 paths and stay 0.  last_query_stats has this library's own counters.
//   visited_         item_count = size of the visited set after the last query's layer-0 search, as upstream;
//                    the set itself lives on the GPU and is not mirrored (get() answers for what insert() put in)
//   layers_          a copy of the graph in the reference's Layer / Node / Neighbors objects, rebuilt by
//                    syncLayers() (and after loadGraph / insertBatch / create when mirror_layers is set);
//                    edge distances are not kept by the flat graph and read 0, as after upstream's loadGraph
//   index_memory_ distances_ data_cleaned_   present; only upstream's disabled uint8 paths ever fill the first two
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "hnsw_amd.h"

using pointer_t = uint32_t;   // src/node.h:13
class Node;

#ifndef H5_VERS_MAJOR
typedef unsigned long long hsize_t;    // as in hdf5.h, for callers that do not include it
#endif

// The reference's hdfReader (src/hdfReader.h:9-17) with the same three static functions.
// libhdf5 is loaded by libhnsw_amd.so at run time.  On failure a message is printed and
// the output is left untouched, as the reference does (src/hdfReader.cpp:19-22).
class hdfReader {
public:
    static void getDimensions(const char* filename, const char* datasetname, hsize_t (*dimensions)[2])
    {
        uint64_t d[2];
        if (hnsw_amd_hdf_get_dimensions(filename, datasetname, d) != HNSW_AMD_OK) { std::cerr << hnsw_amd_last_error() << "\n"; return; }
        (*dimensions)[0] = d[0]; (*dimensions)[1] = d[1];
    }
    static void readData(const char* filename, const char* datasetname, float* data)
    {
        if (hnsw_amd_hdf_read_float(filename, datasetname, data) != HNSW_AMD_OK) std::cerr << hnsw_amd_last_error() << "\n";
    }
    static void readData(const char* filename, const char* datasetname, int* data)
    {
        if (hnsw_amd_hdf_read_int(filename, datasetname, data) != HNSW_AMD_OK) std::cerr << hnsw_amd_last_error() << "\n";
    }
};

// src/node.h:17-40
struct Neighbors {
    float distance;
    pointer_t node_order;
    Node* node;
    Neighbors(const Neighbors&) = default;
    Neighbors& operator=(const Neighbors&) = default;
    Neighbors(Node* n, float dist, pointer_t unique_id) : distance(dist), node_order(unique_id), node(n) {}
};

// src/node.h:69-135 (the members a reader of the graph uses)
class Node {
public:
    std::vector<Neighbors> neighbors;
    Node* lower_layer;
    bool neighbors_sorted = false;
    Node(const int neighbor_size, Node* lower) : lower_layer(lower) { neighbors.reserve(neighbor_size); }
    Node(const Node&) = delete;
};

// src/layer.h:9-38
class Layer {
public:
    int layer_id = 0;
    Node* enter_point = nullptr;
    pointer_t ep_node_order = 0;
    uint32_t node_count = 0;
    std::vector<Node*> nodes;
    Layer() {}
    ~Layer() { for (auto& n : nodes) delete n; }
    Layer(const Layer&) = delete;
};

// src/hnsw.h:55-125: fixed-size open-addressing set of node_orders, slot = index & mask, no overflow check.
class linearHash {
    std::vector<uint32_t> slots_;
    uint32_t mask_;
public:
    uint32_t item_count = 0;
    explicit linearHash(uint32_t asize = 16384) : slots_(asize, 0xFFFFFFFFu), mask_(asize - 1) {}
    void clear() { item_count = 0; std::fill(slots_.begin(), slots_.end(), 0xFFFFFFFFu); }
    void reduce(int shift) { slots_.resize(slots_.size() >> shift); mask_ = (uint32_t)slots_.size() - 1; }
    void insert(uint32_t index)
    {
        uint32_t at = index & mask_;
        while (slots_[at] != 0xFFFFFFFFu) at = at + 1 >= slots_.size() ? 0 : at + 1;
        slots_[at] = index; item_count++;
    }
    bool get(uint32_t index) const
    {
        for (uint32_t at = index & mask_; slots_[at] != 0xFFFFFFFFu; at = at + 1 >= slots_.size() ? 0 : at + 1)
            if (slots_[at] == index) return true;
        return false;
    }
};

// src/hnsw.h:128-184
class HNSW_Stat {
public:
    int precise_distance_computations, distance_computations, distance_computations_false, explored_nodes, overflows,
        neighbor_count, node_count;
    int histogram[200];
    HNSW_Stat() { clear(); }
    void clear()
    {
        precise_distance_computations = distance_computations = distance_computations_false = explored_nodes = 0;
        overflows = neighbor_count = node_count = 0;
        std::fill(histogram, histogram + 200, 0);
    }
    void print()
    {
        std::cout << "No. precise distance computations: " << precise_distance_computations << "\n";
        std::cout << "No. distance computations: " << distance_computations << "\n";
        std::cout << "No. explored nodes: " << explored_nodes << "\n";
    }
    void print_tree_info()
    {
        if (overflows > 0) {
            std::cout << "No. overflows: " << overflows << "\n";
            std::cout << "No. neighbors (avg. neighbor): " << neighbor_count << " (" << (float)neighbor_count / node_count << ")\n";
        }
    }
};

class HNSW {
public:
    // the reference's public data members (src/hnsw.h:221-245, 267); see the table at the top of this file
    int M_, Mmax_, Mmax0_, efConstruction_;
    float ml_;
    int max_node_count_ = 0;
    int min_M_;
    long long index_memory_ = 0;
    std::vector<std::unique_ptr<Layer>> layers_;
    const float* vector_data_ = nullptr;
    bool data_cleaned_ = true;
    int actual_node_count_ = -1;
    uint32_t vector_size_ = 0;
    std::vector<Neighbors> W_;
    std::unordered_map<pointer_t, uint32_t> distances_;
    linearHash visited_;
    HNSW_Stat stat_;
    uint32_t vector_parts = 0;
    // not in the reference
    uint32_t search_flags = HNSW_AMD_DIST_SSE | HNSW_AMD_KEYS_INT32;   // see hnsw_amd.h
    hnsw_amd_query_stats last_query_stats = {};                         // counters of the last aproximateKnn
    bool mirror_layers = false;                                         // keep layers_ current after bulk changes

    HNSW(int M, int Mmax, int efConstruction) : M_(M), Mmax_(Mmax), Mmax0_(Mmax * 2), efConstruction_(efConstruction), min_M_(M / 2)
    {
        check(hnsw_amd_create(M, Mmax, efConstruction, &h_));
        refresh();
    }
    ~HNSW() { hnsw_amd_destroy(h_); }
    HNSW(const HNSW&) = delete;
    HNSW& operator=(const HNSW&) = delete;

    void init(uint32_t vector_size, uint32_t max_node_count)
    {
        check(hnsw_amd_init(h_, vector_size, max_node_count));
        data_cleaned_ = false;
        layers_.clear();
        refresh();
    }
    void clean() { data_cleaned_ = true; }
    void insert(float* q) { check(hnsw_amd_insert(h_, q)); actual_node_count_++; }
    // the driver's insert loop (src/main.cpp:207-210) in one call; threads > 1 builds concurrently
    void insertBatch(const float* rows, uint32_t n, int threads)
    {
        check(hnsw_amd_insert_batch(h_, rows, n, threads));
        refresh();
        if (mirror_layers) syncLayers();
    }
    // the same with the construction searches of layer-0-only vectors run on the GPU (hnsw_amd_insert_batch_assisted):
    // several times faster, a graph of the same rule but not the reference's exact adjacency
    void insertBatchGpu(const float* rows, uint32_t n, int threads, int device = 0)
    {
        check(hnsw_amd_insert_batch_assisted(h_, rows, n, threads, device));
        refresh();
        if (mirror_layers) syncLayers();
    }
    // ... and with selection and linking of layer 0 on the GPU as well (hnsw_amd_insert_batch_linked): the host threads only
    // insert into the upper layers
    void insertBatchDevice(const float* rows, uint32_t n, int threads, int device = 0)
    {
        check(hnsw_amd_insert_batch_linked(h_, rows, n, threads, device));
        refresh();
        if (mirror_layers) syncLayers();
    }
    // construction searches keyed by the float distance instead of the int32 truncation (see hnsw_amd.h)
    void setBuildFloatKeys(bool on) { check(hnsw_amd_set_build_float_keys(h_, on ? 1 : 0)); }
    // insert() with the distance of a -DNO_MANUAL_VECTORIZATION build of the reference (src/hnsw.h:387-400) instead of the shipped SSE loop
    void setBuildScalarDistance(bool on) { check(hnsw_amd_set_build_distance(h_, on ? HNSW_AMD_DIST_SCALAR : HNSW_AMD_DIST_SSE)); }

    // src/hnsw.cpp:373-449: result in W_, up to ef entries ascending; k is ignored as upstream
    void aproximateKnn(float* q, int k, int ef)
    {
        (void)k;
        ids_.resize(ef > 0 ? ef : 1); dist_.resize(ids_.size());
        uint32_t cnt = 0;
        check(hnsw_amd_search_batch(h_, q, 1, ef, (uint32_t)ids_.size(), search_flags, ids_.data(), dist_.data(), &cnt, &last_query_stats));
        W_.clear();
        for (uint32_t i = 0; i < cnt; i++) W_.emplace_back(nullptr, dist_[i], ids_[i]);
        visited_.item_count = last_query_stats.n_visited0;
        if (search_flags & HNSW_AMD_DIST_SCALAR) stat_.precise_distance_computations += (int)last_query_stats.n_dist;
    }
    // nq queries at once: first out_k entries of every query's W_ (row-major nq x out_k)
    void searchBatch(const float* queries, uint32_t nq, int ef, uint32_t out_k, uint32_t* out_ids, float* out_dist,
                     uint32_t* out_count = nullptr, hnsw_amd_query_stats* stats = nullptr)
    {
        check(hnsw_amd_search_batch(h_, queries, nq, ef, out_k, search_flags, out_ids, out_dist, out_count, stats));
    }
    // The batched search on buffers that are already in the GPU's memory, as one blocking call or in two halves (up to
    // four batches in flight on one index; no counterpart in the reference, whose driver runs one blocking query at a time).
    uint32_t searchBatchDevice(const float* d_queries, uint32_t nq, int ef, uint32_t out_k, uint32_t* d_ids, float* d_dist,
                               uint32_t* d_count = nullptr, void* stream = nullptr)
    {
        uint32_t retried = 0;
        check(hnsw_amd_search_batch_device(h_, d_queries, nq, ef, out_k, search_flags, d_ids, d_dist, d_count, nullptr, stream, &retried));
        return retried;
    }
    hnsw_amd_search_call* searchBatchDeviceBegin(const float* d_queries, uint32_t nq, int ef, uint32_t out_k, uint32_t* d_ids,
                                                 float* d_dist, uint32_t* d_count = nullptr, void* stream = nullptr)
    {
        hnsw_amd_search_call* call = nullptr;
        check(hnsw_amd_search_batch_device_begin(h_, d_queries, nq, ef, out_k, search_flags, d_ids, d_dist, d_count, nullptr, stream, &call));
        return call;
    }
    uint32_t searchBatchDeviceFinish(hnsw_amd_search_call* call)
    {
        uint32_t retried = 0;
        check(hnsw_amd_search_batch_device_finish(h_, call, &retried));
        return retried;
    }
    void syncDevice(int device = 0) { check(hnsw_amd_sync_device(h_, device)); }

    // src/hnsw.cpp:1178-1240, 1244-1311: an unopenable file is reported and the call returns; any other failure is
    // reported and ends the program with exit(0), as upstream's catch blocks do
    void saveGraph(const char* filename) { fileCall(hnsw_amd_save_graph(h_, filename), filename); }
    void loadGraph(const char* filename)
    {
        if (!fileCall(hnsw_amd_load_graph(h_, filename), filename)) return;
        data_cleaned_ = false;
        refresh();
        if (mirror_layers) syncLayers();
    }
    // src/hnsw.cpp:1319-1339
    void printInfo(bool /*all*/)
    {
        hnsw_amd_flat_graph g;
        check(hnsw_amd_export_flat(h_, &g, nullptr));
        std::cout << "HNSW info\n";
        for (uint32_t l = 0; l < g.nlayers; l++)
            std::cout << "Layer " << l << " " << (l == 0 ? g.n : g.up_off[l + 1] - g.up_off[l]) << "\n";
        std::cout << "Index memory: " << index_memory_ / (1024 * 1024) << " [MB]\n";
        std::cout << "M_: " << M_ << "\n" << "Mmax_: " << Mmax_ << "\n" << "ml_: " << ml_ << "\n";
        std::cout << "efConstruction_: " << efConstruction_ << "\n";
    }
    // src/hnsw.cpp:1342-1360: the sorted neighbour ids of the first max_count layer-0 nodes
    void print(int max_count)
    {
        hnsw_amd_flat_graph g;
        check(hnsw_amd_export_flat(h_, &g, nullptr));
        for (int i = 0; i < max_count && (uint32_t)i < g.n; i++) {
            std::cout << i << ": ";
            std::vector<int> d(g.adj0 + (size_t)i * g.deg0, g.adj0 + (size_t)i * g.deg0 + g.cnt0[i]);
            std::sort(d.begin(), d.end());
            for (auto item : d) std::cout << item << "  ";
            std::cout << "\n";
        }
        std::cout << "\n\n";
    }
    // Rebuilds layers_ (the reference's object graph, src/hnsw.h:230) from the index: one Node per (vector, layer),
    // neighbours in list order with their Node pointers, lower_layer links, entry points.
    void syncLayers()
    {
        hnsw_amd_flat_graph g;
        check(hnsw_amd_export_flat(h_, &g, nullptr));
        layers_.clear();
        for (uint32_t l = 0; l < g.nlayers; l++) {
            auto L = std::make_unique<Layer>();
            const uint32_t cnt = l == 0 ? g.n : g.up_off[l + 1] - g.up_off[l];
            L->layer_id = (int)l; L->node_count = cnt;
            for (uint32_t s = 0; s < cnt; s++) {
                Node* lower = l == 0 ? nullptr : layers_[l - 1]->nodes[g.up_down[g.up_off[l] + s]];
                L->nodes.push_back(new Node(l == 0 ? Mmax0_ : Mmax_, lower));
            }
            for (uint32_t s = 0; s < cnt; s++) {
                const uint32_t u = l == 0 ? s : g.up_off[l] + s, c = l == 0 ? g.cnt0[s] : g.up_cnt[u];
                for (uint32_t j = 0; j < c; j++) {
                    const uint32_t no = l == 0 ? g.adj0[(size_t)s * g.deg0 + j] : g.up_adj_order[(size_t)u * g.degu + j];
                    const uint32_t ns = l == 0 ? no : g.up_adj_slot[(size_t)u * g.degu + j];
                    L->nodes[s]->neighbors.emplace_back(L->nodes[ns], 0.0f, no);
                }
            }
            L->enter_point = L->nodes[g.ep_slot[l]]; L->ep_node_order = g.ep_order[l];
            layers_.push_back(std::move(L));
        }
    }
    hnsw_amd_index* handle() { return h_; }

    // Working forms of the reference's HDF5 harness (src/hnsw.cpp:36-77 `create`, :80-156
    // `query`; upstream they no longer compile).  create: read dataset `datasetname`
    // (n x d floats) and insert every row.  query: read the query and ground-truth datasets,
    // search all queries as one GPU batch and return the reference's precision figure
    // (ground-truth top-10 found among the first 11 results).
    // normalize = true scales every row (and later every query) to unit length first: the reference has no
    // metric switch and searches angular sets as raw L2 (src/main.cpp:16,29,42 read the arrays as they are);
    // on unit vectors squared L2 ranks like the angle, which is what such a file's "neighbors" were computed with.
    static void normalizeRows(float* rows, size_t n, size_t d)
    {
        for (size_t i = 0; i < n; i++) {
            double s = 0.0;
            for (size_t j = 0; j < d; j++) s += (double)rows[i * d + j] * rows[i * d + j];
            if (s <= 0.0) continue;
            const float inv = (float)(1.0 / std::sqrt(s));
            for (size_t j = 0; j < d; j++) rows[i * d + j] *= inv;
        }
    }
    // (gpu_build: layer 0 is searched, selected and linked on GPU `device`, see insertBatchDevice)
    void create(const char* filename, const char* datasetname, int threads = 1, bool normalize = false, bool gpu_build = false, int device = 0)
    {
        hsize_t dims[2] = {0, 0};
        hdfReader::getDimensions(filename, datasetname, &dims);
        if (dims[0] == 0 || dims[1] == 0) throw std::runtime_error("create: cannot read the dataset dimensions");
        std::clog << "dimensions " << (unsigned long)dims[0] << " x " << (unsigned long)dims[1] << std::endl;
        init((uint32_t)dims[1], (uint32_t)dims[0]);
        std::vector<float> data((size_t)dims[0] * dims[1]);
        hdfReader::readData(filename, datasetname, data.data());
        if (normalize) normalizeRows(data.data(), (size_t)dims[0], (size_t)dims[1]);
        if (gpu_build) insertBatchDevice(data.data(), (uint32_t)dims[0], threads, device);
        else insertBatch(data.data(), (uint32_t)dims[0], threads);       // (keeps layers_ current when mirror_layers is set)
    }
    double query(const char* filename, const char* querydatasetname, const char* resultdatasetname, int ef, bool normalize = false)
    {
        hsize_t dq[2] = {0, 0}, dr[2] = {0, 0};
        hdfReader::getDimensions(filename, querydatasetname, &dq);
        hdfReader::getDimensions(filename, resultdatasetname, &dr);
        if (dq[0] == 0 || dr[0] != dq[0] || dq[1] != vector_size_) throw std::runtime_error("query: datasets do not fit the index");
        std::vector<float> q((size_t)dq[0] * dq[1]);
        std::vector<int> gt((size_t)dr[0] * dr[1]);
        hdfReader::readData(filename, querydatasetname, q.data());
        if (normalize) normalizeRows(q.data(), (size_t)dq[0], (size_t)dq[1]);
        hdfReader::readData(filename, resultdatasetname, gt.data());
        const uint32_t k = 10, take = k + 1, nq = (uint32_t)dq[0];
        if (dr[1] < k) throw std::runtime_error("query: ground truth narrower than k");
        std::vector<uint32_t> ids((size_t)nq * take);
        std::vector<float> dist((size_t)nq * take);
        searchBatch(q.data(), nq, ef < (int)take ? (int)take : ef, take, ids.data(), dist.data());
        double positive = 0;
        for (uint32_t i = 0; i < nq; i++)
            for (uint32_t c = 0; c < k; c++)
                for (uint32_t j = 0; j < take; j++)
                    if (ids[(size_t)i * take + j] == (uint32_t)gt[(size_t)i * dr[1] + c]) { positive++; break; }
        double precision = positive / ((double)nq * k);
        std::cout << "Precision: " << precision << ", ef: " << ef << "\n";
        return precision;
    }

private:
    hnsw_amd_index* h_ = nullptr;
    uint32_t nlayers_ = 0;
    std::vector<uint32_t> ids_;
    std::vector<float> dist_;

    static void check(int rc)
    {
        if (rc == HNSW_AMD_OK) return;
        std::string msg = hnsw_amd_last_error();
        if (rc == HNSW_AMD_E_FULL) throw std::runtime_error(msg);
        throw std::runtime_error("hnsw_amd: " + msg);
    }
    // false: the file could not be opened (reported, the caller carries on, as upstream); other failures do not return
    static bool fileCall(int rc, const char* filename)
    {
        if (rc == HNSW_AMD_OK) return true;
        const std::string msg = hnsw_amd_last_error();
        if (rc == HNSW_AMD_E_IO && msg.find("not successfully opened") != std::string::npos) {
            std::cout << "Save file " << filename << "was not sucesfully opened\n";
            return false;
        }
        std::cout << msg << "\n";
        std::exit(0);
    }
    void refresh()
    {
        int32_t M, Mmax, Mmax0, efc, maxn, act;
        uint32_t d, nl;
        check(hnsw_amd_get_info(h_, &M, &Mmax, &Mmax0, &efc, &d, &maxn, &act, &nl));
        M_ = M; Mmax_ = Mmax; Mmax0_ = Mmax0; efConstruction_ = efc; vector_size_ = d;
        max_node_count_ = maxn; actual_node_count_ = act; nlayers_ = nl;
        ml_ = (float)(1 / std::log(0.8 * M));
        min_M_ = M / 2;
        vector_parts = d / 16;                      // src/hnsw.h:593
        check(hnsw_amd_vector_data(h_, &vector_data_));
    }
};
