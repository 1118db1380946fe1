// hnsw_amd.hpp — the reference's `class HNSW` (RadimBaca/HNSW src/hnsw.h:217-603) as a thin
// header-only wrapper over the C ABI of libhnsw_amd.so.  Same constructor, member functions
// and public data member names, so the reference's driver (src/main.cpp:197-308) compiles
// against it unchanged apart from the include line:
//
//     HNSW hnsw(16, 16, 200);  hnsw.init(d, n);  hnsw.insert(v);  hnsw.loadGraph(f);
//     hnsw.aproximateKnn(q, k, ef);  for (auto item : hnsw.W_) use(item.node_order);
//
// Searches run on the GPU (MI355X); there is no CPU search behind this class.  Errors that the
// reference reports by exception (insert beyond capacity: std::runtime_error,
// src/hnsw.cpp:176-179) are exceptions here too; saveGraph/loadGraph print a message and
// return on an unopenable file like the reference (src/hnsw.cpp:1183-1187,1248-1251).
#pragma once

#include <cmath>
#include <cstdint>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "hnsw_amd.h"

class Node;   // the reference's W_ entries carry a Node*; it is always null here

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
    uint32_t node_order;
    Node* node;
};

class HNSW {
public:
    // public data members the reference's callers read (src/hnsw.h:221-238)
    int M_, Mmax_, Mmax0_, efConstruction_;
    float ml_;
    int max_node_count_ = 0;
    int actual_node_count_ = -1;
    uint32_t vector_size_ = 0;
    std::vector<Neighbors> W_;
    uint32_t search_flags = HNSW_AMD_DIST_SSE | HNSW_AMD_KEYS_INT32;   // see hnsw_amd.h

    HNSW(int M, int Mmax, int efConstruction) : M_(M), Mmax_(Mmax), Mmax0_(Mmax * 2), efConstruction_(efConstruction)
    {
        check(hnsw_amd_create(M, Mmax, efConstruction, &h_));
        refresh();
    }
    ~HNSW() { hnsw_amd_destroy(h_); }
    HNSW(const HNSW&) = delete;
    HNSW& operator=(const HNSW&) = delete;

    void init(uint32_t vector_size, uint32_t max_node_count) { check(hnsw_amd_init(h_, vector_size, max_node_count)); refresh(); }
    void clean() {}
    void insert(float* q) { check(hnsw_amd_insert(h_, q)); actual_node_count_++; }
    // the driver's insert loop (src/main.cpp:207-210) in one call; threads > 1 builds concurrently
    void insertBatch(const float* rows, uint32_t n, int threads) { check(hnsw_amd_insert_batch(h_, rows, n, threads)); refresh(); }
    // construction searches keyed by the float distance instead of the int32 truncation (see hnsw_amd.h)
    void setBuildFloatKeys(bool on) { check(hnsw_amd_set_build_float_keys(h_, on ? 1 : 0)); }

    // src/hnsw.cpp:373-449: result in W_, up to ef entries ascending; k is ignored as upstream
    void aproximateKnn(float* q, int k, int ef)
    {
        ids_.resize(ef > 0 ? ef : 1); dist_.resize(ids_.size());
        uint32_t cnt = 0;
        check(hnsw_amd_aproximate_knn(h_, q, k, ef, search_flags, ids_.data(), dist_.data(), &cnt));
        W_.resize(cnt);
        for (uint32_t i = 0; i < cnt; i++) W_[i] = Neighbors{dist_[i], ids_[i], nullptr};
    }
    // nq queries at once: first out_k entries of every query's W_ (row-major nq x out_k)
    void searchBatch(const float* queries, uint32_t nq, int ef, uint32_t out_k, uint32_t* out_ids, float* out_dist,
                     uint32_t* out_count = nullptr, hnsw_amd_query_stats* stats = nullptr)
    {
        check(hnsw_amd_search_batch(h_, queries, nq, ef, out_k, search_flags, out_ids, out_dist, out_count, stats));
    }
    void syncDevice(int device = 0) { check(hnsw_amd_sync_device(h_, device)); }

    void saveGraph(const char* filename)
    {
        if (hnsw_amd_save_graph(h_, filename) != HNSW_AMD_OK) std::cout << hnsw_amd_last_error() << "\n";
    }
    void loadGraph(const char* filename)
    {
        if (hnsw_amd_load_graph(h_, filename) != HNSW_AMD_OK) std::cout << hnsw_amd_last_error() << "\n";
        refresh();
    }
    // src/hnsw.cpp:1319-1339
    void printInfo(bool /*all*/)
    {
        std::cout << "HNSW info\n";
        std::cout << "Layers: " << nlayers_ << "\n";
        std::cout << "M_: " << M_ << "\n" << "Mmax_: " << Mmax_ << "\n" << "ml_: " << ml_ << "\n";
        std::cout << "efConstruction_: " << efConstruction_ << "\n";
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
    void create(const char* filename, const char* datasetname, int threads = 1, bool normalize = false)
    {
        hsize_t dims[2] = {0, 0};
        hdfReader::getDimensions(filename, datasetname, &dims);
        if (dims[0] == 0 || dims[1] == 0) throw std::runtime_error("create: cannot read the dataset dimensions");
        std::clog << "dimensions " << (unsigned long)dims[0] << " x " << (unsigned long)dims[1] << std::endl;
        init((uint32_t)dims[1], (uint32_t)dims[0]);
        std::vector<float> data((size_t)dims[0] * dims[1]);
        hdfReader::readData(filename, datasetname, data.data());
        if (normalize) normalizeRows(data.data(), (size_t)dims[0], (size_t)dims[1]);
        insertBatch(data.data(), (uint32_t)dims[0], threads);
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
    void refresh()
    {
        int32_t M, Mmax, Mmax0, efc, maxn, act;
        uint32_t d, nl;
        check(hnsw_amd_get_info(h_, &M, &Mmax, &Mmax0, &efc, &d, &maxn, &act, &nl));
        M_ = M; Mmax_ = Mmax; Mmax0_ = Mmax0; efConstruction_ = efc; vector_size_ = d;
        max_node_count_ = maxn; actual_node_count_ = act; nlayers_ = nl;
        ml_ = (float)(1 / std::log(0.8 * M));
    }
};
