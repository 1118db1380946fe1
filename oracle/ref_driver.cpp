// TEST INFRASTRUCTURE — builds only inside the development container.
//
// Driver around the UNMODIFIED reference sources (compiled from where they lie,
// /root/reference/src/{hnsw,layer,node}.cpp, see oracle/Makefile; nothing of
// them is copied here).  It builds an index with the reference's own
// HNSW::insert, dumps the pointer graph as flat arrays, runs the reference's
// own HNSW::aproximateKnn and dumps the full W_ of every query.  Its outputs
// are the golden vectors under tests/golden/ and the "reference" CPU baseline.
//
//   ref_driver build <base.bin> <n> <d> <M> <Mmax> <efC> <graph.out> [<savegraph.out>]
//   ref_driver query <base.bin> <n> <d> <M> <Mmax> <efC> <queries.bin> <nq> <results.out> <ef> [<ef>...]
//   ref_driver time  <base.bin> <n> <d> <M> <Mmax> <efC> <queries.bin> <nq> <ef> <seconds>
//
//   ref_driver lquery <graph.savegraph> <M> <Mmax> <efC> <queries.bin> <nq> <results.out> <ef> [<ef>...]
//   ref_driver ltime  <graph.savegraph> <M> <Mmax> <efC> <queries.bin> <nq> <ef> <seconds>
//   ref_driver lprecision <graph.savegraph> <M> <Mmax> <efC> <queries.bin> <nq> <gt.bin> <gt_width> <ef> [<ef>...]
//
// "lprecision" is the reference driver's "Precision" figure (src/main.cpp:236-280) computed from the reference's
// own aproximateKnn and W_: the first k+1 = 11 entries of W_ are kept (the loop there tests `c1++ >= k` after the
// push), and every one of the first k = 10 ground-truth ids (gt.bin: nq x gt_width u32) found among them counts;
// one JSON line per ef on stderr.
//
// "lquery"/"ltime" read the index with the reference's own HNSW::loadGraph (from a file in its
// saveGraph format, e.g. one written by libhnsw_amd.so) instead of building it.
// "query" and "time" build the index first (the build is deterministic: levels
// come from rand() with the default seed, reference src/hnsw.cpp:165).
//
// graph.out  : u32 magic 'HGF1', n, d, M, Mmax, Mmax0, nlayers;
//              per layer: u32 n_nodes, ep_slot, ep_node_order;
//                per node: u32 node_order, down_slot, degree, then degree x (u32 slot, u32 node_order)
// results.out: u32 magic 'HRS1', nq, n_ef; per ef: u32 ef; per query: u32 count,
//              count x u32 node_order, count x f32 distance (W_ after the reference's sort)
#include "hnsw.h"

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

static std::vector<float> read_floats(const char* path, size_t count) {
    std::vector<float> v(count);
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    size_t got = fread(v.data(), sizeof(float), count, f);
    fclose(f);
    if (got != count) { fprintf(stderr, "%s: short read (%zu of %zu floats)\n", path, got, count); exit(2); }
    return v;
}

static void w32(FILE* f, uint32_t v) { fwrite(&v, 4, 1, f); }

static void dump_graph(HNSW& h, uint32_t n, const char* path) {
    FILE* f = fopen(path, "wb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    const uint32_t nl = (uint32_t)h.layers_.size();
    w32(f, 0x31464748u); w32(f, n); w32(f, h.vector_size_);
    w32(f, (uint32_t)h.M_); w32(f, (uint32_t)h.Mmax_); w32(f, (uint32_t)h.Mmax0_); w32(f, nl);
    std::vector<std::unordered_map<Node*, uint32_t>> slot(nl);
    for (uint32_t l = 0; l < nl; l++) {
        auto& nodes = h.layers_[l]->nodes;
        for (uint32_t s = 0; s < nodes.size(); s++) slot[l][nodes[s]] = s;
    }
    // node_order of an upper-layer node: follow lower_layer down to layer 0,
    // where the slot is the node_order.
    auto order_of = [&](Node* nd, uint32_t l) {
        while (l > 0) { nd = nd->lower_layer; l--; }
        return slot[0].at(nd);
    };
    for (uint32_t l = 0; l < nl; l++) {
        auto& L = *h.layers_[l];
        w32(f, (uint32_t)L.nodes.size()); w32(f, slot[l].at(L.enter_point)); w32(f, L.ep_node_order);
        for (Node* nd : L.nodes) {
            w32(f, order_of(nd, l));
            w32(f, l == 0 ? 0xFFFFFFFFu : slot[l - 1].at(nd->lower_layer));
            w32(f, (uint32_t)nd->neighbors.size());
            for (auto& ne : nd->neighbors) { w32(f, slot[l].at(ne.node)); w32(f, ne.node_order); }
        }
    }
    fclose(f);
}

static int run_queries(HNSW& h, uint32_t d, const char* qpath, uint32_t nq, const char* out, int argc, char** argv, int first_ef) {
    auto q = read_floats(qpath, (size_t)nq * d);
    FILE* f = fopen(out, "wb");
    if (!f) return 2;
    w32(f, 0x31535248u); w32(f, nq); w32(f, (uint32_t)(argc - first_ef));
    for (int a = first_ef; a < argc; a++) {
        int ef = atoi(argv[a]);
        w32(f, (uint32_t)ef);
        for (uint32_t i = 0; i < nq; i++) {
            h.aproximateKnn(&q[(size_t)i * d], 10, ef);
            w32(f, (uint32_t)h.W_.size());
            for (auto& e : h.W_) w32(f, e.node_order);
            for (auto& e : h.W_) fwrite(&e.distance, 4, 1, f);
        }
    }
    fclose(f);
    return 0;
}

// the reference's protocol (src/main.cpp:285-304): passes over all queries, minimum kept
static int run_timing(HNSW& h, uint32_t d, const char* qpath, uint32_t nq, int ef, double budget) {
    auto q = read_floats(qpath, (size_t)nq * d);
    double best = 1e30; int passes = 0;
    auto start = std::chrono::steady_clock::now();
    do {
        auto a = std::chrono::steady_clock::now();
        for (uint32_t i = 0; i < nq; i++) h.aproximateKnn(&q[(size_t)i * d], 10, ef);
        auto b = std::chrono::steady_clock::now();
        best = std::min(best, std::chrono::duration<double>(b - a).count());
        passes++;
    } while (passes < 3 && std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count() < budget);
    fprintf(stderr, "{\"ref_time_s\": %.6f, \"nq\": %u, \"ef\": %d, \"passes\": %d, \"qps\": %.1f}\n",
            best, nq, ef, passes, nq / best);
    return 0;
}

static int run_precision(HNSW& h, uint32_t d, const char* qpath, uint32_t nq, const char* gtpath, uint32_t width,
                         int argc, char** argv, int first_ef) {
    const uint32_t k = 10;
    if (width < k) { fprintf(stderr, "ground truth narrower than k\n"); return 2; }
    auto q = read_floats(qpath, (size_t)nq * d);
    std::vector<uint32_t> gt((size_t)nq * width);
    FILE* f = fopen(gtpath, "rb");
    if (!f || fread(gt.data(), 4, gt.size(), f) != gt.size()) { fprintf(stderr, "cannot read %s\n", gtpath); return 2; }
    fclose(f);
    for (int a = first_ef; a < argc; a++) {
        const int ef = atoi(argv[a]);
        double positive = 0;
        for (uint32_t i = 0; i < nq; i++) {
            h.aproximateKnn(&q[(size_t)i * d], (int)k, ef);
            const size_t take = std::min<size_t>(h.W_.size(), k + 1);
            for (uint32_t c = 0; c < k; c++) {
                const uint32_t want = gt[(size_t)i * width + c];
                for (size_t j = 0; j < take; j++)
                    if (h.W_[j].node_order == want) { positive++; break; }
            }
        }
        fprintf(stderr, "{\"ref_precision\": %.6f, \"nq\": %u, \"ef\": %d}\n", positive / ((double)nq * k), nq, ef);
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 9) { fprintf(stderr, "usage: see header of ref_driver.cpp\n"); return 2; }
    std::string mode = argv[1];
    if (mode == "lquery" || mode == "ltime" || mode == "lprecision") {
        HNSW h(atoi(argv[3]), atoi(argv[4]), atoi(argv[5]));
        fflush(stdout);
        FILE* devnull = freopen("/dev/null", "w", stdout);
        (void)devnull;
        h.loadGraph(argv[2]);
        if (h.layers_.empty()) { fprintf(stderr, "loadGraph(%s) failed\n", argv[2]); return 2; }
        uint32_t nq = (uint32_t)atoll(argv[7]);
        if (mode == "lprecision") {
            if (argc < 11) return 2;
            return run_precision(h, h.vector_size_, argv[6], nq, argv[8], (uint32_t)atoll(argv[9]), argc, argv, 10);
        }
        if (mode == "lquery") {
            if (argc < 10) return 2;
            return run_queries(h, h.vector_size_, argv[6], nq, argv[8], argc, argv, 9);
        }
        if (argc < 10) return 2;
        return run_timing(h, h.vector_size_, argv[6], nq, atoi(argv[8]), atof(argv[9]));
    }
    const char* base_path = argv[2];
    uint32_t n = (uint32_t)atoll(argv[3]), d = (uint32_t)atoll(argv[4]);
    int M = atoi(argv[5]), Mmax = atoi(argv[6]), efC = atoi(argv[7]);
    auto base = read_floats(base_path, (size_t)n * d);

    HNSW h(M, Mmax, efC);
    h.init(d, n);
    auto t0 = std::chrono::steady_clock::now();
    fflush(stdout);
    FILE* devnull = freopen("/dev/null", "w", stdout);  // the reference prints a progress line per 10000 inserts
    (void)devnull;
    for (uint32_t i = 0; i < n; i++) h.insert(&base[(size_t)i * d]);
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "ref build: n=%u d=%u layers=%zu %.2f s\n", n, d, h.layers_.size(),
            std::chrono::duration<double>(t1 - t0).count());

    if (mode == "build") {
        dump_graph(h, n, argv[8]);
        if (argc > 9) h.saveGraph(argv[9]);
        return 0;
    }
    if (mode == "query") {
        if (argc < 12) return 2;
        return run_queries(h, d, argv[8], (uint32_t)atoll(argv[9]), argv[10], argc, argv, 11);
    }
    if (mode == "time") {
        if (argc < 12) return 2;
        return run_timing(h, d, argv[8], (uint32_t)atoll(argv[9]), atoi(argv[10]), atof(argv[11]));
    }
    return 2;
}
