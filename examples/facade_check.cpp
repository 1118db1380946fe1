// Exercises the public data members of the reference's `class HNSW` (src/hnsw.h:221-245) as
// include/hnsw_amd.hpp serves them: stat_, visited_, vector_data_, layers_, print(int), and the
// file calls' error behaviour.  Prints "facade ok" and returns 0 when everything holds.
//
//   facade_check <scratch_dir>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "hnsw_amd.hpp"

#define EXPECT(cond) do { if (!(cond)) { std::fprintf(stderr, "facade_check: %s failed (line %d)\n", #cond, __LINE__); return 1; } } while (0)

int main(int argc, char** argv)
{
    const std::string dir = argc > 1 ? argv[1] : ".";
    const uint32_t n = 2000, d = 20;      // 20 dims: the SSE-order distance uses 16 of them, the scalar one all
    std::vector<float> rows((size_t)n * d);
    uint32_t x = 12345;
    for (auto& v : rows) { x = x * 1664525u + 1013904223u; v = (float)((x >> 16) % 97); }

    HNSW hnsw(8, 8, 60);
    EXPECT(hnsw.min_M_ == 4 && hnsw.Mmax0_ == 16 && hnsw.vector_data_ == nullptr && hnsw.data_cleaned_);
    hnsw.init(d, n);
    EXPECT(!hnsw.data_cleaned_ && hnsw.vector_parts == 1 && hnsw.vector_data_ != nullptr);
    for (uint32_t i = 0; i < n; i++) hnsw.insert(&rows[(size_t)i * d]);
    EXPECT(hnsw.actual_node_count_ == (int)n - 1);
    EXPECT(std::memcmp(hnsw.vector_data_, rows.data(), rows.size() * sizeof(float)) == 0);

    // stat_: counts in the scalar build only, as upstream (src/hnsw.h:390); cleared by clear()
    hnsw.stat_.clear();
    hnsw.aproximateKnn(&rows[5 * d], 10, 40);
    EXPECT(hnsw.W_.size() == 40 && hnsw.W_[0].node_order == 5 && hnsw.W_[0].distance == 0.0f && hnsw.W_[0].node == nullptr);
    EXPECT(hnsw.stat_.precise_distance_computations == 0);
    EXPECT(hnsw.visited_.item_count == hnsw.last_query_stats.n_visited0 && hnsw.visited_.item_count >= 40);
    hnsw.search_flags |= HNSW_AMD_DIST_SCALAR;
    int total = 0;
    for (uint32_t i = 0; i < 5; i++) { hnsw.aproximateKnn(&rows[(size_t)i * d], 10, 40); total += (int)hnsw.last_query_stats.n_dist; }
    EXPECT(total > 0 && hnsw.stat_.precise_distance_computations == total);
    hnsw.stat_.print();
    hnsw.stat_.print_tree_info();
    hnsw.stat_.clear();
    EXPECT(hnsw.stat_.precise_distance_computations == 0 && hnsw.stat_.explored_nodes == 0);

    // layers_: the reference's object graph; node_order i is layers_[0]->nodes[i]
    hnsw.syncLayers();
    EXPECT(!hnsw.layers_.empty() && hnsw.layers_[0]->nodes.size() == n && hnsw.layers_[0]->node_count == n);
    hnsw_amd_flat_graph g;
    EXPECT(hnsw_amd_export_flat(hnsw.handle(), &g, nullptr) == HNSW_AMD_OK);
    EXPECT(hnsw.layers_.size() == g.nlayers);
    for (uint32_t i = 0; i < n; i++) {
        const Node* node = hnsw.layers_[0]->nodes[i];
        EXPECT(node->neighbors.size() == g.cnt0[i] && node->lower_layer == nullptr);
        for (uint32_t j = 0; j < g.cnt0[i]; j++) {
            EXPECT(node->neighbors[j].node_order == g.adj0[(size_t)i * g.deg0 + j]);
            EXPECT(node->neighbors[j].node == hnsw.layers_[0]->nodes[node->neighbors[j].node_order]);
        }
    }
    for (uint32_t l = 1; l < g.nlayers; l++) {
        const Layer& L = *hnsw.layers_[l];
        EXPECT(L.nodes.size() == g.up_off[l + 1] - g.up_off[l] && L.ep_node_order == g.ep_order[l]);
        EXPECT(L.enter_point == L.nodes[g.ep_slot[l]]);
        for (size_t s = 0; s < L.nodes.size(); s++) {
            // following lower_layer down to layer 0 arrives at the vector's own node
            const Node* p = L.nodes[s];
            for (uint32_t k = l; k > 0; k--) { EXPECT(p->lower_layer != nullptr); p = p->lower_layer; }
            EXPECT(p == hnsw.layers_[0]->nodes[g.up_order[g.up_off[l] + s]]);
        }
    }
    hnsw.print(3);
    hnsw.printInfo(false);

    // files: round trip; an unopenable file is reported and the call returns, the index unchanged
    const std::string path = dir + "/facade_graph.bin";
    hnsw.saveGraph(path.c_str());
    HNSW other(8, 8, 60);
    other.mirror_layers = true;
    other.loadGraph(path.c_str());
    EXPECT(other.actual_node_count_ == (int)n - 1 && other.layers_.size() == hnsw.layers_.size());
    other.loadGraph((dir + "/no/such/file.bin").c_str());
    EXPECT(other.actual_node_count_ == (int)n - 1);
    other.search_flags = hnsw.search_flags;
    other.aproximateKnn(&rows[7 * d], 10, 30);
    hnsw.aproximateKnn(&rows[7 * d], 10, 30);
    EXPECT(other.W_.size() == hnsw.W_.size());
    for (size_t i = 0; i < hnsw.W_.size(); i++) EXPECT(other.W_[i].distance == hnsw.W_[i].distance);

    // the batched calls of the class: an index whose layer 0 is built on the GPU, under the reference's plain (all-dims)
    // distance, searched as one batch: every stored row finds itself first
    {
        const uint32_t nb = 6000;
        std::vector<float> big((size_t)nb * d);
        for (auto& v : big) { x = x * 1664525u + 1013904223u; v = (float)((x >> 16) % 251); }
        HNSW built(16, 16, 100);
        built.init(d, nb);
        built.setBuildScalarDistance(true);
        built.search_flags |= HNSW_AMD_DIST_SCALAR;
        built.insertBatchDevice(big.data(), nb, 4);
        EXPECT(built.actual_node_count_ == (int)nb - 1);
        const uint32_t nq = 500, take = 3;
        std::vector<uint32_t> ids((size_t)nq * take);
        std::vector<float> dist((size_t)nq * take);
        built.searchBatch(big.data(), nq, 40, take, ids.data(), dist.data());
        for (uint32_t i = 0; i < nq; i++) EXPECT(dist[(size_t)i * take] == 0.0f);
    }

    // linearHash as a container (src/hnsw.h:55-125)
    linearHash set(64);
    set.clear();
    set.insert(3); set.insert(67); set.insert(131);
    EXPECT(set.item_count == 3 && set.get(67) && set.get(131) && !set.get(195));
    std::printf("facade ok\n");
    return 0;
}
