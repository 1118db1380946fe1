// Host-only driver for sanitizer runs of the index builder (no GPU code linked).
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../hnsw_amd/csrc/hnsw_host.h"
using namespace hnsw_amd;
int main(int argc, char** argv)
{
    const uint32_t n = argc > 1 ? atoi(argv[1]) : 4000, d = 32;
    const int threads = argc > 2 ? atoi(argv[2]) : 8;
    std::mt19937 rng(1);
    std::uniform_int_distribution<int> u(0, 200);
    std::vector<float> rows((size_t)n * d);
    for (auto& x : rows) x = (float)u(rng);
    for (int rep = 0; rep < 3; rep++) {
        HostIndex h(32, 32, 60);
        h.init(d, n);
        h.insert_batch(rows.data(), n / 2, threads);
        h.insert_batch(rows.data() + (size_t)(n / 2) * d, n - n / 2, threads);
        FlatGraph g;
        h.export_flat(g);
        h.save_graph("/tmp/san_graph.bin");
        HostIndex h2(32, 32, 60);
        h2.load_graph("/tmp/san_graph.bin");
        FlatGraph g2;
        h2.export_flat(g2);
        if (g2.adj0 != g.adj0) { std::puts("save/load mismatch"); return 1; }
        HostIndex h3(32, 32, 60);
        h3.import_flat(g, rows.data());
        std::printf("rep %d: n=%u layers=%u\n", rep, g.n, g.nlayers);
    }
    return 0;
}
