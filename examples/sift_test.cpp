// The reference's default driver (src/main.cpp:167-319, `sift_test`) on top of
// include/hnsw_amd.hpp: raw .bin inputs, build or loadGraph, the ef sweep with the
// "Precision" figure (ground-truth top-k found among the first k+1 results) and three timed
// passes per ef.  The reference hard-codes sizes and file names; here they are arguments:
//
//   sift_test <base.bin> <n> <d> <queries.bin> <nq> <gt.bin> <gt_width> [load_graph.bin] [build_threads]
//
// Queries go to the GPU one at a time through aproximateKnn (the reference's loop), then
// once more as one batch (searchBatch), which is how the GPU is meant to be used.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "hnsw_amd.hpp"

template <class T>
static std::vector<T> read_bin(const char* path, size_t count)
{
    std::vector<T> v(count);
    std::ifstream in(path, std::ios::binary);
    if (!in.is_open()) throw std::runtime_error(std::string("Input file not opened: ") + path);
    in.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(count * sizeof(T)));
    if ((size_t)in.gcount() != count * sizeof(T)) throw std::runtime_error(std::string("short read: ") + path);
    return v;
}

int main(int argc, char** argv)
{
    if (argc < 8) { std::fprintf(stderr, "usage: see the header of sift_test.cpp\n"); return 2; }
    const size_t node_count = std::atoll(argv[2]), vecdim = std::atoll(argv[3]);
    const size_t qsize = std::atoll(argv[5]), answer_size = std::atoll(argv[7]);
    const uint32_t k = 10;
    auto mass = read_bin<float>(argv[1], node_count * vecdim);
    auto massQ = read_bin<float>(argv[4], qsize * vecdim);
    auto massQA = read_bin<unsigned int>(argv[6], qsize * answer_size);

    HNSW hnsw(16, 16, 200);
    if (argc > 8 && argv[8][0] != '-') {
        hnsw.loadGraph(argv[8]);
    } else {
        hnsw.init((uint32_t)vecdim, (uint32_t)node_count);
        std::cout << "Start inserting\n";
        auto start = std::chrono::system_clock::now();
        int threads = argc > 9 ? std::atoi(argv[9]) : 1;
        if (threads > 1) hnsw.insertBatch(mass.data(), (uint32_t)node_count, threads);
        else for (size_t i = 0; i < node_count; i++) hnsw.insert(&mass[i * vecdim]);
        auto end = std::chrono::system_clock::now();
        std::cout << "Insert time " << std::chrono::duration<double>(end - start).count() << " [s] \n";
    }
    hnsw.printInfo(false);
    hnsw.syncDevice(0);
    std::cout << "Start querying\n";
    std::vector<uint32_t> ids(qsize * (k + 1));
    std::vector<float> dist(qsize * (k + 1));
    for (int ef = 20; ef <= 200; ef += 10) {
        if (ef > 100) ef += 10;
        // one query at a time, as the reference's loop
        double positive = 0;
        auto t_one = std::chrono::steady_clock::now();
        for (size_t i = 0; i < qsize; i++) {
            hnsw.aproximateKnn(&massQ[i * vecdim], k, ef);
            std::vector<uint32_t> result;
            uint32_t c1 = 0;
            for (auto item : hnsw.W_) { result.push_back(item.node_order); if (c1++ >= k) break; }
            for (uint32_t c2 = 0; c2 < k; c2++)
                if (std::find(result.begin(), result.end(), massQA[i * answer_size + c2]) != result.end()) positive++;
        }
        const double us_one = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_one).count() / qsize;
        std::cout << "Precision: " << positive / (qsize * k) << ", ef: " << ef << ", one query per call: " << us_one << " [us], ";
        // batched: three timed passes, minimum
        double best = 1e30, positive_b = 0;
        for (int pass = 0; pass < 3; pass++) {
            auto a = std::chrono::steady_clock::now();
            hnsw.searchBatch(massQ.data(), (uint32_t)qsize, ef, k + 1, ids.data(), dist.data());
            auto b = std::chrono::steady_clock::now();
            best = std::min(best, std::chrono::duration<double, std::micro>(b - a).count());
        }
        for (size_t i = 0; i < qsize; i++)
            for (uint32_t c2 = 0; c2 < k; c2++)
                if (std::find(&ids[i * (k + 1)], &ids[(i + 1) * (k + 1)], massQA[i * answer_size + c2]) != &ids[(i + 1) * (k + 1)]) positive_b++;
        std::cout << "batched precision: " << positive_b / (qsize * k) << ", min: " << best / qsize << " [us] per query (batched, host to host)\n";
    }
    return 0;
}
