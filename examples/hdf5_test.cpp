// The reference's HDF5 mode (src/main.cpp:140-160 under MAIN_RUN_CREATE_AND_QUERY: create(file, "train"), then
// query(file, "test", "neighbors", ef) for a list of ef values) written against include/hnsw_amd.hpp.
//
//   hdf5_test <file.hdf5> [build threads] [l2|angular|l2+gpu|angular+gpu] [ef ...]
// "angular": rows and queries are scaled to unit length on load (the library's addition, see hnsw_amd.hpp).
// "+gpu": layer 0 of the index is built on the GPU (needs build threads > 1).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hnsw_amd.hpp"

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s <file.hdf5> [threads] [l2|angular|l2+gpu|angular+gpu] [ef ...]\n", argv[0]); return 2; }
    const char* file = argv[1];
    const int threads = argc > 2 ? std::atoi(argv[2]) : 1;
    const bool angular = argc > 3 && std::strncmp(argv[3], "angular", 7) == 0;
    const bool gpu_build = argc > 3 && std::strstr(argv[3], "+gpu") != nullptr;
    std::vector<int> efs;
    for (int i = 4; i < argc; i++) efs.push_back(std::atoi(argv[i]));
    if (efs.empty()) efs = {20, 50, 100};
    try {
        HNSW hnsw(16, 16, 200);
        hnsw.create(file, "train", threads, angular, gpu_build);
        for (int ef : efs) hnsw.query(file, "test", "neighbors", ef, angular);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
