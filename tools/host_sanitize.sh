#!/bin/bash
# Host code only (no GPU): the index builder, export/import and the graph file under ThreadSanitizer and under
# AddressSanitizer + UBSan.  usage: bash tools/host_sanitize.sh   (from the repo root)
set -e
mkdir -p build/san
F="-std=c++17 -O1 -g -mavx2 -mfma -ffp-contract=off tools/host_sanitizer_driver.cpp hnsw_amd/csrc/hnsw_host.cpp -lpthread"
g++ -fsanitize=thread $F -o build/san/tsan_driver
g++ -fsanitize=address,undefined -fno-omit-frame-pointer $F -o build/san/asan_driver
./build/san/tsan_driver 3000 8
./build/san/asan_driver 6000 8
./build/san/asan_driver 3000 1
echo "sanitizers: clean"
