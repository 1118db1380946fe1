#!/bin/bash
# Bench lines for the other configurations (one JSON file each under gpurun_out/configs/).
# usage: bash tools/config_records.sh <set>   with set = a | b
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/configs
mkdir -p $O
run() { name=$1; shift; echo "== $name: $*"; timeout -k 10 900 python -u $R/bench.py "$@" > $O/$name.json || { echo "$name failed rc=$?"; return 1; }; tail -c 600 $O/$name.json; echo; }
case "$1" in
a)
  run sift128_1m_rows_u8 --rows u8 &&
  run fmnist784_60k --kind fmnist --base-size 60000 --ef 64 &&
  run fmnist784_60k_rows_u8 --kind fmnist --base-size 60000 --ef 64 --rows u8 --cpu-baseline none &&
  run glove100_1.18M_scalar_int32_sweep --kind glove --base-size 1183514 --dist scalar --build gpu --sweep 10,20,30,40,60,80,100,150,200 --cpu-baseline none &&
  run glove100_1.18M_scalar_float_sweep --kind glove --base-size 1183514 --dist scalar --build gpu --keys float --sweep 10,20,30,40,60,80,100,150,200,300 --cpu-baseline none
  ;;
g)
  # the GloVe-shaped configuration as full bench lines (operating point, roofline with the scalar-order gather ceiling, CPU baseline =
  # the reference's scalar build where its visited table survives the queries): index built AND searched with the scalar distance
  export HNSW_AMD_BENCH_CACHE=/tmp/bcache; mkdir -p /tmp/bcache
  run glove100_1.18M_scalar_int32 --kind glove --base-size 1183514 --dist scalar --build gpu --target-recall 0.95 &&
  run glove100_1.18M_scalar_float --kind glove --base-size 1183514 --dist scalar --build gpu --keys float --target-recall 0.95 --cpu-baseline port
  ;;
b)
  run gist960_1m --kind gist --base-size 1000000 --queries 1000 --cpu-baseline none &&
  run deep96_4m_single_shard --kind deep --base-size 4000000 --cpu-baseline none
  ;;
d)
  run deep96_12.5M_single_shard_float_build_keys --kind deep --base-size 12500000 --build-keys float --cpu-baseline none
  ;;
e)
  run sift128_1m_reference_ef_sweep --sweep 20,30,40,50,60,70,80,90,100,120,140,160,180,200 --cpu-baseline none
  ;;
c)
  run glove100_1.18M_sse_float_sweep --kind glove --base-size 1183514 --keys float --sweep 20,40,60,100 --cpu-baseline none &&
  run glove100_1.18M_sse_float_q8_sweep --kind glove --base-size 1183514 --keys float --rows q8 --sweep 20,40,60,100 --cpu-baseline none &&
  run sift128_1m_q8_sweep --rows q8 --keys float --sweep 30,40,50,60 --cpu-baseline none &&
  run sift128_1m_float_sweep --keys float --sweep 30,40,50,60 --cpu-baseline none
  ;;
esac
