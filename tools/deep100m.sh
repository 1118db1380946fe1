#!/bin/bash
# The DEEP-96 100M configuration as one command: 8 ranks (one per GPU) over RCCL, 12.5M rows each, layer 0 built on the GPU
# (search, selection, linking) with float construction keys, search with the reference's int32 keys, all-gather + device merge.
#   bash tools/deep100m.sh                 the configuration itself (8-GPU node; about 16 GB of host memory per rank)
#   bash tools/deep100m.sh rehearse        the same code path with 2 ranks over gloo sharing one GPU, 400k rows
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
if [ "${1:-}" = rehearse ]; then
  export HNSW_AMD_BENCH_BACKEND=gloo HNSW_AMD_BENCH_PIPELINE_GLOO=1 HNSW_AMD_BENCH_WATCHDOG=${HNSW_AMD_BENCH_WATCHDOG:-300}
  exec timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
       bench.py --gpus 2 --preset deep100m --base-size 400000 --queries 2000 --steps 10 --warmup 2 --build-threads ${BUILD_THREADS:-4}
fi
N=${GPUS:-8}
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port ${MASTER_PORT:-29540} \
     bench.py --gpus $N --preset deep100m --steps ${STEPS:-20} --warmup ${WARMUP:-3}
