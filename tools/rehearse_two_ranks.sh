set -e
export HNSW_AMD_BENCH_BACKEND=gloo
export HNSW_AMD_BENCH_WATCHDOG=${HNSW_AMD_BENCH_WATCHDOG:-120}      # a hang prints the stacks of both ranks
cd $GRAFT_REPO_ROOT
for mode in sharded replicated; do
  echo "== mode $mode"
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 30 --warmup 5 --base-size 200000 --queries 4000 --build-threads 8 --mode $mode 2>gpurun_out/rehearse_$mode.err | tee gpurun_out/rehearse_$mode.json
done
