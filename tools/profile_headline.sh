set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_r3
cd /tmp
export HNSW_AMD_BENCH_CACHE=/tmp/bcache
mkdir -p /tmp/bcache
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r3 -- python3 $R/bench.py --steps 10 --cpu-baseline none > $R/gpurun_out/prof_r3/bench_under_rocprof.json 2> $R/gpurun_out/prof_r3/err.log
cd $R
timeout -k 10 900 bash tools/pmc_passes.sh pmc_r3
python3 tools/pmc_summary.py gpurun_out/pmc_r3 > gpurun_out/pmc_r3/summary.json
timeout -k 10 400 python -u bench.py 2>/dev/null | tee gpurun_out/bench_final_r3.json
