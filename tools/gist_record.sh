#!/bin/bash
# The GIST-960-shaped set (1M x 960, a 3.84 GB table) with a full batch of 10 000 queries: the bench line, then the
# memory-side counter passes over the same index; and the configuration's own 1 000-query batch.
# usage (GPU box, repo root): bash tools/gist_record.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/gist1m
mkdir -p $O /tmp/bcache
export HNSW_AMD_BENCH_CACHE=/tmp/bcache
timeout -k 10 600 python -u $R/bench.py --kind gist --base-size 1000000 --queries 10000 --cpu-baseline none > $O/bench_10k.json 2> $O/bench_10k.err || { echo "bench failed"; tail -5 $O/bench_10k.err; exit 1; }
tail -c 700 $O/bench_10k.json; echo
timeout -k 10 300 python -u $R/bench.py --kind gist --base-size 1000000 --queries 1000 --cpu-baseline none > $O/bench_1k.json 2> $O/bench_1k.err || { echo "bench 1k failed"; exit 1; }
EF=$(python3 -c "import json; print(json.load(open('$O/bench_10k.json'))['config']['ef'])")
export PMC_CACHE=/tmp/bcache/gist_n1000000_s1_r0.npz PMC_PASSES="3 4" PMC_TIMEOUT=300
timeout -k 10 900 bash $R/tools/pmc_passes.sh gist1m/pmc --kind gist --n 1000000 --ef $EF
python3 $R/tools/pmc_summary.py $R/gpurun_out/gist1m/pmc > $O/pmc_summary.json
grep -h "algorithmic_bytes_per_launch" $R/gpurun_out/gist1m/pmc/p3.log | tail -1 > $O/pmc_algorithmic.txt
tail -8 $O/pmc_summary.json; cat $O/pmc_algorithmic.txt
