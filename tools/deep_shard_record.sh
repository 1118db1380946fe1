#!/bin/bash
# One full-size shard of the DEEP-96 100M configuration (12.5M x 96): the bench line, then the memory-side counter
# passes (FETCH_SIZE / WRITE_SIZE) over the same index.  usage (GPU box, repo root): bash tools/deep_shard_record.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/deep12m
mkdir -p $O /tmp/bcache
export HNSW_AMD_BENCH_CACHE=/tmp/bcache
# DEEP_BUILD=gpu: the GPU-assisted build (the host links); DEEP_BUILD=device: layer 0 linked on the GPU as well (same construction rule)
B=${DEEP_BUILD:-host}
timeout -k 10 700 python -u $R/bench.py --kind deep --base-size 12500000 --build-keys float --build $B --cpu-baseline none > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
tail -c 900 $O/bench.json; echo
SUF=$([ $B = gpu ] && echo _gb || ([ $B = device ] && echo _dl || echo ''))
export PMC_CACHE=/tmp/bcache/deep_n12500000_s1_r0_fk$SUF.npz PMC_PASSES="3 4" PMC_TIMEOUT=300
timeout -k 10 900 bash $R/tools/pmc_passes.sh deep12m/pmc --kind deep --n 12500000 --ef 30 --build-keys float
python3 $R/tools/pmc_summary.py $R/gpurun_out/deep12m/pmc > $O/pmc_summary.json
grep -h "algorithmic_bytes_per_launch" $R/gpurun_out/deep12m/pmc/p3.log | tail -1 > $O/pmc_algorithmic.txt
cat $O/pmc_summary.json | tail -12; cat $O/pmc_algorithmic.txt
