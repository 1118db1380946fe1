#!/bin/bash
# The full-size DEEP-shaped shard under several bulk fractions of the bucket visited set (HNSW_AMD_BULK_FRAC): one build, cached.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/deep_frac
mkdir -p $O /tmp/bcache
export HNSW_AMD_BENCH_CACHE=/tmp/bcache
for f in 0.45 0.3 0.2 0.6; do
  HNSW_AMD_BULK_FRAC=$f timeout -k 10 700 python -u $R/bench.py --kind deep --base-size 12500000 --build-keys float --build gpu --cpu-baseline none --ef 30 > $O/bench_$f.json 2> $O/bench_$f.err || { echo "frac $f failed"; tail -3 $O/bench_$f.err; exit 1; }
  python3 -c "
import json
j=json.loads([x for x in open('$O/bench_$f.json') if x.startswith('{')][-1])
r=j['roofline']; print('frac $f:', j['ms_per_step'], 'ms', r['achieved'], 'GB/s', r['launch'], 'large:', j['large_batch']['ms_per_step'], j['large_batch']['achieved_gbs'])
"
done
