#!/bin/bash
# as ab_rows.sh, for the byte rows (SIFT-shaped 1M): kernel for any row length against the 128-dim kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=/tmp/ab_sift_1000000.npz
for mode in any shaped any shaped; do
  unset HNSW_AMD_GENERIC_ROWS
  if [ $mode = any ]; then export HNSW_AMD_GENERIC_ROWS=1; fi
  HNSW_AMD_VERBOSE=1 timeout -k 10 900 python -u $R/tools/run_search.py --kind sift --n 1000000 --ef 20,50,100 --reps 4 --cache $C --rows u8 > $R/gpurun_out/ab_u8_$mode.log 2>&1 || { echo "$mode failed"; tail -5 $R/gpurun_out/ab_u8_$mode.log; exit 1; }
  echo "== $mode"; grep -E "^ef=|search kernel" $R/gpurun_out/ab_u8_$mode.log | awk '/^ef=/{ef=$0} /search kernel/{t[ef]=$4} END{for (e in t) print e, t[e], "ms (last launch)"}' | sort -t= -k2 -n
done
