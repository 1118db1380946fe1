#!/bin/bash
# A/B of the row-shape kernels against the kernel for any row length (HNSW_AMD_GENERIC_ROWS=1) on one index.
# usage: bash tools/ab_rows.sh <kind> <n> <ef list> [extra run_search.py arguments]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
kind=$1; n=$2; efs=$3; shift 3
C=/tmp/ab_${kind}_${n}.npz
O=$R/gpurun_out/ab_${kind}_${n}
mkdir -p $R/gpurun_out
for mode in any shaped any shaped; do
  unset HNSW_AMD_GENERIC_ROWS
  if [ $mode = any ]; then export HNSW_AMD_GENERIC_ROWS=1; fi
  HNSW_AMD_VERBOSE=1 timeout -k 10 900 python -u $R/tools/run_search.py --kind $kind --n $n --ef $efs --reps 4 --cache $C "$@" > ${O}_$mode.log 2>&1 || { echo "$mode failed"; tail -5 ${O}_$mode.log; exit 1; }
  echo "== $mode"; grep -E "^ef=|search kernel" ${O}_$mode.log | awk '/^ef=/{ef=$0} /search kernel/{t[ef]=$4} END{for (e in t) print e, t[e], "ms (last launch)"}' | sort -t= -k2 -n
done
