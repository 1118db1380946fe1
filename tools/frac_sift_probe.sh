set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=/tmp/ab_sift_1000000.npz
for f in 0.45 0.3 0.45 0.3; do
  HNSW_AMD_BULK_FRAC=$f HNSW_AMD_VERBOSE=1 timeout -k 10 900 python -u $R/tools/run_search.py --kind sift --n 1000000 --ef 100,140,200 --reps 5 --cache $C > $R/gpurun_out/frac_sift_$f.log 2>&1 || { echo failed; tail -3 $R/gpurun_out/frac_sift_$f.log; exit 1; }
  echo "== frac $f"; grep -E "^ef=|search kernel|search launch" $R/gpurun_out/frac_sift_$f.log | awk '/^ef=/{ef=$0} /search kernel/{t[ef]=$4} /search launch/{l[ef]=$(NF-0)} END{for (e in t) print e, t[e], "ms", "per CU", l[e]}' | sort -t= -k2 -n
done
