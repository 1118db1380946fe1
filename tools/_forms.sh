cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -u tools/run_search.py --cache /tmp/g1m.npz --reps 1 > /dev/null || exit 1
for mode in auto; do
  [ $mode = v32 ] && export HNSW_AMD_VISITED32=1
  echo "== $mode"
  HNSW_AMD_VERBOSE=1 timeout -k 10 200 python -u tools/run_search.py --cache /tmp/g1m.npz --ef 64,80,100,128,200 --reps 4 2>&1 | grep -E "^ef=|search launch|search kernel" | awk '/^ef=/{print} /launch/{l=$0} /kernel:/{k=$0; n++} /kernel:/ && n%4==0 {print l; print k}' | cut -c1-170
done
