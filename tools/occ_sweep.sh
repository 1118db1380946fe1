for pad in 0 2400 4600 7500 11500 17000 25000 38000 66000; do
  HNSW_AMD_LDS_PAD=$pad HNSW_AMD_VERBOSE=1 timeout -k 10 100 python -u tools/run_search.py --cache /tmp/g1m.npz --reps 4 2>&1 | grep -E "rep 3|resident" | tail -2 | tr '\n' ' '; echo
done
