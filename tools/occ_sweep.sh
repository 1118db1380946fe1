# Kernel time against queries resident per CU (limited by padding the dynamic LDS size), at two batch sizes.
# usage: bash tools/occ_sweep.sh   (on the GPU box; prints one line per setting)
timeout -k 10 200 python -u tools/run_search.py --cache /tmp/g1m.npz --reps 1 > /dev/null || exit 1
for nq in 10000 40000; do
for pad in 0 600 1800 2800 4200 5600 9600 16000; do
  echo -n "nq=$nq pad=$pad: "
  HNSW_AMD_LDS_PAD=$pad HNSW_AMD_VERBOSE=1 timeout -k 10 100 python -u tools/run_search.py --cache /tmp/g1m.npz --nq $nq --reps 4 2>&1 | grep -E "rep 3|resident" | tail -2 | tr '\n' ' '; echo
done
done
