# Kernel time against queries resident per CU (limited by padding the dynamic LDS size) at the headline batch size.
# usage: bash tools/occ_sweep.sh   (on the GPU box; prints one line per setting: residency, then the kernel times)
timeout -k 10 200 python -u tools/run_search.py --cache /tmp/g1m.npz --reps 1 > /dev/null || exit 1
for nq in 10000; do
for pad in 0 600 1400 2300 3400 4600 6100; do
  echo -n "nq=$nq pad=$pad: "
  HNSW_AMD_LDS_PAD=$pad HNSW_AMD_VERBOSE=1 timeout -k 10 100 python -u tools/run_search.py --cache /tmp/g1m.npz --nq $nq --reps 6 2>&1 | grep -E "search kernel|resident" | sed -e 's/.*resident queries per CU/occ/' -e 's/.*search kernel: \([0-9.]*\) ms.*/\1/' | tail -7 | tr '\n' ' '; echo
done
done
