set -e
for ef in 50 100 200; do
  echo "=== ef=$ef"
  HNSW_AMD_LIB=hnsw_amd/libhnsw_amd_diag.so timeout -k 10 200 python -u tools/diag_phases.py --ef $ef
done
