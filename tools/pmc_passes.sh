#!/bin/bash
# PMC passes over tools/run_search.py on the GPU box (each pass its own run, counters only).
# usage (from the repo root on the box): bash tools/pmc_passes.sh <out-subdir-of-gpurun_out> [run_search args]
# PMC_PASSES="3 4" runs only those passes (3 and 4 are the memory-side byte counters); PMC_CACHE names the graph cache file.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CACHE=${PMC_CACHE:-/tmp/g1m.npz}
python3 $R/tools/run_search.py --cache $CACHE --reps 1 "$@" > $OUT/prep.log 2>&1 || exit 1
i=0
for pass in \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" \
  "FETCH_SIZE TCC_HIT_sum" \
  "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
  "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN2_sum" \
  "SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_WAVES" ; do
  i=$((i+1))
  if [ -n "${PMC_PASSES:-}" ] && ! echo " $PMC_PASSES " | grep -q " $i "; then continue; fi
  timeout -s KILL ${PMC_TIMEOUT:-120} rocprofv3 --pmc $pass --output-format csv -d $OUT/p$i -- python3 $R/tools/run_search.py --cache $CACHE --reps 3 "$@" > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
