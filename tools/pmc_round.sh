#!/bin/bash
# PMC counters for the NTT / ct_mul kernels (separate passes; no trace domains combined with --pmc)
TAG=${1:-pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set -d $OUT/p$i -o pmc -- python tools/ntt_bench.py 1024 2048 > $OUT/p$i.log 2>&1
  echo "pass $i rc=$? : $set"
  f=$(find $OUT/p$i -name "*.db" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f "ntt_|ct_mul" > $OUT/p$i.txt 2>&1 && cat $OUT/p$i.txt | head -60
done
find $OUT -name "*.db" -delete
