#!/bin/bash
OUT=gpurun_out/icache; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -iE "ICACHE|IFETCH|INST_LEVEL|SQ_INSTS_SMEM|SQC_" | head -40 > $OUT/list.txt
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set -d $OUT/p$i -o pmc -- python tools/ntt_bench.py 1024 2048 > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*.db" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f "ct_mul_kernel<FoldArith, 12|ntt_fwd_kernel<FoldArith, 12" > $OUT/p$i.txt 2>&1
  echo "pass $i rc=$? ($set)"; cat $OUT/p$i.txt | head -30; tail -3 $OUT/p$i.log | grep -iE "error|invalid|not" | head -3
done
find $OUT -name "*.db" -delete
