#!/bin/bash
# Round-end evidence: bench line, rocprofv3 kernel stats, PMC HBM traffic of the dominant kernels.
# usage (repo root on the GPU box): bash tools/collect_round.sh r01
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?"
f=$(find $OUT/prof -name "*.db" | head -1); [ -n "$f" ] && python tools/prof_summary.py $f $OUT/kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline" | head -8
i=0
for set in "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $set -d $OUT/pmc$i -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc$i.log 2>&1
  f=$(find $OUT/pmc$i -name "*.db" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f "ntt_|ct_mul|reduce_" > $OUT/pmc$i.txt 2>&1
  echo "pmc pass $i rc=$? ($set)"; head -12 $OUT/pmc$i.txt
done
find $OUT -name "*.db" -delete   # summaries are kept; raw databases exceed the 64 MiB copy-back limit
