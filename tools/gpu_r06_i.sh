#!/bin/bash
# round 6, GPU call I: the round's evidence set (tools/collect_round.sh r06: bench line, probes, rocprofv3 kernel stats, packed-layer stats, PMC passes)
# + kernel stats and HBM counters of the class kernels (tools/class_bench.py)
export TMPDIR=/tmp
bash tools/collect_round.sh r06 bench probes prof packed pmc 2>&1 | tail -30
OUT=gpurun_out/r06; db() { find $1 -name "*.db" | head -1; }
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_cls -o cls -- python tools/class_bench.py 2048 12 > $OUT/class_bench_prof.log 2>&1
f=$(db $OUT/prof_cls); [ -n "$f" ] && python tools/prof_summary.py $f $OUT/class_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python tools/class_bench.py 2048 12  (per class: 1024 RNS polys forward / inverse, 2048-pair fused multiply, N = 4096, L = 4)" > /dev/null
rm -rf $OUT/prof_cls
i=0
for set in "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set -d $OUT/pmcc$i -o pmc -- python tools/class_bench.py 2048 12 > $OUT/pmcc$i.log 2>&1
  f=$(db $OUT/pmcc$i); [ -n "$f" ] && python tools/pmc_summary.py $f "F64|FoldScaled|classes_kernel" > $OUT/pmc_class_pass$i.txt 2>&1
  echo "pmc class pass $i rc=$?"; rm -rf $OUT/pmcc$i
done
find $OUT -name "*.db" -delete; ls $OUT
