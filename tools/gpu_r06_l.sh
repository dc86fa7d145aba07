#!/bin/bash
# round 6, GPU call L: does the shard-local reduce co-reside with the fused multiply once it fits the 32 registers the multiply leaves per SIMD lane?
# HEAD (reduce_partial_kernel at 30 VGPRs) against var_reduce4 (34 VGPRs, 4 loads in flight: round 5's kernel), alternated; overlap / serial / mulonly
OUT=gpurun_out/r06l; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2 3; do
  for v in HEAD reduce4; do
    if [ $v = HEAD ]; then unset DPFHE_AB_LIB; else export DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_$v.so; fi
    for mode in overlap serial mulonly; do TAG=$v timeout 300 python tools/step_bench.py 12 $mode 2>&1 | grep "ms/step"; done
  done
done | tee $OUT/step_ab.txt
