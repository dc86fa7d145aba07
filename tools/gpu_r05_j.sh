#!/bin/bash
# round 5, GPU call J: workgroup timelines of the batched forward transform - N = 4096 (256 threads), N = 8192 (512 threads: the library's kernel) and
# N = 8192 in halves form (256 threads) - from diagnostic builds (tools/ntt_trace.py); each arm three times, alternated
OUT=gpurun_out/r05j; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2 3; do
  DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_ntttrace.so timeout 200 python tools/ntt_trace.py n4096 2>&1 | grep NTTTRACE
  DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_ntttrace.so timeout 200 python tools/ntt_trace.py n8192 2>&1 | grep NTTTRACE
  DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_ntttraceh.so timeout 200 python tools/ntt_trace.py n8192 2>&1 | grep NTTTRACE
done | tee $OUT/ntt_workgroup_timelines.txt
