#!/bin/bash
# round 5, GPU call K: N = 8192 forward / inverse transform against the batch size, the library's 512-thread kernels (HEAD) and the halves form (var_halves.so), alternated
OUT=gpurun_out/r05k; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2; do
  timeout 300 python tools/ntt13_batch_sweep.py 2>&1 | grep SWEEP13
  DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_halves.so timeout 300 python tools/ntt13_batch_sweep.py 2>&1 | grep SWEEP13
done | tee $OUT/ntt13_batch_sweep.txt
