#!/bin/bash
# round 6, GPU call M: the N = 16384 transforms in "quarters" form (ntt_quarters.h) against the 1024-thread kernel (var_noquarters.so), alternated; parity first
OUT=gpurun_out/r06m; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_rlwe_semantics.py tests/test_gpu_bsgs_qp.py -q -x -p no:cacheprovider -m gpu -k "fold14 or large_ring or 16384" 2>&1 | tail -4 | tee $OUT/pytest_fold14.txt
for i in 1 2; do
  for v in HEAD noquarters; do
    if [ $v = HEAD ]; then unset DPFHE_AB_LIB; else export DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_$v.so; fi
    timeout 300 python tools/ntt14_batch_sweep.py 2>&1 | grep SWEEP14
  done
done | tee $OUT/ntt14_sweep.txt
