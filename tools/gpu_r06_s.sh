#!/bin/bash
# round 6, GPU call S: the tensor step through the twiddle chain (FoldArith::prod_tw / mul_ptw_add) against the four-mul60 build (var_old.so), alternated on one box; parity first
OUT=gpurun_out/r06s; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_limb_classes.py -q -p no:cacheprovider -x -k "ct_mul or multiply or class" 2>&1 | tail -3 | tee $OUT/pytest_subset.txt
for i in 1 2 3; do
  for v in old HEAD; do
    if [ $v = HEAD ]; then unset DPFHE_AB_LIB; else export DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_$v.so; fi
    timeout 300 python tools/ab_bench.py 2>&1 | grep -i "ct_mul\|ntt" | sed "s/^/$v /"
  done
done | tee $OUT/ab_tensor.txt
