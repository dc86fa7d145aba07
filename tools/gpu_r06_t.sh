#!/bin/bash
# round 6, GPU call T: key products through the twiddle chain (FoldArith::mac_var) in relin / hoisted kernels against the mul60 build (var_old.so); parity first
OUT=gpurun_out/r06t; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bsgs_qp.py tests/test_rlwe_semantics.py -q -p no:cacheprovider -x -m gpu 2>&1 | tail -3 | tee $OUT/pytest_subset.txt
for i in 1 2 3; do
  for v in old HEAD; do
    if [ $v = HEAD ]; then unset DPFHE_AB_LIB; else export DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_$v.so; fi
    timeout 300 python tools/ab_relin.py 2>&1 | grep ABRELIN
    timeout 300 python tools/ab_relin13.py 2>&1 | grep RELIN13
    timeout 300 python tools/ab_packed.py 2>&1 | grep -i "switch_key_qp\|rotate_hoisted_qp" | sed "s/^/PACKED /"
  done
done | tee $OUT/ab_keyproducts.txt
