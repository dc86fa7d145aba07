#!/bin/bash
# round 6, GPU call Y: relin_kernel with lazily summed twiddle-chain products (var_ptwlazy.so: acc += mul_ptw(x, prod_tw(e))) against mul60 (HEAD)
OUT=gpurun_out/r06y; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2 3; do
  for v in HEAD ptwlazy; do
    if [ $v = HEAD ]; then unset DPFHE_AB_LIB; else export DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_$v.so; fi
    timeout 300 python tools/ab_relin.py 2>&1 | grep ABRELIN
    timeout 300 python tools/ab_packed.py 2>&1 | grep -i "switch_key_qp" | sed "s/^/PACKED /"
  done
done | tee $OUT/ab_ptwlazy.txt
