#!/bin/bash
# round 6, GPU call U: the shipped combination (tensor + hoisted kernels on the twiddle chain, relin kernels on mul60): whole GPU suite, then the relin / packed A/B once more
OUT=gpurun_out/r06u; mkdir -p $OUT; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -q -p no:cacheprovider -m gpu 2>&1 | tail -4 | tee $OUT/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
for i in 1 2; do
  for v in old HEAD; do
    if [ $v = HEAD ]; then unset DPFHE_AB_LIB; else export DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_$v.so; fi
    timeout 300 python tools/ab_relin.py 2>&1 | grep ABRELIN
    timeout 300 python tools/ab_packed.py 2>&1 | grep -i "switch_key_qp" | sed "s/^/PACKED /"
  done
done | tee $OUT/ab_final.txt
