#!/bin/bash
# round 5, GPU call A: N = 8192 batched transforms in "halves" form (256 threads, column stage + two 4096-point sub-transforms through one
# LDS buffer) against the 512-thread kernels, same box, alternated; parity of everything that launches them; tuner-free context creation.
# (arms as built at that commit: HEAD = the halves kernels ON (today: bash tools/ab_variant.sh halves -DDPFHE_N13_HALVES=1), var_base.so = the 512-thread kernels = what the
# library ships, var_h4.so = -DDPFHE_HALVES_OCC=4, var_hne.so = -DDPFHE_HALVES_INV_EARLY=0)
OUT=gpurun_out/r05a; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider -k "ntt or every_form or config2 or sliced or errors or policy or identities" 2>&1 | tail -5 | tee $OUT/pytest_subset.txt
for i in 1 2 3; do
  for v in base HEAD h4 hne; do
    if [ $v = HEAD ]; then unset DPFHE_AB_LIB; else export DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_$v.so; fi
    timeout 300 python tools/ab_bench.py 2>&1 | grep -E "n8192 (ntt|ct_mul)"
  done
done | tee $OUT/ab_halves.txt
unset DPFHE_AB_LIB
# sustained: 2 s of back-to-back launches per arm, with board power and clock (tools/ab_sustained.py)
for v in base HEAD; do
  if [ $v = HEAD ]; then unset DPFHE_AB_LIB; else export DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_$v.so; fi
  timeout 300 python tools/ab_sustained.py 1.5 2>&1 | grep n8192
done | tee $OUT/ab_halves_sustained.txt
