#!/bin/bash
# round 5, GPU call C: key-switch kernels without the digit canonicalisation (HEAD) against the previous commit's library (var_prev.so), alternated;
# parity of every key-switch path; the generic-prime (Shoup) arm next to the fold arm at the headline shape.
OUT=gpurun_out/r05c; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bsgs_qp.py tests/test_rlwe_semantics.py tests/test_gpu_random_params.py tests/test_gpu_cpp_api.py -x -q -p no:cacheprovider -m gpu 2>&1 | tail -4 | tee $OUT/pytest_subset.txt
for i in 1 2 3; do
  for v in prev HEAD; do
    if [ $v = HEAD ]; then unset DPFHE_AB_LIB; else export DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_$v.so; fi
    timeout 300 python tools/ab_bench.py 2>&1 | grep -E "relinearize|keyswitch"
    timeout 300 python tools/ab_relin13.py 2>&1 | grep RELIN13
  done
done | tee $OUT/ab_relin_nocanon.txt
unset DPFHE_AB_LIB
timeout 300 python tools/shoup_bench.py 2048 2>&1 | grep -v amdgpu.ids | tee $OUT/shoup_bench.txt
