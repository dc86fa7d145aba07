#!/bin/bash
# round 5, GPU call M (the kernel it measured was removed again - see profiles/r05_ntt13_persistent_ab.txt): persistent prefetching workgroups for the N = 8192 forward transform against the plain 512-thread kernel
# 512-thread kernel (var_multi0.so: -DDPFHE_NTT13_MULTI=0) and with early twiddle requests (var_multie.so), alternated; parity
OUT=gpurun_out/r05m; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bsgs_qp.py -q -p no:cacheprovider -k "n8192 or halves or ntt_forward" 2>&1 | tail -3 | tee $OUT/pytest_subset.txt
for i in 1 2 3; do
  for v in multi0 HEAD multie; do
    if [ $v = HEAD ]; then unset DPFHE_AB_LIB; else export DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_$v.so; fi
    timeout 300 python tools/ntt13_batch_sweep.py 96 128 171 256 320 383 2>&1 | grep SWEEP13
  done
done | tee $OUT/ntt13_multi_sweep.txt
