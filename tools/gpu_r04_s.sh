#!/bin/bash
# plaintext-product + baby-step kernels with the fold that skips the reduction of S0 (f0new) against the committed form (f0old), alternated; parity
mkdir -p gpurun_out/r04s
for i in 1 2 3; do
  for v in f0old f0new; do
    DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_$v.so python tools/ab_packed.py 8 64 2>&1 | grep -E "rotate_hoisted|matvec_plain_multi"
  done
done | tee gpurun_out/r04s/ab_fold0.txt
timeout 900 python -m pytest tests/test_gpu_bsgs_qp.py tests/test_gpu_parity.py -x -q -k "qp or matvec or config3" 2>&1 | grep -E "passed|failed" | tee -a gpurun_out/r04s/ab_fold0.txt
