#!/bin/bash
# plaintext-product kernels, branch-free form: parity (whole-buffer config 3, multi-right-hand-side shapes), configs[2] A/B (mvbuf1 = the single-column kernel still with per-column checks)
mkdir -p gpurun_out/r04n
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "matvec or config3" 2>&1 | grep -E "passed|failed" | tee gpurun_out/r04n/ab_matvec_full3.txt
for i in 1 2; do
  DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_mvbuf1.so python tools/misc_bench.py 2>&1 | grep -E "^matvec_plain" | sed 's/^/mvbuf1 /'
  python tools/misc_bench.py 2>&1 | grep -E "^matvec_plain" | sed 's/^/HEAD   /'
done | tee -a gpurun_out/r04n/ab_matvec_full3.txt
