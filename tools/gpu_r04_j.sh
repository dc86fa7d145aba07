#!/bin/bash
# A/B: giant-step key inner products at N = 8192 with 8 words per thread (DPFHE_RELIN13_LOGE3=1) against the default 16
mkdir -p gpurun_out/r04j
for i in 1 2; do
  python tools/ab_relin13.py 2>&1 | grep RELIN13
  DPFHE_RELIN13_LOGE3=1 python tools/ab_relin13.py 2>&1 | grep RELIN13
done | tee gpurun_out/r04j/ab_relin13.txt
DPFHE_RELIN13_LOGE3=1 timeout 600 python -m pytest tests/test_gpu_bsgs_qp.py -x -q -k "n8192" 2>&1 | grep -E "passed|failed" | tee -a gpurun_out/r04j/ab_relin13.txt
