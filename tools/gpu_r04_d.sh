#!/bin/bash
# round 4, GPU call D: exact multiply + activated FFN (correctness first), finer workgroup trace, baby-step pairs-per-thread A/B
TAG=${1:-r04d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_exact_multiply.py -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest_exact.txt
timeout 300 ./examples/encrypted_gpt2_ffn_act 2 1 text 2>&1 | tail -6 | tee $OUT/ffn_act_2.txt
timeout 300 ./examples/encrypted_gpt2_ffn_act 8 2 json 2>&1 | tail -3 | tee $OUT/ffn_act_8.txt
timeout 200 python tools/ctmul_trace.py 8192 2>&1 | grep -v amdgpu.ids | tee $OUT/ctmul_trace_8192.txt
for pp in 2 1 4; do DPFHE_QP_PAIRS=$pp timeout 200 python tools/ab_packed.py 8 64 2>&1 | grep -E "rotate_hoisted" | sed "s/^/[pairs per thread $pp] /" | tee -a $OUT/ab_packed_pp.txt; done
for pp in 2 1 4; do DPFHE_QP_PAIRS=$pp timeout 200 python tools/ab_packed.py 1 64 2>&1 | grep -E "rotate_hoisted" | sed "s/^/[1 token, pairs per thread $pp] /" | tee -a $OUT/ab_packed_pp.txt; done
timeout 600 python -m pytest tests/test_gpu_cpp_api.py -x -q -m gpu -k "activated or ffn_block" 2>&1 | tail -4 | tee $OUT/pytest_cpp.txt
