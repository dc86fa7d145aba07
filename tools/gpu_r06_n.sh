#!/bin/bash
# round 6, GPU call N: quarters form shipped from 768 residue polynomials per launch - parity of the N = 16384 cases, the activated block at N = 16384, the bench line
OUT=gpurun_out/r06n; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_rlwe_semantics.py tests/test_gpu_bsgs_qp.py tests/test_gpu_cpp_api.py tests/test_gpu_host_threads.py -q -x -p no:cacheprovider -m gpu -k "fold14 or large or 16384 or n16384 or threads or quarters" 2>&1 | tail -4 | tee $OUT/pytest_n16384.txt
for i in 1 2 3; do timeout 300 ./examples/encrypted_gpt2_block_act 8 5 json ladder 14 2>&1 | grep -o '"ms_per_token": [0-9.]*\|"correct": [a-z]*' | tr '\n' ' '; echo; done | tee $OUT/block_act_n16384.txt
timeout 300 ./examples/encrypted_gpt2_linear qkv 5 text 8 14 2>&1 | tail -2 | tee $OUT/linear_n16384.txt
timeout 300 python tools/large_ring_bench.py 2>&1 | grep "N= 16384" | tee $OUT/large_ring.txt
