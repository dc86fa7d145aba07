#!/bin/bash
# round 4, late: the multi-block stack on a modulus chain + the deep-level base-extension tests
mkdir -p gpurun_out/r04h
timeout 900 python -m pytest tests/test_gpu_exact_multiply.py -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -8
timeout 900 ./examples/encrypted_gpt2_stack 2 1 text 10 > gpurun_out/r04h/stack_l10_t2.txt 2>&1; echo "stack l10 rc=$?"
tail -12 gpurun_out/r04h/stack_l10_t2.txt | cut -c1-700
timeout 900 ./examples/encrypted_gpt2_stack 8 2 json 7 > gpurun_out/r04h/stack_l7_t8.txt 2>&1; echo "stack l7 t8 rc=$?"
tail -3 gpurun_out/r04h/stack_l7_t8.txt | cut -c1-1200
timeout 900 ./examples/encrypted_gpt2_stack 8 2 json 10 > gpurun_out/r04h/stack_l10_t8.txt 2>&1; echo "stack l10 t8 rc=$?"
tail -3 gpurun_out/r04h/stack_l10_t8.txt | cut -c1-1200
