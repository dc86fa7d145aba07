#!/bin/bash
# round 6, GPU call R: the three-block stack with two tokens per ciphertext
OUT=gpurun_out/r06r; mkdir -p $OUT; export TMPDIR=/tmp
for tpc in 1 2; do T=$((8 * tpc)); timeout 900 ./examples/encrypted_gpt2_stack $T 1 json 10 0 $tpc 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stack tokens', d.get('tokens'), 'tpc', $tpc, 'blocks', d.get('blocks'), 'correct_blocks', d.get('correct_blocks'), 'ms_per_token', d.get('ms_per_token'), 'per block', d.get('ms_per_token_per_block'), 'correct', d.get('correct'))"; done | tee $OUT/stack_two_tokens.txt
