#!/bin/bash
# round 6, GPU call P: two tokens per ciphertext (the slot rows carry two tokens) in the activated block, N = 8192 and N = 16384, against one token per ciphertext
OUT=gpurun_out/r06p; mkdir -p $OUT; export TMPDIR=/tmp
for ln in 13 14; do for tpc in 1 2; do
  for i in 1 2; do timeout 400 ./examples/encrypted_gpt2_block_act 8 5 json ladder $ln $tpc 2>&1 | tail -2 | head -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('log2n', d['log2_n'], 'tokens_per_ct', d['tokens_per_ciphertext'], 'tokens', d['tokens'], 'ms_per_token', d['ms_per_token'], 'correct', d['correct'], 'budget', d['budget_bits'])"; done
done; done | tee $OUT/block_act_two_tokens.txt
timeout 400 ./examples/encrypted_gpt2_block_act 16 5 json ladder 14 2 2>&1 | tail -2 | cut -c1-400 | tee -a $OUT/block_act_two_tokens.txt
