#!/bin/bash
# round 6, GPU call Q: two tokens per ciphertext - the C++ API tests, per-layer figures, 16-token block runs
OUT=gpurun_out/r06q; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_cpp_api.py -q -x -p no:cacheprovider -m gpu 2>&1 | tail -4 | tee $OUT/pytest_cpp.txt
for ln in 13 14; do for tpc in 1 2; do
  T=$((8 * tpc))
  timeout 400 ./examples/encrypted_gpt2_linear all 5 text $T $ln $tpc 2>&1 | grep -v "^OK" | sed "s/^/log2n $ln tpc $tpc: /"
  timeout 400 ./examples/encrypted_gpt2_block_act $T 5 json ladder $ln $tpc 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('log2n', d['log2_n'], 'block: tokens_per_ct', d['tokens_per_ciphertext'], 'tokens', d['tokens'], 'ms_per_token', d['ms_per_token'], 'correct', d['correct'], 'budget', d['budget_bits'])"
done; done | tee $OUT/two_tokens.txt
