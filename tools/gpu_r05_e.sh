#!/bin/bash
# round 5, GPU call E: the packed pipeline's rotation entries at N = 16384 (composed from the batched transforms, one key per item group): parity against the
# oracle, and one GPT-2 layer at N = 16384 through the C++ operator API, decrypted and compared
OUT=gpurun_out/r05e; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_bsgs_qp.py tests/test_rlwe_semantics.py -q -p no:cacheprovider -m gpu -k "fold14 or n8192" 2>&1 | tail -8 | tee $OUT/pytest_fold14.txt
for a in "qkv 2 text 1 14" "qkv 2 text 8 14" "square 2 text 8 14" "qkv 2 text 8 13"; do
  timeout 600 ./examples/encrypted_gpt2_linear $a 2>&1 | grep -v amdgpu.ids
done | tee $OUT/packed_linear_n16384.txt
