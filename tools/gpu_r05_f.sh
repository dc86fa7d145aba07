#!/bin/bash
# round 5, GPU call F: the packed layers at N = 16384 through the C++ operator API (a process without PyTorch), after the scratch of the composed
# operations moved from a stream-ordered memory pool to per-stream arenas; the large-ring cases of the Python suite on the same library
OUT=gpurun_out/r05f; mkdir -p $OUT
export TMPDIR=/tmp
for a in "qkv 2 text 1 14" "qkv 2 text 8 14" "all 2 text 8 14" "1024x1024 1 text 4 14" "qkv 2 text 8 13"; do
  timeout 600 ./examples/encrypted_gpt2_linear $a 2>&1 | grep -v amdgpu.ids
done | tee $OUT/packed_linear_n16384.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bsgs_qp.py tests/test_rlwe_semantics.py tests/test_gpu_cpp_api.py -q -p no:cacheprovider -m gpu -k "fold14 or large or sliced or n16384 or shoup14 or fold15" 2>&1 | tail -4 | tee $OUT/pytest_large.txt
