#!/bin/bash
# round 6, GPU call H: key switching on the class of a uniform context (with_policy) - parity of the class suite + rlwe semantics, relinearize throughput per class
OUT=gpurun_out/r06h; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_limb_classes.py tests/test_gpu_random_params.py tests/test_rlwe_semantics.py -q -x -p no:cacheprovider -m gpu 2>&1 | tail -8 | tee $OUT/pytest.txt
timeout 600 python tools/class_relin_bench.py 2>&1 | tee $OUT/class_relin.txt
