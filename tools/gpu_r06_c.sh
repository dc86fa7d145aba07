#!/bin/bash
# round 6, GPU call C: one-launch class transforms (ntt_classes_kernel) - parity, class bench, then the full bench line
OUT=gpurun_out/r06c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_limb_classes.py tests/test_gpu_random_params.py -q -x -p no:cacheprovider -m gpu 2>&1 | tail -5 | tee $OUT/pytest_classes.txt
timeout 600 python tools/class_bench.py 2048 12 2>&1 | grep -v CLASS_BENCH | tee $OUT/class_bench_n4096.txt
timeout 900 python bench.py 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-3000
tail -5 $OUT/bench.err
