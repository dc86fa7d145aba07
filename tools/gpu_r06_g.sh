#!/bin/bash
# round 6, GPU call G: the f64_wide class (primes of 47 ... 50 bits) - parity, class bench
OUT=gpurun_out/r06g; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_limb_classes.py tests/test_gpu_random_params.py -q -x -p no:cacheprovider -m gpu 2>&1 | tail -8 | tee $OUT/pytest_classes.txt
timeout 600 python tools/class_bench.py 2048 12 2>&1 | grep -v CLASS_BENCH | tee $OUT/class_bench_n4096.txt
timeout 600 python tools/class_bench.py 2048 13 2>&1 | grep -v CLASS_BENCH | tee $OUT/class_bench_n8192.txt
