#!/bin/bash
# round 6, GPU call K: class launches of the hoisted / Galois-inverse kernels on mixed contexts; then the whole GPU suite and the timed default bench
OUT=gpurun_out/r06k; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_limb_classes.py tests/test_gpu_bsgs_qp.py -q -x -p no:cacheprovider -m gpu 2>&1 | tail -6 | tee $OUT/pytest_subset.txt
timeout 2400 python -m pytest tests -q -x -p no:cacheprovider -m gpu 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -4 | tee $OUT/bench_time.txt
cut -c1-300 $OUT/bench.json
