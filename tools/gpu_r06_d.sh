#!/bin/bash
# round 6, GPU call D: host-thread safety at the boundary + the whole GPU suite on the round-6 library
OUT=gpurun_out/r06d; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_host_threads.py -q -x -p no:cacheprovider -m gpu 2>&1 | tail -15 | tee $OUT/pytest_threads.txt
timeout 2400 python -m pytest tests -q -x -p no:cacheprovider -m gpu 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
