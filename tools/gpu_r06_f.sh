#!/bin/bash
# round 6, GPU call F: after the pruning - whole GPU suite, the (15, 1) shape of the giant steps in the key-major order (time + FETCH_SIZE), full bench line
OUT=gpurun_out/r06f; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -x -p no:cacheprovider -m gpu 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
timeout 300 python tools/giant_traffic.py 2>&1 | grep GIANT | tee $OUT/giant_timing.txt
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE -d $OUT/p1 -o pmc -- python tools/giant_traffic.py > $OUT/p1.log 2>&1
f=$(find $OUT/p1 -name "*.db" | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f "relin_kernel" | grep -v GRBM | tee $OUT/giant_fetch.txt
find $OUT -name "*.db" -delete; rm -rf $OUT/p1
timeout 900 python bench.py 2>$OUT/bench.err > $OUT/bench.json; tail -3 $OUT/bench.err; cut -c1-600 $OUT/bench.json
