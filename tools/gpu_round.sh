#!/bin/bash
# One GPU-box session: parity tests, micro-benchmarks, bench line, rocprofv3 kernel stats.
# Usage (from the repo root on the box):  bash tools/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== rocminfo ==" > $OUT/env.log
(rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -12; rocm-smi --showmeminfo vram | head -8; nproc; grep -m1 "model name" /proc/cpuinfo) >> $OUT/env.log 2>&1
echo "== pytest -m gpu ==" 
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -15 $OUT/pytest.log
echo "== smoke =="
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
echo "== ubench =="
timeout 600 ./tools/ubench > $OUT/ubench.log 2>&1; echo "ubench rc=$?" | tee -a $OUT/ubench.log
cat $OUT/ubench.log
echo "== bench =="
timeout 900 python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -5 $OUT/bench.err
echo "== rocprofv3 kernel trace =="
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?"
ls -R $OUT/prof | head -20
find $OUT/prof -name "*kernel_stats*" | head -1 | xargs -r head -20
# keep the merged-back payload small: drop the raw trace, keep stats
find $OUT -name "*.db" -delete
