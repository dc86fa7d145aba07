#!/bin/bash
# round 4, GPU call G: the straight-line two-pair form (quad2) against the others, correctness of every form, one bench line
OUT=gpurun_out/r04g; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_form or all_domains" 2>&1 | tail -4 | tee $OUT/pytest.txt
timeout 300 python tools/ab_forms.py 8192 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_forms.txt
timeout 300 python bench.py --skip-other --no-cpu-baseline > $OUT/bench_lean.json 2> $OUT/bench_lean.err; python - <<P
import json
d = json.loads([l for l in open("$OUT/bench_lean.json") if l.startswith("{")][-1])
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "compute_only_ct_mul_per_s", "reduce_consistent")})); print(json.dumps(d["config"]["autotune"])); print(d["roofline"]["kernel"], d["roofline"]["frac"])
P
