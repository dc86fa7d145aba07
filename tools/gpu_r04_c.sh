#!/bin/bash
# round 4, GPU call C: quadpf (prefetch distance sweep), Dot30 baby steps, exact-division inverse NTT; the whole GPU suite once
TAG=${1:-r04c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for d in 48 16 32 96 192; do
  DPFHE_CTMUL_PF_DIST=$d timeout 200 python tools/ab_forms.py 8192 n4096 2>&1 | grep -E "quadpf|quad " | sed "s/^/[pf_dist=$d] /" >> $OUT/ab_forms.txt
done
cat $OUT/ab_forms.txt
timeout 200 python tools/ab_forms.py 8192 2>&1 | grep -v amdgpu.ids > $OUT/ab_forms_full.txt; grep -E "N=8192" -A20 $OUT/ab_forms_full.txt | grep -E "quad|dual" | head -12
timeout 200 python tools/ab_packed.py 8 64 2>&1 | grep -v amdgpu.ids > $OUT/ab_packed.txt; cat $OUT/ab_packed.txt
timeout 200 python tools/ab_packed.py 1 64 2>&1 | grep -E "rotate_hoisted|switch_key" | sed "s/^/[1 token] /" >> $OUT/ab_packed.txt
timeout 300 python bench.py --skip-other --no-cpu-baseline > $OUT/bench_lean.json 2> $OUT/bench_lean.err; python - <<P
import json
try:
    d = json.loads([l for l in open("$OUT/bench_lean.json") if l.startswith("{")][-1])
    print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "compute_only_ct_mul_per_s", "reduce_consistent")}))
    print(json.dumps(d["config"]["autotune"])); nv = d["roofline"]["ntt"]; print({k: nv[k] for k in ("fwd_frac", "inv_frac", "fwd_us", "inv_us", "out_of_place_fwd_frac", "out_of_place_inv_frac")}, nv.get("sustained_2s"))
except Exception as e:
    print("bench parse failed", e); print(open("$OUT/bench_lean.err").read()[-1500:])
P
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
