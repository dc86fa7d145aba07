#!/bin/bash
# round 5, GPU call D: the whole GPU suite + smoke + one driver-style bench line on the round's code
OUT=gpurun_out/r05d; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'P'
import json
d = json.loads([l for l in open("gpurun_out/r05d/bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "bit_exact_sample", "reduce_consistent")})
print({k: v for k, v in d["roofline"].items() if not isinstance(v, (dict, list))})
print(json.dumps(d.get("other_configs"))[:1800])
print(len(json.dumps(d)))
P
cp gpurun_out/bench_detail.json $OUT/bench_detail.json 2>/dev/null
