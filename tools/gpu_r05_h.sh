#!/bin/bash
# round 5, GPU call H: the whole GPU suite, smoke and a driver-style bench line on the round's final code (timed)
OUT=gpurun_out/r05h; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1; grep -E "passed|failed" $OUT/pytest_gpu.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
T0=$(date +%s.%N); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench wall $(echo "$(date +%s.%N) - $T0" | bc) s" | tee $OUT/bench_wall.txt
python - <<'P'
import json
d = json.loads([l for l in open("gpurun_out/r05h/bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "bit_exact_sample", "reduce_consistent")})
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["roofline"].items() if not isinstance(v, (dict, list)) and k != "traffic_source"})
print({k: v for k, v in d["other_configs"]["packed_linear"].items() if k in ("all_correct", "ms_per_token")}, d["other_configs"]["packed_linear"].get("activated_block_n16384", {}).get("correct"), len(json.dumps(d)))
P
