#!/bin/bash
# round 6, GPU call W: the sum of a shard's products taken in the NTT domain (one inverse transform of the total): parity, then the bench line with the new entry
OUT=gpurun_out/r06w; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x -k "sum_in_ntt_domain or bench or sharded or ntt or mul" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" | tail -4 | tee $OUT/pytest_subset.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
b=json.load(open("gpurun_out/r06w/bench.json"))
print(b["value"], b["ms_per_step"], b["other_configs"].get("sum_in_ntt_domain"))
PY
