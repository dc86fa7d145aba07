#!/bin/bash
# round 5, GPU call L: the halves form of the N = 8192 transforms in the library for large batches: parity on both sides of the threshold, the batch sweep of the
# shipped library against the 512-thread-only build (var_base.so: -DDPFHE_N13_HALVES=0), a bench line
OUT=gpurun_out/r05l; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "n8192 or halves or ntt" 2>&1 | tail -3 | tee $OUT/pytest_subset.txt
for i in 1 2; do
  DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_base.so timeout 300 python tools/ntt13_batch_sweep.py 256 384 512 1024 2048 2>&1 | grep SWEEP13
  timeout 300 python tools/ntt13_batch_sweep.py 256 384 512 1024 2048 2>&1 | grep SWEEP13
done | tee $OUT/ntt13_batch_sweep_shipped.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'P'
import json
d = json.loads([l for l in open("gpurun_out/r05l/bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "bit_exact_sample", "reduce_consistent")})
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["roofline"].items() if k.startswith("n8192") or k in ("frac", "ntt_fwd_frac", "ntt_inv_frac")})
print(d["other_configs"]["n8192_l6"], len(json.dumps(d)))
P
