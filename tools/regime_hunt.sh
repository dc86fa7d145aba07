#!/bin/bash
# One fingerprint of this box (the multiply alone, the workgroup timeline); if it is a SLOW-regime box (quad form above 3.25 ms per 8192 pairs,
# MEASUREMENTS.md section 5) collect what round 3 lacked there: every form side by side, the timeline, memory-latency / TLB counters, partition modes.
OUT=gpurun_out/hunt_$(date +%H%M%S); mkdir -p $OUT; export TMPDIR=/tmp
db() { find $1 -name "*.db" | head -1; }
timeout 200 python tools/ctmul_trace.py 8192 json 2>/dev/null | grep "^{" > $OUT/trace.json
python - <<P
import json
r = json.load(open("$OUT/trace.json"))
s = r["segments_us"]
print("FINGERPRINT plain %.1f us  lifetime %.2f  wait_first %.2f (%.1f %%)  forward %.2f  inverse %.2f  R %.3f" % (r["kernel_us_plain"], s["lifetime"]["median"], s["wait_first_load"]["median"],
      100 * r["share_of_lifetime"]["wait_first_load"], s["forward"]["median"], s["inverse"]["median"], r["start_phase_concentration_R"]))
open("$OUT/slow", "w").write("1" if r["kernel_us_plain"] > 3250 else "0")
P
if [ "$(cat $OUT/slow)" = "1" ]; then
  echo "SLOW BOX: collecting"
  rocm-smi --showcomputepartition --showmemorypartition --showclocks --showmaxpower --showmeminfo vram > $OUT/env.txt 2>&1
  timeout 200 python tools/ctmul_trace.py 8192 2>&1 | grep -v amdgpu.ids | tee $OUT/trace.txt
  timeout 300 python tools/ab_forms.py 8192 n4096 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_forms.txt
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
             "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
             "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc$i -o pmc -- python tools/ntt_bench.py 1024 8192 > $OUT/pmc$i.log 2>&1
    f=$(db $OUT/pmc$i); [ -n "$f" ] && python tools/pmc_summary.py $f "ct_mul" > $OUT/pmc_lat_pass$i.txt 2>&1; grep -A9 "ct_mul_quad" $OUT/pmc_lat_pass$i.txt | head -10; rm -rf $OUT/pmc$i
  done
  timeout 300 python bench.py --skip-other --no-cpu-baseline > $OUT/bench_lean.json 2> $OUT/bench.err; tail -c 400 $OUT/bench_lean.json
fi
