#!/bin/bash
# round 4, GPU call A: which box is this (regime fingerprint), the forms of the fused multiply side by side, the workgroup timeline,
# the changed tests, one driver-style bench line, and memory-latency counters of the multiply.
TAG=${1:-r04a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
db() { find $1 -name "*.db" | head -1; }
(rocm-smi --showcomputepartition --showmemorypartition --showclocks --showmaxpower --showmeminfo vram 2>&1 | grep -v "^$" | head -40; rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -8; nproc; cat /sys/fs/cgroup/cpu.max) > $OUT/env.txt 2>&1
(rocprofv3 --list-avail 2>&1 || rocprofv3 -L 2>&1) | grep -E "^\s*(Name|name)|TCP_|TCC_|UTCL|TCA_|TA_" | head -400 > $OUT/counters_avail.txt
timeout 300 python tools/ab_forms.py 8192 2>&1 | grep -v amdgpu.ids > $OUT/ab_forms.txt; echo "ab_forms rc=$?"; cat $OUT/ab_forms.txt
timeout 200 python tools/ctmul_trace.py 2048 2>&1 | grep -v amdgpu.ids > $OUT/ctmul_trace_2048.txt; echo "trace rc=$?"; cat $OUT/ctmul_trace_2048.txt
timeout 200 python tools/ctmul_trace.py 8192 2>&1 | grep -v amdgpu.ids > $OUT/ctmul_trace_8192.txt; cat $OUT/ctmul_trace_8192.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_form or full_size or large_batch or bench_py or per_gpu_shard" 2>&1 | tail -8 | tee $OUT/pytest_subset.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<P
import json
try:
    d = json.loads([l for l in open("$OUT/bench.json") if l.startswith("{")][-1])
    print("LINE BYTES", len(json.dumps(d)))
    print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "compute_only_ct_mul_per_s", "reduce_consistent", "bit_exact_sample")}))
    print(json.dumps(d["config"]["autotune"])); print(json.dumps(d["roofline"]["ntt"])); print(json.dumps(d["roofline"]["regime"])); print(json.dumps(d.get("other_configs"))[:1500])
    print("frac", d["roofline"]["frac"], "frac_alu", d["roofline"]["frac_alu"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
except Exception as e:
    print("bench parse failed", e); print(open("$OUT/bench.err").read()[-1500:])
P
cp gpurun_out/bench_detail.json $OUT/bench_detail.json 2>/dev/null
# memory-latency / TLB counters of the multiply (one set per run; names that this rocprofv3 does not know fail that pass only)
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc$i -o pmc -- python tools/ntt_bench.py 1024 8192 > $OUT/pmc$i.log 2>&1
  rc=$?
  f=$(db $OUT/pmc$i)
  [ -n "$f" ] && python tools/pmc_summary.py $f "ct_mul|ntt_fwd|ntt_inv" > $OUT/pmc_lat_pass$i.txt 2>&1
  [ -n "$f" ] && python tools/prof_summary.py $f $OUT/pmc_lat_pass${i}_durations.txt "dispatch durations of the same run ($set)" > /dev/null 2>&1
  echo "pmc pass $i rc=$rc ($set)"; [ -f $OUT/pmc_lat_pass$i.txt ] && grep -E "ct_mul" $OUT/pmc_lat_pass$i.txt | head -6; [ $rc -ne 0 ] && tail -3 $OUT/pmc$i.log
  rm -rf $OUT/pmc$i
done
find $OUT -name "*.db" -delete
du -sh $OUT; ls $OUT
