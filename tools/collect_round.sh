#!/bin/bash
# Round evidence: bench line, rocprofv3 kernel stats, PMC passes (SQ utilisation, HBM traffic) of the dominant kernels, the
# register-only ALU ceiling with in-kernel clocks, the shader clock while the kernels run, the packed-layer profile.
# usage (repo root on the GPU box): bash tools/collect_round.sh r03
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo; cat /sys/fs/cgroup/cpu.max) > $OUT/env.log 2>&1
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
./tools/bin/ubench2 > $OUT/ubench2.log 2>&1; echo "ubench2 rc=$?"
./tools/clock_probe > $OUT/clock_probe.txt 2>&1; echo "clock_probe rc=$?"
python tools/power_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/power_probe.txt; echo "power_probe rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?"
f=$(find $OUT/prof -name "*.db" | head -1); [ -n "$f" ] && python tools/prof_summary.py $f $OUT/kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline" > /dev/null
[ -n "$f" ] && python tools/prof_dispatches.py $f $OUT/kernel_dispatches.txt "per launch size: rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline (the 4096-workgroup NTT launches are BASELINE configs[1])" "ntt_|ct_mul|copy_kernel|matvec|relin|hoisted|rescale|reduce" > /dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_pl -o pl -- ./examples/encrypted_gpt2_linear qkv 5 text 8 > $OUT/packed_linear.log 2> $OUT/prof_pl.err; echo "rocprof packed rc=$?"
f=$(find $OUT/prof_pl -name "*.db" | head -1); [ -n "$f" ] && python tools/prof_summary.py $f $OUT/packed_linear_8tokens_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- ./examples/encrypted_gpt2_linear qkv 5 text 8  (setup + 6 applications of 8 tokens)" > /dev/null
# HBM traffic and VALU utilisation of the packed layer's kernels (one counter set per run)
i=0
for set in "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set -d $OUT/pmcp$i -o pmc -- ./examples/encrypted_gpt2_linear qkv 3 text 8 > $OUT/pmcp$i.log 2>&1
  f=$(find $OUT/pmcp$i -name "*.db" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f "hoisted_qp|matvec_fold|relin_kernel|ntt_inv_galois|rescale" > $OUT/pmc_packed_pass$i.txt 2>&1
  echo "pmc packed pass $i rc=$? ($set)"
done
i=0
for set in "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $set -d $OUT/pmc$i -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc$i.log 2>&1
  f=$(find $OUT/pmc$i -name "*.db" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f "ntt_|ct_mul|reduce_" > $OUT/pmc_bench_pass$i.txt 2>&1
  echo "pmc bench pass $i rc=$? ($set)"
done
# the NTT kernels at a steady-state batch (8192 RNS polynomials = 32768 residue polynomials) with dispatch durations, for VALU-busy and in-kernel clock
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace -d $OUT/pmcn$i -o pmc -- python tools/ntt_bench.py 8192 2048 > $OUT/pmcn$i.log 2>&1
  f=$(find $OUT/pmcn$i -name "*.db" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f "ntt_|ct_mul" > $OUT/pmc_ntt_pass$i.txt 2>&1
  [ -n "$f" ] && python tools/prof_summary.py $f $OUT/pmc_ntt_pass${i}_durations.txt "dispatch durations of the same run (rocprofv3 --pmc ... --kernel-trace -- python tools/ntt_bench.py 8192 2048)" > /dev/null 2>&1
  echo "pmc ntt pass $i rc=$? ($set)"
done
find $OUT -name "*.db" -delete   # summaries are kept; raw databases exceed the 64 MiB copy-back limit
rm -rf $OUT/prof $OUT/prof_pl $OUT/pmc[0-9] $OUT/pmcn[0-9] $OUT/pmcp[0-9]
python tools/derive_round.py $OUT $TAG > /dev/null 2>&1; echo "derive rc=$?"
ls $OUT
