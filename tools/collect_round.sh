#!/bin/bash
# Round evidence, in small bounded steps (every rocprofv3 run serialises the launches: the profiled commands skip bench.py's
# sustained windows and secondary blocks; raw databases are deleted as soon as they are summarised - gpurun copies back <= 64 MiB).
# usage (repo root on the GPU box): bash tools/collect_round.sh r04 [bench|prof|pmc|packed|probes ...]   (default: all)
TAG=${1:-r04}; shift
STEPS=${@:-bench probes prof packed pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
PROF_ARGS="--steps 5 --warmup 1 --no-cpu-baseline --sustained-seconds 0 --skip-other --no-live-traffic"
db() { find $1 -name "*.db" | head -1; }
for step in $STEPS; do
case $step in
bench)
  (rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo; cat /sys/fs/cgroup/cpu.max) > $OUT/env.log 2>&1
  timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" ;;
probes)
  timeout 300 ./tools/bin/ubench2 > $OUT/ubench2.log 2>&1; echo "ubench2 rc=$?"
  timeout 200 python tools/power_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/power_probe.txt; echo "power_probe rc=$?" ;;
prof)
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py $PROF_ARGS > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?"
  f=$(db $OUT/prof)
  [ -n "$f" ] && python tools/prof_summary.py $f $OUT/kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py $PROF_ARGS" > /dev/null
  [ -n "$f" ] && python tools/prof_dispatches.py $f $OUT/kernel_dispatches.txt "per launch size: rocprofv3 --kernel-trace --stats -- python bench.py $PROF_ARGS (the 4096-workgroup NTT launches are BASELINE configs[1])" "ntt_|ct_mul|copy_kernel|reduce" > /dev/null
  rm -rf $OUT/prof ;;
packed)
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_pl -o pl -- ./examples/encrypted_gpt2_linear qkv 5 text 8 > $OUT/packed_linear.log 2> $OUT/prof_pl.err; echo "rocprof packed rc=$?"
  f=$(db $OUT/prof_pl); [ -n "$f" ] && python tools/prof_summary.py $f $OUT/packed_linear_8tokens_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- ./examples/encrypted_gpt2_linear qkv 5 text 8  (setup + 6 applications of 8 tokens)" > /dev/null
  rm -rf $OUT/prof_pl
  i=0
  for set in "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set -d $OUT/pmcp$i -o pmc -- ./examples/encrypted_gpt2_linear qkv 2 text 8 > $OUT/pmcp$i.log 2>&1
    f=$(db $OUT/pmcp$i)
    [ -n "$f" ] && python tools/pmc_summary.py $f "hoisted_qp|matvec_fold|relin_kernel|ntt_inv_galois|rescale" > $OUT/pmc_packed_pass$i.txt 2>&1
    echo "pmc packed pass $i rc=$? ($set)"; rm -rf $OUT/pmcp$i
  done ;;
pmc)
  # (round 5: contexts no longer probe at creation, so the counter runs see the default forms only without any switch)
  # the metric kernel and the NTT kernels at BASELINE configs[1] / configs[3] sizes (tools/ntt_bench.py <ntt polys> <ct_mul pairs>), one counter set per run
  i=0
  for set in "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set -d $OUT/pmc$i -o pmc -- python tools/ntt_bench.py 1024 8192 > $OUT/pmc$i.log 2>&1
    f=$(db $OUT/pmc$i)
    [ -n "$f" ] && python tools/pmc_summary.py $f "ntt_|ct_mul|reduce_" > $OUT/pmc_bench_pass$i.txt 2>&1
    echo "pmc pass $i rc=$? ($set)"; rm -rf $OUT/pmc$i
  done
  # steady-state batch with dispatch durations of the same launches, for VALU busy and the in-kernel clock
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace -d $OUT/pmcn1 -o pmc -- python tools/ntt_bench.py 8192 2048 > $OUT/pmcn1.log 2>&1
  f=$(db $OUT/pmcn1)
  [ -n "$f" ] && python tools/pmc_summary.py $f "ntt_|ct_mul" > $OUT/pmc_ntt_pass1.txt 2>&1
  [ -n "$f" ] && python tools/prof_summary.py $f $OUT/pmc_ntt_pass1_durations.txt "dispatch durations of the same run (rocprofv3 --pmc SQ_* --kernel-trace -- python tools/ntt_bench.py 8192 2048)" > /dev/null 2>&1
  echo "pmc ntt pass rc=$?"; rm -rf $OUT/pmcn1
  python tools/derive_round.py $OUT $TAG > /dev/null 2>&1; echo "derive rc=$?" ;;
esac
done
find $OUT -name "*.db" -delete
du -sh $OUT; ls $OUT
