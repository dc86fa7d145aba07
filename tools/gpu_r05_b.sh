#!/bin/bash
# round 5, GPU call B: the giant-step key inner products (relin MODE 4, N = 8192) as one 256-thread workgroup per (item, limb, half) against the
# 512-thread kernel, same box, alternated; parity of everything on the packed pipeline; counters of both forms of the batched N = 8192 transforms.
# (arms as built at that commit: HEAD = halves kernels ON (today: -DDPFHE_N13_HALVES=1), var_base.so = the shipped 512-thread kernels, var_r3.so = -DDPFHE_RELIN_HALF_OCC=3)
OUT=gpurun_out/r05b; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bsgs_qp.py tests/test_rlwe_semantics.py tests/test_gpu_cpp_api.py -x -q -p no:cacheprovider -m gpu 2>&1 | tail -5 | tee $OUT/pytest_subset.txt
for i in 1 2 3; do
  for v in base HEAD r3; do
    if [ $v = HEAD ]; then unset DPFHE_AB_LIB; else export DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_$v.so; fi
    timeout 300 python tools/ab_relin13.py 2>&1 | grep RELIN13
  done
done | tee $OUT/ab_relin_half.txt
for v in base HEAD; do
  if [ $v = HEAD ]; then unset DPFHE_AB_LIB; else export DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_$v.so; fi
  timeout 300 python tools/ab_packed.py 8 64 2>&1 | grep -E "median"
done | tee $OUT/ab_packed.txt
# counters: N = 8192 batched transforms, both forms (one counter set, with the dispatch durations of the same launches)
for v in base HEAD; do
  if [ $v = HEAD ]; then unset DPFHE_AB_LIB; else export DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_$v.so; fi
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace -d $OUT/pmc_$v -o pmc -- python tools/ab_sustained.py 0.05 > $OUT/pmc_$v.log 2>&1
  f=$(find $OUT/pmc_$v -name "*.db" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f "ntt_" > $OUT/pmc_ntt13_$v.txt 2>&1
  [ -n "$f" ] && python tools/prof_summary.py $f $OUT/pmc_ntt13_${v}_durations.txt "dispatch durations of the same run" > /dev/null 2>&1
  rm -rf $OUT/pmc_$v
done
unset DPFHE_AB_LIB
# counters of the key inner products, both forms
for v in base HEAD; do
  if [ $v = HEAD ]; then unset DPFHE_AB_LIB; else export DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_$v.so; fi
  i=0
  for set in "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/pmcr_$v$i -o pmc -- python tools/ab_relin13.py > $OUT/pmcr_$v$i.log 2>&1
    f=$(find $OUT/pmcr_$v$i -name "*.db" | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py $f "relin" > $OUT/pmc_relin13_${v}_pass$i.txt 2>&1
    [ -n "$f" ] && [ $i = 3 ] && python tools/prof_summary.py $f $OUT/pmc_relin13_${v}_durations.txt "dispatch durations of the same run" > /dev/null 2>&1
    rm -rf $OUT/pmcr_$v$i
  done
done
find $OUT -name "*.db" -delete; ls $OUT
