#!/bin/bash
# round 6, GPU call E: where the giant-step key inner products fetch more than their algorithmic bytes (tools/giant_traffic.py), HEAD against non-temporal key tiles
OUT=gpurun_out/r06e; mkdir -p $OUT; export TMPDIR=/tmp
for v in HEAD ntkeys; do
  if [ $v = HEAD ]; then unset DPFHE_AB_LIB; else export DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_$v.so; fi
  timeout 300 python tools/giant_traffic.py 2>&1 | grep GIANT | tee $OUT/timing_$v.txt
  i=0
  for set in "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $set -d $OUT/p_${v}_$i -o pmc -- python tools/giant_traffic.py > $OUT/p_${v}_$i.log 2>&1
    f=$(find $OUT/p_${v}_$i -name "*.db" | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py $f "relin_kernel" > $OUT/pmc_${v}_$i.txt 2>&1
    echo "== $v pass $i: $set"; cat $OUT/pmc_${v}_$i.txt | head -40
  done
done
find $OUT -name "*.db" -delete; rm -rf $OUT/p_*_[0-9]
