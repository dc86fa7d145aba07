#!/bin/bash
# round 4, GPU call B: looped quad form (pairs per workgroup, prefetch depth), streamed baby-step pass, key-major giant steps - same-box A/B + correctness
TAG=${1:-r04b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
db() { find $1 -name "*.db" | head -1; }
V=deeppowers_amd/csrc/build
timeout 600 python -m pytest tests/test_gpu_bsgs_qp.py tests/test_gpu_parity.py -x -q -m gpu -k "qp or galois or bsgs or every_form" 2>&1 | tail -5 | tee $OUT/pytest_subset.txt
for k in 4 2 8; do
  DPFHE_CTMUL_LOOP_PAIRS=$k timeout 200 python tools/ab_forms.py 8192 n4096 2>&1 | grep -E "quadloop|quad " >> $OUT/ab_forms.txt
done
DPFHE_AB_LIB=$V/var_pf1.so timeout 200 python tools/ab_forms.py 8192 n4096 2>&1 | grep -E "quadloop|quad " >> $OUT/ab_forms.txt
timeout 200 python tools/ab_forms.py 8192 2>&1 | grep -v amdgpu.ids >> $OUT/ab_forms_full.txt
cat $OUT/ab_forms.txt; grep -E "^#|N=8192" -A0 $OUT/ab_forms_full.txt | head; grep "HEAD" $OUT/ab_forms_full.txt | tail -16
for arm in HEAD fused relinold; do
  case $arm in
    HEAD) timeout 200 python tools/ab_packed.py 8 64 2>&1 | grep -v amdgpu.ids | sed "s/^/[stream+keymajor] /" >> $OUT/ab_packed.txt ;;
    fused) DPFHE_HOISTED_QP=fused timeout 200 python tools/ab_packed.py 8 64 2>&1 | grep -E "rotate_hoisted" | sed "s/^/[old fused baby steps] /" >> $OUT/ab_packed.txt ;;
    relinold) DPFHE_AB_LIB=$V/var_relinold.so timeout 200 python tools/ab_packed.py 8 64 2>&1 | grep -E "switch_key_qp" | sed "s/^/[old giant-step layout] /" >> $OUT/ab_packed.txt ;;
  esac
done
timeout 200 python tools/ab_packed.py 1 64 2>&1 | grep -E "rotate_hoisted|switch_key" | sed "s/^/[1 token] /" >> $OUT/ab_packed.txt
DPFHE_HOISTED_QP=fused timeout 200 python tools/ab_packed.py 1 64 2>&1 | grep -E "rotate_hoisted" | sed "s/^/[1 token, old fused baby steps] /" >> $OUT/ab_packed.txt
cat $OUT/ab_packed.txt
timeout 200 python tools/ctmul_trace.py 8192 2>&1 | grep -v amdgpu.ids > $OUT/ctmul_trace_8192.txt; head -3 $OUT/ctmul_trace_8192.txt
for t in 8 1; do timeout 300 ./examples/encrypted_gpt2_linear all 3 text $t 2>&1 | tail -6 | sed "s/^/[$t tokens] /" >> $OUT/packed_linear.txt; done; cat $OUT/packed_linear.txt
i=0
for set in "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $OUT/pmcp$i -o pmc -- ./examples/encrypted_gpt2_linear qkv 2 text 8 > $OUT/pmcp$i.log 2>&1
  f=$(db $OUT/pmcp$i)
  [ -n "$f" ] && python tools/pmc_summary.py $f "hoisted_qp|matvec_fold|relin_kernel|ntt_inv_galois|rescale" > $OUT/pmc_packed_pass$i.txt 2>&1
  echo "pmc packed pass $i rc=$? ($set)"; grep -A3 -E "hoisted_qp|relin_kernel" $OUT/pmc_packed_pass$i.txt | head -24; rm -rf $OUT/pmcp$i
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_pl -o pl -- ./examples/encrypted_gpt2_linear qkv 5 text 8 > $OUT/packed_linear_prof.log 2> $OUT/prof_pl.err
f=$(db $OUT/prof_pl); [ -n "$f" ] && python tools/prof_summary.py $f $OUT/packed_linear_8tokens_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- ./examples/encrypted_gpt2_linear qkv 5 text 8  (setup + 6 applications of 8 tokens)" > /dev/null; head -20 $OUT/packed_linear_8tokens_kernel_stats.txt
rm -rf $OUT/prof_pl
find $OUT -name "*.db" -delete
