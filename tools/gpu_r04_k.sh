#!/bin/bash
# final-code kernel statistics (the r04 set was collected before the up-front baby-step kernel and the multi-block example existed)
OUT=gpurun_out/r04e; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/collect_round.sh r04e prof
db() { find $1 -name "*.db" | head -1; }
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_pl -o pl -- ./examples/encrypted_gpt2_linear qkv 5 text 8 > $OUT/packed_linear.log 2> $OUT/prof_pl.err; echo "rocprof packed rc=$?"
f=$(db $OUT/prof_pl); [ -n "$f" ] && python tools/prof_summary.py $f $OUT/packed_linear_8tokens_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- ./examples/encrypted_gpt2_linear qkv 5 text 8  (setup + 6 applications of 8 tokens)" > /dev/null
rm -rf $OUT/prof_pl
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_st -o st -- ./examples/encrypted_gpt2_stack 8 2 text 10 > $OUT/stack.log 2> $OUT/prof_st.err; echo "rocprof stack rc=$?"
f=$(db $OUT/prof_st); [ -n "$f" ] && python tools/prof_summary.py $f $OUT/stack_3blocks_8tokens_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- ./examples/encrypted_gpt2_stack 8 2 text 10  (setup + 3 passes over three blocks on ten data limbs, 8 tokens)" > /dev/null
rm -rf $OUT/prof_st
find $OUT -name "*.db" -delete; ls -la $OUT; tail -4 $OUT/stack.log | cut -c1-300
