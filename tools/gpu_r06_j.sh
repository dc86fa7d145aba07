#!/bin/bash
# round 6, GPU call J: where the activated block at N = 16384 spends its time (kernel stats of examples/encrypted_gpt2_block_act ... ladder 14)
OUT=gpurun_out/r06j; mkdir -p $OUT; export TMPDIR=/tmp
db() { find $1 -name "*.db" | head -1; }
timeout 600 ./examples/encrypted_gpt2_block_act 8 3 json ladder 14 2>&1 | tail -3 | cut -c1-600
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o blk -- ./examples/encrypted_gpt2_block_act 8 6 json ladder 14 > $OUT/block_act_n16384.log 2> $OUT/prof.err
f=$(db $OUT/prof); [ -n "$f" ] && python tools/prof_summary.py $f $OUT/block_act_n16384_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- ./examples/encrypted_gpt2_block_act 8 6 json ladder 14  (N = 16384: setup + 7 applications of the activated block on 8 tokens)" > /dev/null
rm -rf $OUT/prof; find $OUT -name "*.db" -delete
head -30 $OUT/block_act_n16384_kernel_stats.txt | cut -c1-170
