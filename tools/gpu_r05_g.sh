#!/bin/bash
# round 5, GPU call G: the round's evidence (tools/collect_round.sh r05: bench line, probes, rocprofv3 kernel stats, counter passes) + where the N = 16384 layer spends its time
bash tools/collect_round.sh r05 bench probes prof packed pmc 2>&1 | tail -40
OUT=gpurun_out/r05; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_pl14 -o pl -- ./examples/encrypted_gpt2_linear qkv 5 text 8 14 > $OUT/packed_linear_n16384.log 2> $OUT/prof_pl14.err
f=$(find $OUT/prof_pl14 -name "*.db" | head -1); [ -n "$f" ] && python tools/prof_summary.py $f $OUT/packed_linear_n16384_8tokens_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- ./examples/encrypted_gpt2_linear qkv 5 text 8 14  (N = 16384: setup + 6 applications of 8 tokens)" > /dev/null
rm -rf $OUT/prof_pl14; find $OUT -name "*.db" -delete; ls $OUT
