#!/bin/bash
# round 6, GPU call Z: per-kernel profile of the activated block at N = 16384, two tokens per ciphertext (16 tokens in 8 ciphertexts), 30 applications so that the applies outweigh the set-up
OUT=gpurun_out/r06z; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -- $GRAFT_REPO_ROOT/examples/encrypted_gpt2_block_act 16 30 json ladder 14 2 > $GRAFT_REPO_ROOT/$OUT/run.log 2>&1
cd $GRAFT_REPO_ROOT; tail -1 $OUT/run.log | cut -c1-300
python tools/prof_summary.py $OUT/prof > $OUT/kernel_stats.txt 2>&1 || ls -R $OUT/prof | head
head -40 $OUT/kernel_stats.txt | cut -c1-200
