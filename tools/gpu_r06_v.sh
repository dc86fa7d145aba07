#!/bin/bash
# round 6, GPU call V: the bench step with the multiply in K chunks and each chunk's reduce chasing it (Infinity-Cache reuse), non-temporal stores (HEAD) and plain stores (var_st.so)
OUT=gpurun_out/r06v; mkdir -p $OUT; export TMPDIR=/tmp
for v in HEAD st HEAD st; do
  if [ $v = HEAD ]; then unset DPFHE_AB_LIB; else export DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_$v.so; fi
  timeout 600 python tools/chunked_step.py 1 4 8 16 32 64 2>&1 | grep "CHUNK\|Error\|error"
done | tee $OUT/chunked_step.txt
