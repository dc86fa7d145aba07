#!/bin/bash
# round 6, GPU call O: the whole GPU suite + smoke on the clean build
OUT=gpurun_out/r06o; mkdir -p $OUT; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -q -p no:cacheprovider -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" | tail -6 | tee $OUT/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
