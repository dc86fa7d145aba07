#!/bin/bash
# Build the HIP library with extra flags into csrc/build/var_<tag>.so for same-box A/B timing:
#   bash tools/ab_variant.sh <tag> [-DFLAG ...]      then on the box:  DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_<tag>.so python tools/ntt_bench.py
set -e
TAG=$1; shift
cd "$(dirname "$0")/../deeppowers_amd/csrc"
D=build/ab_$TAG; mkdir -p $D
for f in dpfhe_cabi k_ntt_fold k_ntt_shoup k_ctmul_fold k_ctmul_shoup k_ctmul_var k_bx_fold k_bx_shoup; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Rpass-analysis=kernel-resource-usage "$@" -c -o $D/$f.o $f.hip 2> $D/$f.log &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o build/var_$TAG.so $D/*.o -ldl
ls -la build/var_$TAG.so
