#!/bin/bash
# Build the HIP library with extra flags into csrc/build/var_<tag>.so for same-box A/B timing:
#   bash tools/ab_variant.sh <tag> [-DFLAG ...]      then on the box:  DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_<tag>.so python tools/ntt_bench.py
set -e
TAG=$1; shift
cd "$(dirname "$0")/../deeppowers_amd/csrc"
D=build/ab_$TAG; mkdir -p $D
for src in dpfhe_cabi.hip k_*.hip; do
  f=${src%.hip}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Rpass-analysis=kernel-resource-usage "$@" -c -o $D/$f.o $f.hip 2> $D/$f.log &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o build/var_$TAG.so $D/*.o -ldl
ls -la build/var_$TAG.so
