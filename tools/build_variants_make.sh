#!/bin/bash
# Pre-build compile-time variants of the builder HERE (hipcc cross-compiles), so that the GPU box only measures them:
#   [VARIANT_SRC=traverse] tools/build_variants_make.sh name1 "-Dflags1" name2 "-Dflags2" ...   ->  tools/bin/variants/<name>.so
#   (VARIANT_SRC: which of build / traverse / scene / api .hip takes the flags; default build)
#   (on the GPU box)  tools/build_variants_run.sh out.txt
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-function -Wno-pass-failed"
mkdir -p tools/bin/variants
(cd nanort_amd/csrc && make -s >/dev/null)
src=${VARIANT_SRC:-build}
others=$(for o in api traverse build scene group; do [ $o != $src ] && echo -n "$o.o "; done)
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  (cd nanort_amd/csrc && /opt/rocm/bin/hipcc $F $flags -c $src.hip -o /tmp/${src}_variant_$name.o &&
   /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/variants/$name.so $others /tmp/${src}_variant_$name.o -ldl) &
done
wait
ls -la tools/bin/variants
