#!/bin/bash
# A/B of pre-built libraries (tools/bin/variants/*.so, e.g. from tools/build_variants_make.sh or a build of another commit)
# on bench configs with tools/tune_probe.py; every library is measured twice, interleaved (A B A B), against drift.
#   (GPU box, repo root)  tools/lib_variants_ab.sh out.txt C3,C4tile
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/lib_variants.txt}; cfgs=${2:-C3}
: > "$out"
cp nanort_amd/lib/libnanort_hip.so /tmp/libnanort_hip.keep
trap 'cp /tmp/libnanort_hip.keep nanort_amd/lib/libnanort_hip.so' EXIT
for pass in 1 2; do
  for lib in tools/bin/variants/*.so; do
    cp "$lib" nanort_amd/lib/libnanort_hip.so
    echo "== $(basename $lib .so) (pass $pass)" >> "$out"
    ROUNDS=${ROUNDS:-3} timeout 300 python tools/tune_probe.py "$cfgs" "dict()" 2>&1 | grep -v amdgpu >> "$out"
  done
done
cat "$out"
