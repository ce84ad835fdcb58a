#!/bin/bash
# A/B of pre-built libraries (tools/bin/variants/*.so) on the scene probe (fixture, 10 000 and 100 000 instances), each twice, interleaved.
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/scene_variants.txt}
: > "$out"
cp nanort_amd/lib/libnanort_hip.so /tmp/libnanort_hip.keep
trap 'cp /tmp/libnanort_hip.keep nanort_amd/lib/libnanort_hip.so' EXIT
for pass in 1 2; do
  for lib in tools/bin/variants/*.so; do
    cp "$lib" nanort_amd/lib/libnanort_hip.so
    echo "== $(basename $lib .so) (pass $pass)" >> "$out"
    timeout 200 python tools/scene_probe.py ${COUNTS:-10000 100000} 2>&1 | grep -E "fixture|instances" | sed -E 's/\(.*1920x1080\)//; s/ \| its plane.*//; s/of a 2208-triangle mesh: commit [0-9.]+ ms, //' >> "$out"
  done
done
cat "$out"
