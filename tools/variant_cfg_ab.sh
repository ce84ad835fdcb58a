#!/bin/bash
# A/B of compile-time variants of the traversal kernels on any bench configs (tools/tune_probe.py), each against a private copy
# of the library.  Usage (GPU box, repo root): tools/variant_cfg_ab.sh out.txt C4tile,C3 "" "-DNRT_W4_WAVES=6"
set -e
cd "$(dirname "$0")/.."
out=$1; cfgs=$2; shift; shift
: > "$out"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-function"
cp nanort_amd/lib/libnanort_hip.so /tmp/libnanort_hip.keep
trap 'cp /tmp/libnanort_hip.keep nanort_amd/lib/libnanort_hip.so' EXIT
for flags in "$@"; do
  (cd nanort_amd/csrc && /opt/rocm/bin/hipcc $F $flags -c traverse.hip -o /tmp/traverse_probe.o && /opt/rocm/bin/hipcc $F $flags -c api.hip -o /tmp/api_probe.o &&
   /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libnanort_hip.so /tmp/api_probe.o /tmp/traverse_probe.o build.o scene.o)
  echo "== ${flags:-as shipped}" >> "$out"
  ROUNDS=3 python tools/tune_probe.py "$cfgs" "dict()" 2>&1 | grep -v amdgpu >> "$out"
done
cat "$out"
