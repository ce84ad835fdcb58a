#!/bin/bash
# A/B of compile-time variants of the traversal kernels on C3's two waves.  Each argument is a string of -D flags (the
# empty string = as shipped), optionally followed by "|ENV=VALUE ..." to set for the timing run.  Builds a private copy of
# the library per variant, times it with tools/trav_tune.py, restores the shipped library.  Run on the GPU box from the
# (VARIANT_C5=1 also times the fp64 primary wave, tools/c5_probe.py; VARIANT_SPHERES=1 the 1M-sphere workload, tools/sphere_probe.py)
# repo root:  tools/variant_ab.sh out.txt "" "-DNRT_W4_WAVES=6" "-DNRT_PROBE_EXTRA_LOADS=1|NRT_WIDE4=0"
set -e
cd "$(dirname "$0")/.."
out=$1; shift
: > "$out"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-function"
cp nanort_amd/lib/libnanort_hip.so /tmp/libnanort_hip.keep
trap 'cp /tmp/libnanort_hip.keep nanort_amd/lib/libnanort_hip.so' EXIT
for spec in "$@"; do
  flags=${spec%%|*}; envs=""; [[ "$spec" == *"|"* ]] && envs=${spec#*|}
  (cd nanort_amd/csrc && /opt/rocm/bin/hipcc $F $flags -c traverse.hip -o /tmp/traverse_probe.o &&
   /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libnanort_hip.so api.o /tmp/traverse_probe.o build.o scene.o)
  echo "== ${flags:-as shipped} ${envs}" >> "$out"
  env $envs python tools/trav_tune.py "dict()" "dict()" 2>&1 | grep primary >> "$out"
  [ -n "$VARIANT_C5" ] && env $envs python tools/c5_probe.py 2>&1 | grep float64 >> "$out"
  [ -n "$VARIANT_SPHERES" ] && env $envs python tools/sphere_probe.py 2>&1 | grep spheres >> "$out"
done
cat "$out"
