#!/bin/bash
# A/B of compile-time variants of the builder (build.hip): each argument is a string of -D flags ("" = as shipped).
# Builds a private copy of the library per variant, prints median build ms (1M-triangle plane fp32 / fp64, 70k sphere, 10M plane)
# and a fingerprint of the fp32 1M tree (must not change), restores the shipped library.
#   tools/build_variant_ab.sh out.txt "" "-DNRT_BUILD_TILE=1024"
set -e
cd "$(dirname "$0")/.."
out=$1; shift
: > "$out"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-function"
cp nanort_amd/lib/libnanort_hip.so /tmp/libnanort_hip.keep
trap 'cp /tmp/libnanort_hip.keep nanort_amd/lib/libnanort_hip.so' EXIT
for flags in "$@"; do
  (cd nanort_amd/csrc && /opt/rocm/bin/hipcc $F $flags -c build.hip -o /tmp/build_probe.o &&
   /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libnanort_hip.so api.o traverse.o /tmp/build_probe.o scene.o)
  echo "== ${flags:-as shipped}" >> "$out"
  python - >> "$out" 2>&1 <<'PY'
import sys, hashlib, numpy as np
sys.path.insert(0, '.')
from nanort_amd import BVHAccel, TriangleMesh, scenes
def med(a, m, n=7):
    ts = []
    for _ in range(n):
        a.Build(m.num_faces, m); ts.append(a.LastBuildMs())
    return float(np.median(ts[2:]))
v, f = scenes.plane(1000, 500)
a = BVHAccel(np.float32); m = TriangleMesh(v, f); t32 = med(a, m)
nodes, idx = a.GetTree(); fp = hashlib.md5(nodes.tobytes() + idx.tobytes()).hexdigest()[:12]
a64 = BVHAccel(np.float64); m64 = TriangleMesh(v.astype(np.float64), f); t64 = med(a64, m64)
sv, sf = scenes.sphere(); b = BVHAccel(np.float32); ts = med(b, TriangleMesh(sv, sf))
bv, bf = scenes.plane(2500, 2000); c = BVHAccel(np.float32); tb = med(c, TriangleMesh(bv, bf), 5)
print("plane1M f32 %.3f ms  f64 %.3f ms  sphere70k %.3f ms  plane10M %.3f ms  tree %s" % (t32, t64, ts, tb, fp))
PY
done
cat "$out"
