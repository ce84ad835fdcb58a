#!/bin/bash
# Measure the pre-built builder variants of tools/bin/variants (see build_variants_make.sh): median build ms of the
# 1M-triangle plane (fp32 / fp64), the 70k sphere and the 10M plane, and a fingerprint of the fp32 1M tree and the sphere
# tree (must not change).  Restores the shipped library.
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/build_variants.txt}
: > "$out"
cp nanort_amd/lib/libnanort_hip.so /tmp/libnanort_hip.keep
trap 'cp /tmp/libnanort_hip.keep nanort_amd/lib/libnanort_hip.so' EXIT
for lib in /tmp/libnanort_hip.keep tools/bin/variants/*.so /tmp/libnanort_hip.keep; do
  cp "$lib" nanort_amd/lib/libnanort_hip.so
  echo "== $(basename $lib .so)" >> "$out"
  timeout 300 python - >> "$out" 2>&1 <<'PY'
import sys, hashlib, numpy as np
sys.path.insert(0, '.')
from nanort_amd import BVHAccel, TriangleMesh, scenes
def med(a, m, n=9):
    ts = []
    for _ in range(n):
        a.Build(m.num_faces, m); ts.append(a.LastBuildMs())
    return float(np.median(ts[2:])), float(np.min(ts[2:]))
def fp(a):
    nodes, idx = a.GetTree(); return hashlib.md5(nodes.tobytes() + idx.tobytes()).hexdigest()[:10]
v, f = scenes.plane(1000, 500)
a = BVHAccel(np.float32); m = TriangleMesh(v, f); t32 = med(a, m); f32 = fp(a)
a64 = BVHAccel(np.float64); m64 = TriangleMesh(v.astype(np.float64), f); t64 = med(a64, m64); f64 = fp(a64)
sv, sf = scenes.sphere(); b = BVHAccel(np.float32); ts = med(b, TriangleMesh(sv, sf)); fs = fp(b)
bv, bf = scenes.plane(2500, 2000); c = BVHAccel(np.float32); tb = med(c, TriangleMesh(bv, bf), 5)
print("plane1M f32 %.3f (min %.3f) f64 %.3f (min %.3f) sphere70k %.3f plane10M %.3f  trees %s %s %s" % (t32 + t64 + (ts[0], tb[0], f32, f64, fs)))
PY
done
cat "$out"
