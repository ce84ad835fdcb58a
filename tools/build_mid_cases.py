"""One (case, build_mid) build in this process: prints the tree fingerprint and the median build ms (tools/build_mid_probe.sh
runs the grid in separate processes so that a faulting case does not hide the others)."""
import hashlib, sys
import numpy as np
sys.path.insert(0, '.')
from nanort_amd import BVHAccel, TriangleMesh, scenes
from nanort_amd.wire import default_build_options
from tests.bvh_check import validate_bvh

case, mid = sys.argv[1], int(sys.argv[2])
real = np.float64 if case.endswith("f64") else np.float32
rng = np.random.default_rng(5)
opt = {}
if case.startswith("soup200k"):
    v = rng.uniform(-1, 1, size=(30000, 3)).astype(np.float32); f = rng.integers(0, 30000, size=(200000, 3)).astype(np.uint32)
elif case.startswith("soup20k"):
    v = rng.uniform(-1, 1, size=(3000, 3)).astype(np.float32); f = rng.integers(0, 3000, size=(20000, 3)).astype(np.uint32)
elif case.startswith("clustered"):
    v = (rng.normal(size=(5000, 3)) * 0.01).astype(np.float32); v[:100] += 5.0
    f = rng.integers(0, 5000, size=(60000, 3)).astype(np.uint32)
elif case.startswith("plane2400"):
    v, f = scenes.plane(40, 30)
elif case.startswith("plane10m"):
    v, f = scenes.plane(2500, 2000)
elif case.startswith("plane1m"):
    v, f = scenes.plane(1000, 500)
elif case.startswith("sphere"):
    v, f = scenes.sphere()
else:
    raise SystemExit("unknown case")
a = BVHAccel(real)
a.SetTunable("build_mid", mid)
o = default_build_options(real)
m = TriangleMesh(v.astype(real), f)
ts = []
for _ in range(6):
    assert a.Build(m.num_faces, m, o)
    ts.append(a.LastBuildMs())
nodes, idx = a.GetTree()
if f.shape[0] <= 300000:
    validate_bvh(nodes, idx, v.astype(real), f, stats=a.GetStatistics(), low_side_first=True)
print("%-16s mid %-5d %.3f ms  %s  nodes %d" % (case, mid, float(np.median(ts[2:])), hashlib.md5(nodes.tobytes() + idx.tobytes()).hexdigest()[:12], nodes.shape[0]), flush=True)
