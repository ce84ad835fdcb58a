"""The builder's block-resident middle phase (tunable build_mid: nodes of at most this many primitives leave the
level-synchronous phase and are finished level by level by one block each, k_mid): the node array + index permutation must be
the SAME at every hand-over size; median build ms per size.
    python tools/build_mid_probe.py [sizes...]        (default 0 512 1024 2048)"""
import hashlib, sys
import numpy as np
sys.path.insert(0, '.')
from nanort_amd import BVHAccel, TriangleMesh, scenes
from nanort_amd.wire import default_build_options

sizes = [int(x) for x in sys.argv[1:]] or [0, 512, 1024, 2048]


def fp(a):
    nodes, idx = a.GetTree()
    return hashlib.md5(nodes.tobytes() + idx.tobytes()).hexdigest()[:12]


def run(name, real, v, f, reps=7, **opt):
    out = []
    for mid in sizes:
        a = BVHAccel(real)
        a.SetTunable("build_mid", mid)
        o = default_build_options(real)
        for k, val in opt.items():
            o[k] = val
        m = TriangleMesh(v.astype(real), f)
        ts = []
        for _ in range(reps):
            assert a.Build(m.num_faces, m, o)
            ts.append(a.LastBuildMs())
        out.append((mid, float(np.median(ts[2:])), fp(a)))
    same = len({x[2] for x in out}) == 1
    print("%-28s %s  %s" % (name, "SAME TREE" if same else "TREES DIFFER", "  ".join("mid %d: %.3f ms %s" % x for x in out)), flush=True)
    return same


ok = True
v, f = scenes.plane(1000, 500)
ok &= run("plane 1M f32", np.float32, v, f)
ok &= run("plane 1M f64", np.float64, v, f)
sv, sf = scenes.sphere()
ok &= run("sphere 70K f32", np.float32, sv, sf)
ok &= run("sphere 70K f32 leaf1 bins8", np.float32, sv, sf, min_leaf_primitives=1, bin_size=8)
ok &= run("sphere 70K f32 depth12", np.float32, sv, sf, max_tree_depth=12)
rng = np.random.default_rng(5)
rv = rng.uniform(-1, 1, size=(30000, 3)).astype(np.float32)
rf = rng.integers(0, 30000, size=(200000, 3)).astype(np.uint32)
ok &= run("random soup 200K f32", np.float32, rv, rf)
ok &= run("random soup 200K f64", np.float64, rv, rf)
cv = (rng.normal(size=(5000, 3)) * 0.01).astype(np.float32)
cv[:100] += 5.0  # a far cluster: lopsided splits
cf = rng.integers(0, 5000, size=(60000, 3)).astype(np.uint32)
ok &= run("clustered 60K f32", np.float32, cv, cf)
v3, f3 = scenes.plane(40, 30)
ok &= run("plane 2400 f32", np.float32, v3, f3)
bv, bf = scenes.plane(2500, 2000)
ok &= run("plane 10M f32", np.float32, bv, bf, reps=5)
print("ALL SAME" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
