"""Randomised soak of the GPU BUILDER alone: random meshes of 300 ... 600 000 triangles (smooth, clustered / lopsided, grids,
soups), fp32 and fp64, random build options, Morton pre-pass at random — every tree validated invariant by invariant
(tests/bvh_check.py: permutation, pre-order, exact bounds, leaf rule, the statistics the builder reports) and built twice
(same bits), and a sample of rays traced against the CPU restatement on the same node array.
    python tests/checks/fuzz_build.py [seconds] [seed]"""
import hashlib, sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from nanort_amd import BVHAccel, TriangleMesh, scenes
from nanort_amd.wire import default_build_options, ray_dtype
from bvh_check import validate_bvh
from oracle.bindings import Oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
orc = Oracle()
t_end = time.time() + budget
rounds = tris = 0
big = 0
while time.time() < t_end:
    real = np.float32 if rng.random() < 0.7 else np.float64
    n = int(np.exp(rng.uniform(np.log(300), np.log(600000))))
    kind = int(rng.integers(0, 4))
    if kind == 0:    # displaced grid (coherent order, like the bench meshes)
        gx = max(2, int(np.sqrt(n / 2) * rng.uniform(0.5, 2.0))); gy = max(1, n // (2 * gx))
        v, f = scenes.plane(gx, gy)
    elif kind == 1:  # soup over shared vertices (lopsided boxes)
        nv = max(3, n // 6)
        v = rng.uniform(-1, 1, size=(nv, 3)).astype(np.float32); f = rng.integers(0, nv, size=(n, 3)).astype(np.uint32)
    elif kind == 2:  # tight cluster + far outliers (lopsided splits, deep trees)
        nv = max(3, n // 4)
        v = (rng.normal(size=(nv, 3)) * 0.01).astype(np.float32); v[: max(1, nv // 200)] += rng.uniform(2, 9)
        f = rng.integers(0, nv, size=(n, 3)).astype(np.uint32)
    else:            # small local triangles in random order (incoherent input)
        c = rng.uniform(-5, 5, size=(n, 1, 3)); v = (c + rng.normal(size=(n, 3, 3)) * 0.02).reshape(-1, 3).astype(np.float32)
        f = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    n = f.shape[0]
    o = default_build_options(real)
    o["min_leaf_primitives"] = int(rng.choice([1, 2, 4, 8, 16]))
    o["bin_size"] = int(rng.choice([4, 8, 16, 64, 64, 64]))
    o["max_tree_depth"] = int(rng.choice([256, 256, 256, 40, 18]))
    a = BVHAccel(real)
    a.SetTunable("morton", int(rng.random() < 0.15))
    m = TriangleMesh(v.astype(real), f)
    assert a.Build(n, m, o)
    nodes, idx = a.GetTree()
    st = a.GetStatistics()
    validate_bvh(nodes, idx, v.astype(real), f, min_leaf=o["min_leaf_primitives"], max_depth=o["max_tree_depth"], stats=st, low_side_first=True)
    h1 = hashlib.md5(nodes.tobytes() + idx.tobytes()).hexdigest()
    assert a.Build(n, m, o)
    nodes2, idx2 = a.GetTree()
    assert hashlib.md5(nodes2.tobytes() + idx2.tobytes()).hexdigest() == h1, "the builder is not deterministic"
    # a few rays against the restatement on the same node array
    R = ray_dtype(real); nr = 2000
    rays = np.zeros(nr, dtype=R)
    lo, hi = v.min(0), v.max(0)
    rays["org"] = rng.uniform(lo - 1, hi + 1, size=(nr, 3)); tgt = rng.uniform(lo, hi, size=(nr, 3))
    rays["dir"] = tgt - rays["org"]; rays["min_t"] = 0; rays["max_t"] = 1e30
    h, mk = a.TraverseBatch(rays)
    oh, om = orc.traverse(nodes, idx, v.astype(real), f, rays)
    hit = om == 1
    if not (np.array_equal(mk, om) and all(np.ascontiguousarray(h[k][hit]).tobytes() == np.ascontiguousarray(oh[k][hit]).tobytes() for k in ("t", "u", "v", "prim_id"))):  # (fieldwise: the fp64 record has padding)
        bad = np.nonzero((mk != om) | ((om == 1) & ((h["t"] != oh["t"]) | (h["prim_id"] != oh["prim_id"]) | (h["u"] != oh["u"]) | (h["v"] != oh["v"]))))[0]
        print("MISMATCH round", rounds, "kind", kind, "n", n, real.__name__, dict(o) if isinstance(o, dict) else o, "morton", a.GetTunable("morton"), "bad rays", len(bad), "mask diffs", int((mk != om).sum()))
        for b in bad[:5]:
            print("   ray", b, rays[b], "gpu", mk[b], h[b], "oracle", om[b], oh[b])
        np.savez("gpurun_out/fuzz_build_case.npz", v=v, f=f, rays=rays, nodes=nodes, idx=idx)
        raise AssertionError("records differ from the restatement")
    rounds += 1; tris += n; big += int(n >= 131072)
print("fuzz_build ok: %d builds validated (%d of >= 131072 triangles), %d triangles in all, seed %d" % (rounds, big, tris, seed))
