"""Randomised parity soak for the widened rows: sphere and cylinder primitives and two-level
scenes, GPU vs the CPU restatements on the SAME node arrays.  Spheres: t / prim_id / mask bit for bit, u / v within
1e-6 (double atan2 / acos); cylinders: every field bit for bit (NaNs included); scenes: every field bit for bit, with
random node transforms (rotation, non-uniform and mirrored scale, translation, nearly flat), up to 90 nodes (more than
nanosg's 64-entry list), shared meshes, coincident instances and bounded ray intervals.
Usage: python tests/checks/fuzz_prims_scenes.py [seconds] [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from nanort_amd import BVHAccel, CylinderGeometry, Scene, SphereGeometry, TriangleMesh, scenes  # noqa: E402
from nanort_amd.wire import RAY_F32, default_build_options  # noqa: E402
from oracle import bindings as ob  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
orc = ob.Oracle()
sph, cyl = ob.SphereOracle(), ob.CylinderOracle()


def random_rays(m, targets, spread=6.0, bounded=0.5):
    r = np.zeros(m, dtype=RAY_F32)
    r["org"] = (rng.normal(size=(m, 3)) * spread).astype(np.float32)
    inside = rng.random(m) < 0.15  # origins on / inside the primitives
    r["org"][inside] = targets[rng.integers(0, targets.shape[0], int(inside.sum()))]
    tgt = targets[rng.integers(0, targets.shape[0], m)] + rng.normal(size=(m, 3)).astype(np.float32) * 0.05
    d = tgt - r["org"]
    nrm = np.linalg.norm(d, axis=1, keepdims=True)
    unit = rng.random((m, 1)) < 0.7
    d = np.where(unit & (nrm > 0), d / np.where(nrm > 0, nrm, 1), d)
    d[: m // 16] = rng.integers(-1, 2, size=(m // 16, 3))  # axis-parallel and zero directions
    d[m // 16: m // 16 + 8, 0] = np.nan
    d[m // 16 + 8: m // 16 + 16, 2] = np.inf
    r["dir"] = d.astype(np.float32)
    r["min_t"] = rng.choice([0.0, 0.0, 0.0, 1e-3, 0.7], m).astype(np.float32)
    far = rng.choice([1e30, 3.4028234663852886e38, 5.0, 1.5, 0.0, -1.0], m)
    r["max_t"] = np.where(rng.random(m) < bounded, far, 1e30).astype(np.float32)
    return r


def build_options():
    bo = default_build_options(np.float32)
    bo["min_leaf_primitives"] = int(rng.choice([1, 2, 4, 4, 8, 16]))
    bo["bin_size"] = int(rng.choice([2, 8, 64, 64, 200]))
    bo["max_tree_depth"] = int(rng.choice([256, 256, 256, 10, 2]))
    return bo


def prim_range(n):
    if rng.random() < 0.6:
        return (0, 0x7FFFFFFF)
    lo = int(rng.integers(0, n))
    return (lo, int(rng.integers(lo, n + 2)))


def sphere_round():
    n = int(rng.choice([1, 2, 5, 33, 400, 3000]))
    c = (rng.normal(size=(n, 3)) * rng.choice([0.5, 3.0])).astype(np.float32)
    if rng.random() < 0.3:
        c = np.round(c)  # lattice: coincident centres, tangent spheres
    r = rng.uniform(0.01, rng.choice([0.1, 1.0, 4.0]), n).astype(np.float32)
    k = max(1, n // 10)
    r[:k] = rng.choice([0.0, -0.2, 1e-20, 50.0], k)
    a = BVHAccel(np.float32)
    assert a.Build(n, SphereGeometry(c, r), build_options())
    nodes, idx = a.GetTree()
    rays = random_rays(3000, c)
    rg = prim_range(n)
    from nanort_amd.wire import default_trace_options

    o = default_trace_options()
    o["prim_ids_range"] = rg
    h, m = a.TraverseBatch(rays, o)
    oh, om = sph.traverse(nodes, idx, c, r, rays, prim_ids_range=rg)
    assert np.array_equal(m, om), "sphere hit flags"
    assert h["t"].tobytes() == oh["t"].tobytes() and np.array_equal(h["prim_id"], oh["prim_id"]), "sphere t / prim_id"
    ok = np.isfinite(oh["u"]) & np.isfinite(oh["v"])
    assert np.array_equal(np.isnan(h["u"]), np.isnan(oh["u"])) and np.array_equal(np.isnan(h["v"]), np.isnan(oh["v"]))
    assert np.max(np.abs(h["u"][ok] - oh["u"][ok]), initial=0.0) <= 1e-6 and np.max(np.abs(h["v"][ok] - oh["v"][ok]), initial=0.0) <= 1e-6
    return rays.shape[0]


def cylinder_round():
    n = int(rng.choice([1, 2, 5, 33, 400, 3000]))
    v = (rng.normal(size=(n, 2, 3)) * rng.choice([0.5, 3.0])).astype(np.float32)
    if rng.random() < 0.3:
        v = np.round(v)  # axis-aligned, zero-length and coincident cylinders
    r = rng.uniform(0.01, rng.choice([0.1, 1.0]), (n, 2)).astype(np.float32)
    k = max(1, n // 10)
    r[:k] = rng.choice([0.0, 1e-20, 10.0], (k, 2))
    v[k: 2 * k, 1] = v[k: 2 * k, 0]
    cap = bool(rng.random() < 0.5)
    a = BVHAccel(np.float32)
    assert a.Build(n, CylinderGeometry(v, r, test_cap=cap), build_options())
    nodes, idx = a.GetTree()
    rays = random_rays(3000, v.reshape(-1, 3))
    rg = prim_range(n)
    from nanort_amd.wire import default_trace_options

    o = default_trace_options()
    o["prim_ids_range"] = rg
    h, m = a.TraverseBatch(rays, o)
    oh, om = cyl.traverse(nodes, idx, v, r, rays, prim_ids_range=rg, test_cap=cap)
    assert np.array_equal(m, om), "cylinder hit flags"
    for f in ("t", "u", "v", "prim_id", "normal"):
        assert np.array_equal(h[f], oh[f], equal_nan=True), "cylinder " + f
    return rays.shape[0]


def random_xform():
    def rot(axis, a):
        c, s = np.cos(a), np.sin(a)
        m = np.eye(4)
        i, j = [(1, 2), (0, 2), (0, 1)][axis]
        m[i, i], m[i, j], m[j, i], m[j, j] = c, s, -s, c
        return m

    kind = rng.integers(0, 6)
    sc = rng.uniform(0.2, 2.0, 3)
    if kind == 1:
        sc[:] = sc[0]
    if kind == 2:
        sc[rng.integers(0, 3)] *= -1  # mirrored
    if kind == 3:
        sc[rng.integers(0, 3)] = 1e-3  # nearly flat
    M = np.diag([sc[0], sc[1], sc[2], 1.0])
    if kind != 4:  # kind 4: axis-aligned (pure scale + translation)
        M = M @ rot(0, rng.uniform(0, 6.3)) @ rot(1, rng.uniform(0, 6.3)) @ rot(2, rng.uniform(0, 6.3))
    if kind == 5:
        M = np.eye(4)
    M[3, :3] = rng.normal(size=3) * rng.choice([0.0, 2.0, 6.0])
    return M.astype(np.float32)


MESHES = None


def scene_round():
    global MESHES
    if MESHES is None:
        sv, sf = scenes.sphere(24, 12)
        sv = sv - sv.mean(axis=0)
        pv, pf = scenes.plane(12, 8)
        pv = pv - pv.mean(axis=0)
        g = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], dtype=np.float32)
        cube_f = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4],
                           [1, 5, 7], [1, 7, 3]], dtype=np.uint32)
        MESHES = []
        quad_v = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], dtype=np.float32)  # a one-leaf tree
        quad_f = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32)
        for v, f in ((sv.astype(np.float32), sf), (pv.astype(np.float32) * 0.2, pf), (g, cube_f), (quad_v, quad_f)):
            a = BVHAccel(np.float32)
            assert a.Build(f.shape[0], TriangleMesh(v, f))
            MESHES.append((v, f, a, a.GetTree()))
    count = int(rng.choice([1, 2, 7, 30, 90, 400]))
    sc, O = Scene(), ob.SceneOracle(orc)
    centres = []
    prev = None
    for _ in range(count):
        v, f, a, tree = MESHES[rng.integers(0, len(MESHES))]
        x = random_xform() if (prev is None or rng.random() > 0.1) else prev  # sometimes an exact duplicate instance
        prev = x
        sc.AddNode(a, x)
        O.add_node(v, f, x, tree=tree)
        centres.append(x[3, :3])
    assert sc.Commit() and O.commit()
    # which path traces the scene (0: listing + trace, 2: the single-pass walk whatever the size, 1: the default rule) and the
    # thresholds of its phases never change a record
    sc.SetTunable("single_pass", int(rng.choice([0, 1, 2, 2])))
    for name in ("trav_min", "walk_trav_min"):
        sc.SetTunable(name, int(rng.choice([1, 8, 24, 64])))
    for name in ("refill_min", "walk_refill_min"):
        sc.SetTunable(name, int(rng.choice([1, 24, 56, 64])))
    sc.SetTunable("cand_min", int(rng.choice([1, 1, 16])))
    spread = max(2.0, float(np.abs(np.array(centres)).max()))
    pts = np.array(centres, dtype=np.float32) + rng.normal(size=(count, 3)).astype(np.float32) * 0.3
    rays = random_rays(4000, pts, spread=spread * 1.5, bounded=0.4)
    h, m = sc.TraverseBatch(rays)
    oh, om = O.traverse(rays)
    assert np.array_equal(m, om), "scene hit flags"
    for f in ("t", "u", "v", "prim_id", "node_id"):
        assert np.array_equal(h[f], oh[f], equal_nan=True), "scene " + f
    return rays.shape[0]


t_end = time.time() + budget
rounds = {"spheres": 0, "cylinders": 0, "scenes": 0}
total = 0
kinds = [("spheres", sphere_round), ("cylinders", cylinder_round), ("scenes", scene_round)]
i = 0
while time.time() < t_end:
    name, fn = kinds[i % 3]
    i += 1
    state = rng.bit_generator.state
    try:
        total += fn()
    except AssertionError as e:
        import pickle

        pickle.dump(state, open("gpurun_out/fuzz_prims_fail_%d_%d.pkl" % (seed, i), "wb"))
        print("MISMATCH in %s round %d (seed %d): %s — generator state saved under gpurun_out/" % (name, i, seed, e))
        sys.exit(1)
    rounds[name] += 1
print("fuzz ok: %s rounds, %d rays, seed %d" % (rounds, total, seed))
