"""CPU-only soak that pins the CHECKERS of the widened rows: the C restatements of the sphere and cylinder intersectors
(oracle/sphere_oracle.c, oracle/cylinder_oracle.c) and of NanoSG's two-level traversal (oracle/nanosg_oracle.c) against the
unmodified reference code (oracle/_ref/lib{sphere,cylinder,nanosg}_ref.so) on random inputs: degenerate radii, lattice
positions, origins inside the primitives, zero / NaN / infinite ray components, bounded intervals, random node
transforms.  Every field of every record, bit for bit (NaNs included).  Needs the build container.  Usage:
python tests/checks/fuzz_prim_oracles_vs_reference.py [seconds] [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from nanort_amd import scenes  # noqa: E402
from nanort_amd.wire import RAY_F32  # noqa: E402
from oracle import bindings as ob  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
orc = ob.Oracle()
sph_o, cyl_o = ob.SphereOracle(), ob.CylinderOracle()
sph_r, cyl_r = ob.SphereReference(), ob.CylinderReference()


def random_rays(m, targets, spread=6.0, bounded=0.5):
    r = np.zeros(m, dtype=RAY_F32)
    r["org"] = (rng.normal(size=(m, 3)) * spread).astype(np.float32)
    inside = rng.random(m) < 0.15
    r["org"][inside] = targets[rng.integers(0, targets.shape[0], int(inside.sum()))]
    tgt = targets[rng.integers(0, targets.shape[0], m)] + rng.normal(size=(m, 3)).astype(np.float32) * 0.05
    d = tgt - r["org"]
    nrm = np.linalg.norm(d, axis=1, keepdims=True)
    unit = rng.random((m, 1)) < 0.7
    d = np.where(unit & (nrm > 0), d / np.where(nrm > 0, nrm, 1), d)
    d[: m // 16] = rng.integers(-1, 2, size=(m // 16, 3))
    d[m // 16: m // 16 + 8, 0] = np.nan
    d[m // 16 + 8: m // 16 + 16, 2] = np.inf
    r["dir"] = d.astype(np.float32)
    r["min_t"] = rng.choice([0.0, 0.0, 0.0, 1e-3, 0.7], m).astype(np.float32)
    far = rng.choice([1e30, 3.4028234663852886e38, 5.0, 1.5, 0.0, -1.0], m)
    r["max_t"] = np.where(rng.random(m) < bounded, far, 1e30).astype(np.float32)
    return r


def prim_range(n):
    if rng.random() < 0.6:
        return (0, 0x7FFFFFFF)
    lo = int(rng.integers(0, n))
    return (lo, int(rng.integers(lo, n + 2)))


def same(a, b, fields, what):
    for f in fields:
        assert np.array_equal(a[f], b[f], equal_nan=True), "%s: %s differs" % (what, f)


def sphere_round():
    n = int(rng.choice([1, 2, 5, 33, 400, 2000]))
    c = (rng.normal(size=(n, 3)) * rng.choice([0.5, 3.0])).astype(np.float32)
    if rng.random() < 0.3:
        c = np.round(c)
    r = rng.uniform(0.01, rng.choice([0.1, 1.0, 4.0]), n).astype(np.float32)
    k = max(1, n // 10)
    r[:k] = rng.choice([0.0, -0.2, 1e-20, 50.0], k)
    nodes, idx, _ = sph_r.build(c, r)
    rays = random_rays(2000, c)
    rg = prim_range(n)
    rh, rm = sph_r.traverse(rays, prim_ids_range=rg)
    oh, om = sph_o.traverse(nodes, idx, c, r, rays, prim_ids_range=rg)
    assert np.array_equal(rm, om), "spheres: hit flags differ"
    same(rh, oh, ("t", "u", "v", "prim_id"), "spheres")
    return rays.shape[0]


def cylinder_round():
    n = int(rng.choice([1, 2, 5, 33, 400, 2000]))
    v = (rng.normal(size=(n, 2, 3)) * rng.choice([0.5, 3.0])).astype(np.float32)
    if rng.random() < 0.3:
        v = np.round(v)
    r = rng.uniform(0.01, rng.choice([0.1, 1.0]), (n, 2)).astype(np.float32)
    k = max(1, n // 10)
    r[:k] = rng.choice([0.0, 1e-20, 10.0], (k, 2))
    v[k: 2 * k, 1] = v[k: 2 * k, 0]
    cap = bool(rng.random() < 0.5)
    nodes, idx, _ = cyl_r.build(v, r)
    rays = random_rays(2000, v.reshape(-1, 3))
    rg = prim_range(n)
    rh, rm = cyl_r.traverse(rays, prim_ids_range=rg, test_cap=cap)
    oh, om = cyl_o.traverse(nodes, idx, v, r, rays, prim_ids_range=rg, test_cap=cap)
    assert np.array_equal(rm, om), "cylinders: hit flags differ"
    same(rh, oh, ("t", "u", "v", "prim_id", "normal"), "cylinders")
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
        sc[rng.integers(0, 3)] *= -1
    if kind == 3:
        sc[rng.integers(0, 3)] = 1e-3
    M = np.diag([sc[0], sc[1], sc[2], 1.0])
    if kind != 4:
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
        pv, pf = scenes.plane(12, 8)
        g = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], dtype=np.float32)
        cube_f = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4],
                           [1, 5, 7], [1, 7, 3]], dtype=np.uint32)
        MESHES = [((sv - sv.mean(axis=0)).astype(np.float32), sf), (((pv - pv.mean(axis=0)) * 0.2).astype(np.float32), pf), (g, cube_f)]
        MESHES = [(v, f, orc.build(v, f)[:2]) for v, f in MESHES]
    count = int(rng.choice([1, 2, 7, 30, 90]))
    R, O = ob.SceneReference(), ob.SceneOracle(orc)
    centres, prev = [], None
    for _ in range(count):
        v, f, tree = MESHES[rng.integers(0, len(MESHES))]
        x = random_xform() if (prev is None or rng.random() > 0.1) else prev
        prev = x
        R.add_node(v, f, x)
        O.add_node(v, f, x, tree=tree)
        centres.append(x[3, :3])
    assert R.commit() and O.commit()
    for i in range(count):
        a, b = R.node_state(i), O.node_state(i)
        for k in ("xbmin", "xbmax", "xform", "inv_xform", "inv_xform33"):
            assert np.array_equal(a[k], b[k], equal_nan=True), "scene node %d: %s differs" % (i, k)
    spread = max(2.0, float(np.abs(np.array(centres)).max()))
    pts = np.array(centres, dtype=np.float32) + rng.normal(size=(count, 3)).astype(np.float32) * 0.3
    rays = random_rays(1500, pts, spread=spread * 1.5, bounded=0.4)
    rh, rm = R.traverse(rays)
    oh, om = O.traverse(rays)
    assert np.array_equal(rm, om), "scenes: hit flags differ"
    same(rh, oh, ("t", "u", "v", "prim_id"), "scenes")
    # node_id: equal, except that of several COINCIDENT instances (same world box, same hit record) either may be named —
    # nodes entered at exactly the same distance are visited in node order here, in the order its std::priority_queue
    # happens to pop them in the reference (nanort.h:2608-2692)
    for i in np.nonzero(rh["node_id"] != oh["node_id"])[0]:
        a, b = R.node_state(int(rh["node_id"][i])), O.node_state(int(oh["node_id"][i]))
        assert np.array_equal(a["xbmin"], b["xbmin"]) and np.array_equal(a["xbmax"], b["xbmax"]) and np.array_equal(a["xform"], b["xform"]), \
            "scenes: node_id differs between nodes that are not coincident"
    return rays.shape[0]


t_end = time.time() + budget
rounds = {"spheres": 0, "cylinders": 0, "scenes": 0}
total = 0
kinds = [("spheres", sphere_round), ("cylinders", cylinder_round), ("scenes", scene_round)]
i = 0
while time.time() < t_end:
    name, fn = kinds[i % 3]
    i += 1
    try:
        total += fn()
    except AssertionError as e:
        print("MISMATCH in %s round %d (seed %d): %s" % (name, i, seed, e))
        sys.exit(1)
    rounds[name] += 1
print("restatements == references: %s rounds, %d rays, seed %d" % (rounds, total, seed))
