"""CPU-only soak that pins the CHECKER: the plain-C restatement (oracle/nanort_oracle.c) against the unmodified reference
header (oracle/_ref/libnanort_ref.so) on random meshes built to provoke the edge rules — the same generators as
fuzz_parity.py (integer grids, flat sheets, duplicated and degenerate triangles; axis-parallel, zero, NaN and infinite
ray components; random trace and build options), fp32 and fp64.  Checked per round: the serial Build() — node array,
index permutation and statistics bit for bit — and Traverse() over that tree — hit flags and, for the hits, every field
bit for bit (NaNs included).  Needs the build container (the reference tree).  Usage:
python tests/checks/fuzz_oracle_vs_reference.py [seconds] [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from nanort_amd.wire import default_trace_options, ray_dtype  # noqa: E402
from oracle.bindings import Oracle, Reference  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
orc = Oracle()
t_end = time.time() + budget
rounds = rays_total = 0
while time.time() < t_end:
    real = np.float32 if rng.random() < 0.6 else np.float64
    n = int(rng.choice([1, 2, 3, 5, 17, 64, 257, 1500, 6000]))
    kind = rng.integers(0, 3)
    nv = max(3, n)
    if kind == 0:
        v = rng.integers(-4, 5, size=(nv, 3)).astype(real)
    elif kind == 1:
        v = rng.normal(size=(nv, 3)).astype(real) * 3
    else:
        v = np.column_stack([rng.uniform(-5, 5, nv), rng.uniform(-5, 5, nv), rng.integers(-2, 3, nv)]).astype(real)
    f = rng.integers(0, v.shape[0], size=(n, 3)).astype(np.uint32)
    if n > 10:
        f[: n // 10] = f[n // 10: 2 * (n // 10)]
        f[-1] = f[-1][[0, 0, 1]]
    ml, bins, md = int(rng.choice([1, 2, 4, 4, 8, 16])), int(rng.choice([2, 4, 16, 64, 64, 200])), int(rng.choice([256, 256, 12, 3]))
    R = Reference(v, f)
    ok, rst = R.build(parallel=False, min_leaf=ml, max_depth=md, bin_size=bins)
    assert ok
    rn, ri = R.tree()
    on, oi, ost = orc.build(v, f, min_leaf=ml, max_depth=md, bin_size=bins)
    tag = "round %d seed %d real %s n %d kind %d opts (%d, %d, %d)" % (rounds, seed, real.__name__, n, kind, ml, bins, md)
    rc = rn.copy()
    rc["axis"][rc["flag"] == 1] = 0  # the reference never writes a leaf's axis (nanort.h:1785-1799): stack garbage
    assert rn.shape == on.shape and rc.tobytes() == on.tobytes(), "node arrays differ: " + tag
    assert np.array_equal(ri, oi), "index permutations differ: " + tag
    for k in ("max_tree_depth", "num_leaf_nodes", "num_branch_nodes"):
        assert int(rst[k]) == int(ost[k]), "statistics differ (%s): %s" % (k, tag)
    m = 3000
    rays = np.zeros(m, dtype=ray_dtype(real))
    rays["org"] = rng.integers(-6, 7, size=(m, 3)).astype(real) if rng.random() < 0.5 else rng.normal(size=(m, 3)).astype(real) * 6
    tgt = v[rng.integers(0, v.shape[0], m)] + (rng.integers(-1, 2, size=(m, 3)) * (rng.random((m, 1)) < 0.3)).astype(real)
    d = tgt - rays["org"]
    d[: m // 8] = rng.integers(-1, 2, size=(m // 8, 3))
    d[m // 8: m // 8 + 20, 0] = np.nan
    d[m // 8 + 20: m // 8 + 40, 1] = np.inf
    rays["dir"] = d.astype(real)
    rays["min_t"] = rng.choice([0.0, 0.0, 1e-3, 0.5], m).astype(real)
    rays["max_t"] = rng.choice([1e30, 1e30, 2.0, 1.0, 0.0, -1.0], m).astype(real)
    opts = default_trace_options()
    if rng.random() < 0.5:
        lo = int(rng.integers(0, n))
        opts["prim_ids_range"] = (lo, int(rng.integers(lo, n + 3)))
    if rng.random() < 0.5:
        opts["skip_prim_id"] = int(rng.integers(0, n))
    opts["cull_back_face"] = int(rng.random() < 0.3)
    rh, rm, _ = R.traverse(rays, opts, threads=1)
    oh, om = orc.traverse(rn, ri, v, f, rays, opts)
    assert np.array_equal(rm, om), "hit flags differ: " + tag
    hit = rm != 0
    for k in ("t", "u", "v", "prim_id"):
        assert np.array_equal(rh[k][hit], oh[k][hit], equal_nan=True) and rh[k][hit].tobytes() == oh[k][hit].tobytes(), "%s differs: %s" % (k, tag)
    rounds += 1
    rays_total += m
print("oracle == reference: %d rounds, %d rays, seed %d" % (rounds, rays_total, seed))
