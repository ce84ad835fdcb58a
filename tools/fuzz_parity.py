"""Randomised parity soak: GPU traversal vs the CPU restatement on the SAME node array, bit for bit, over random
meshes built to provoke the edge rules (grid-aligned vertices -> rays through edges and vertices -> exact zeros in the
edge functions and exact t ties; degenerate and duplicated triangles; axis-parallel, zero, NaN and infinite ray
components; random trace options), fp32 and fp64, GPU-built trees and adopted oracle-built trees, and the occlusion
query's flags.  Usage: python tools/fuzz_parity.py [seconds] [seed]"""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from nanort_amd import BVHAccel, TriangleMesh
from nanort_amd.wire import ray_dtype, default_trace_options
from oracle.bindings import Oracle
from helpers import assert_hits_identical

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
orc = Oracle()
t_end = time.time() + budget
rounds = rays_total = 0
while time.time() < t_end:
    real = np.float32 if rng.random() < 0.7 else np.float64
    n = int(rng.choice([1, 2, 3, 5, 17, 64, 257, 1500, 6000]))
    kind = rng.integers(0, 3)
    if kind == 0:      # integer grid: lots of shared edges, coplanar faces, exact ties
        v = rng.integers(-4, 5, size=(max(3, n), 3)).astype(real)
    elif kind == 1:    # smooth random
        v = rng.normal(size=(max(3, n), 3)).astype(real) * 3
    else:              # flat sheets at integer heights
        v = np.column_stack([rng.uniform(-5, 5, max(3, n)), rng.uniform(-5, 5, max(3, n)), rng.integers(-2, 3, max(3, n))]).astype(real)
    f = rng.integers(0, v.shape[0], size=(n, 3)).astype(np.uint32)
    if n > 4:
        f[: n // 10] = f[n // 10: 2 * (n // 10)][: n // 10] if n // 10 else f[: 0]   # duplicated triangles
        f[-1] = f[-1][[0, 0, 1]]                                                       # a degenerate one
    m = 4000
    rays = np.zeros(m, dtype=ray_dtype(real))
    rays["org"] = rng.integers(-6, 7, size=(m, 3)).astype(real) if rng.random() < 0.5 else rng.normal(size=(m, 3)).astype(real) * 6
    tgt = v[rng.integers(0, v.shape[0], m)] + (rng.integers(-1, 2, size=(m, 3)) * (rng.random((m, 1)) < 0.3)).astype(real)
    d = tgt - rays["org"]
    d[: m // 8] = rng.integers(-1, 2, size=(m // 8, 3))          # axis-parallel and zero directions
    d[m // 8: m // 8 + 20, 0] = np.nan
    d[m // 8 + 20: m // 8 + 40, 1] = np.inf
    rays["dir"] = d.astype(real)
    rays["min_t"] = rng.choice([0.0, 0.0, 1e-3, 0.5], m).astype(real)
    rays["max_t"] = rng.choice([1e30, 1e30, 2.0, 1.0, 0.0, -1.0], m).astype(real)
    opts = default_trace_options()
    if rng.random() < 0.5:
        lo = int(rng.integers(0, n)); opts["prim_ids_range"] = (lo, int(rng.integers(lo, n + 3)))
    if rng.random() < 0.5:
        opts["skip_prim_id"] = int(rng.integers(0, n))
    opts["cull_back_face"] = int(rng.random() < 0.3)
    mesh = TriangleMesh(v, f)
    a = BVHAccel(real)
    if rng.random() < 0.5:
        assert a.Build(n, mesh)
        nodes, idx = a.GetTree()
    else:
        nodes, idx, _ = orc.build(v, f)
        a.SetMesh(mesh); a.SetTree(nodes, idx)
    h, mk = a.TraverseBatch(rays, opts)
    oh, om = orc.traverse(nodes, idx, v, f, rays, opts)
    try:
        assert_hits_identical(oh, om, h, mk)
        assert np.array_equal(a.OccludedBatch(rays, opts), om)
    except AssertionError as e:
        np.savez("gpurun_out/fuzz_fail_%d_%d.npz" % (seed, rounds), v=v, f=f, rays=rays, opts=opts, nodes=nodes, idx=idx)
        print("MISMATCH round", rounds, "real", real.__name__, "n", n, "kind", int(kind), str(e)[:300], flush=True)
        sys.exit(1)
    rounds += 1; rays_total += m
print("fuzz ok: %d rounds, %d rays, seed %d" % (rounds, rays_total, seed))
