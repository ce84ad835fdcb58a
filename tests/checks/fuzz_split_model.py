"""CPU soak of the work-splitting fold rule (oracle/split_model_body.inc, the model of the traversal kernel's drain-time
splitting) against the restated reference loop, on the hostile generator of fuzz_parity.py: integer-grid meshes (exact
ties, t one ulp below a box's entry distance), flat sheets, duplicated and degenerate triangles; rays through vertices
and edges, axis-parallel / zero / NaN / infinite components, bounded intervals; random trace options; fp32 and fp64;
trees from the restated reference builder with random options.
Usage: python tests/checks/fuzz_split_model.py [seconds] [seed]     (needs no GPU)"""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from nanort_amd.wire import ray_dtype, default_trace_options
from oracle.bindings import Oracle


def hostile_case(rng, orc):
    """One random hostile mesh + ray batch + trace options + a tree from the restated reference builder."""
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
        f[: n // 10] = f[n // 10: 2 * (n // 10)][: n // 10] if n // 10 else f[: 0]
        f[-1] = f[-1][[0, 0, 1]]
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
    rays["max_t"] = rng.choice([1e30, 1e30, 2.0, 1.0, 0.0, -1.0, np.inf], m).astype(real)
    opts = default_trace_options()
    if rng.random() < 0.5:
        lo = int(rng.integers(0, n)); opts["prim_ids_range"] = (lo, int(rng.integers(lo, n + 3)))
    if rng.random() < 0.5:
        opts["skip_prim_id"] = int(rng.integers(0, n))
    opts["cull_back_face"] = int(rng.random() < 0.3)
    nodes, idx, _ = orc.build(v, f, min_leaf=int(rng.choice([1, 2, 4, 8, 16])), bin_size=int(rng.choice([2, 4, 16, 64])),
                              max_depth=int(rng.choice([256, 256, 12, 3])))
    return v, f, rays, opts, nodes, idx


def one_round(rng, orc, stats):
    v, f, rays, opts, nodes, idx = hostile_case(rng, orc)
    m = rays.shape[0]
    oh, om = orc.traverse(nodes, idx, v, f, rays, opts)
    pm = int(rng.choice([50, 300, 1000]))
    sh, sm, fl, sp = orc.traverse_split_model(nodes, idx, v, f, rays, opts, split_permille=pm, seed=int(rng.integers(1, 1 << 30)))
    same = np.array_equal(om, sm) and all(oh[k].tobytes() == sh[k].tobytes() for k in ("t", "u", "v", "prim_id"))
    # what the consistency flag is for: the same fold WITHOUT it
    uh, um, _, _ = orc.traverse_split_model(nodes, idx, v, f, rays, opts, split_permille=pm, seed=1, check_helpers=False)
    tb = lambda h: np.ascontiguousarray(h["t"]).view(np.uint8).reshape(m, -1)
    unflagged_diff = int((um != om).sum() + ((um == om) & ((tb(uh) != tb(oh)).any(axis=1) | (uh["prim_id"] != oh["prim_id"]))).sum())
    stats["rays"] += m; stats["split_rays"] += int((sp > 0).sum()); stats["segments"] += int(sp.sum())
    stats["flagged"] += int(fl.sum()); stats["would_differ_without_flag"] += unflagged_diff
    return same, (v, f, rays, opts, nodes, idx)


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    orc = Oracle()
    stats = {"rays": 0, "split_rays": 0, "segments": 0, "flagged": 0, "would_differ_without_flag": 0}
    t_end = time.time() + budget
    rounds = 0
    while time.time() < t_end:
        ok, case = one_round(rng, orc, stats)
        if not ok:
            v, f, rays, opts, nodes, idx = case
            np.savez("/tmp/split_model_fail_%d_%d.npz" % (seed, rounds), v=v, f=f, rays=rays, opts=opts, nodes=nodes, idx=idx)
            print("MISMATCH round", rounds, stats, flush=True)
            sys.exit(1)
        rounds += 1
    print("split model ok: %d rounds, seed %d, %s" % (rounds, seed, stats))
