"""The hostile generator shared by the randomised soaks (fuzz_parity.py, fuzz_wide4_model.py): integer-grid meshes (exact
ties, t one ulp below a box's entry distance), flat sheets, duplicated and degenerate triangles; rays through vertices and
edges, axis-parallel / zero / NaN / infinite components, bounded intervals; random trace options; fp32 and fp64; trees from
the restated reference builder with random options."""
import numpy as np

from nanort_amd.wire import default_trace_options, ray_dtype


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
