"""GPU parity, build: nrtBuild emits a valid reference-format tree (structural validator), the
reference's CPU traversal over that tree and the GPU traversal over it agree bit-for-bit, and the
hit records equal the reference's own (reference-built tree) up to verified exact ties."""
import json
import os

import numpy as np
import pytest

from bvh_check import sah_cost, validate_bvh

_validate_bvh = validate_bvh


def validate_bvh(*args, **kw):  # every tree in this file comes from the GPU builder: N1's low-side-first rule is asserted too
    kw.setdefault("low_side_first", True)
    return _validate_bvh(*args, **kw)

from helpers import assert_hits_identical, assert_hits_match, is_distance_order_two_level_walk, is_reference_order_two_level_walk
from nanort_amd import BVHAccel, TriangleMesh, scenes
from nanort_amd.wire import default_build_options, widen_rays

pytestmark = pytest.mark.gpu


def build(real, v, f, **opt):
    a = BVHAccel(real)
    o = default_build_options(real)
    for k, val in opt.items():
        o[k] = val
    assert a.Build(f.shape[0], TriangleMesh(v, f), o)
    nodes, idx = a.GetTree()
    return a, nodes, idx


def check_opt_in_distance_order(a, rays, h0, m0, oracle, nodes, idx, v, f, max_ties=None):
    """The opt-in walk (tunable order4 = 1) against the default walk's records on the SAME tree, whole batch: flags and t
    bit-equal, u / v / prim_id equal except at exact-t ties, each of which is re-verified against the restatement restricted to
    the reported primitive (helpers.assert_hits_match).  Leaves the context on the default walk."""
    a.SetTunable("order4", 1)
    try:
        h1, m1 = a.TraverseBatch(rays)
        assert is_distance_order_two_level_walk(a.LastKernelName()), a.LastKernelName()
    finally:
        a.SetTunable("order4", 0)
    return assert_hits_match(h0, m0, h1, m1, oracle, nodes, idx, v, f, rays, max_ties=max_ties)


def test_c1_build_valid_and_parity(oracle, c1_mesh, golden_dir):
    v, f = c1_mesh
    a, nodes, idx = build(np.float32, v, f)
    m = validate_bvh(nodes, idx, v, f, stats=a.GetStatistics())
    g = np.load(os.path.join(golden_dir, "c1_ref.npz"))
    assert m["sah_cost"] <= sah_cost(g["nodes_f32"]), "GPU tree must not be worse than the reference's (SAH)"
    rays = scenes.camera_rays(256, 256)
    h, mk = a.TraverseBatch(rays)
    # (ii) CPU restatement of the reference traversal over the GPU-built array: identical bits
    oh, om = oracle.traverse(nodes, idx, v, f, rays)
    assert_hits_identical(oh, om, h, mk)
    # (iii) vs the reference's own records (its own tree): equal up to verified ties
    ties = assert_hits_match(g["hits_256_f32"], g["mask_256_f32"], h, mk, oracle, g["nodes_f32"], g["indices_f32"], v, f, rays)
    assert ties <= 64
    bmin, bmax = a.BoundingBox()
    assert np.array_equal(bmin, g["nodes_f32"][0]["bmin"]) and np.array_equal(bmax, g["nodes_f32"][0]["bmax"])
    assert a.LastBuildMs() > 0 and a.GetStatistics()["build_secs"] > 0


@pytest.mark.parametrize("opt", [
    dict(min_leaf_primitives=1), dict(min_leaf_primitives=8), dict(min_leaf_primitives=16, bin_size=8),
    dict(bin_size=2), dict(bin_size=1024), dict(max_tree_depth=6), dict(max_tree_depth=0), dict(max_tree_depth=1),
    dict(min_leaf_primitives=300), dict(min_leaf_primitives=1000, bin_size=16), dict(min_leaf_primitives=5000),
])
def test_build_options_are_honoured(oracle, opt):
    v, f = scenes.plane(60, 40)  # 4800 triangles: top phase + subtree phase
    a, nodes, idx = build(np.float32, v, f, **opt)
    validate_bvh(nodes, idx, v, f, min_leaf=opt.get("min_leaf_primitives", 4),
                 max_depth=opt.get("max_tree_depth", 256), stats=a.GetStatistics())
    rays = scenes.camera_rays(160, 90)
    h, mk = a.TraverseBatch(rays)
    oh, om = oracle.traverse(nodes, idx, v, f, rays)
    assert_hits_identical(oh, om, h, mk)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 9, 255, 256, 257, 513])
def test_tiny_and_boundary_primitive_counts(oracle, n):
    v, f = scenes.plane(40, 20)
    f = f[:n]
    a, nodes, idx = build(np.float32, v, f)
    validate_bvh(nodes, idx, v, f, stats=a.GetStatistics())
    rays = scenes.camera_rays(64, 36)
    h, mk = a.TraverseBatch(rays)
    oh, om = oracle.traverse(nodes, idx, v, f, rays)
    assert_hits_identical(oh, om, h, mk)


def test_empty_build_returns_false_like_the_reference(c1_mesh):
    v, f = c1_mesh
    a = BVHAccel(np.float32)
    assert a.Build(0, TriangleMesh(v, f[:0])) is False  # nanort.h:1907-1909
    assert not a.IsValid()
    bmin, bmax = a.BoundingBox()
    assert bmin[0] == np.finfo(np.float32).max and bmax[0] == -np.finfo(np.float32).max  # nanort.h:792-796


def test_coincident_centroids_use_the_median_fallback(oracle):
    """1000 copies of one triangle + a few others: no centroid separates them (nanort.h:1845-1850)."""
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [5, 5, 5], [6, 5, 5], [5, 6, 5]], dtype=np.float32)
    f = np.array([[0, 1, 2]] * 1000 + [[3, 4, 5]] * 3, dtype=np.uint32)
    a, nodes, idx = build(np.float32, v, f)
    validate_bvh(nodes, idx, v, f, stats=a.GetStatistics())
    rays = scenes.camera_rays(32, 32)
    rays["org"] = (0.25, 0.25, 3.0)
    rays["dir"] = (0.0, 0.0, -1.0)
    h, mk = a.TraverseBatch(rays)
    oh, om = oracle.traverse(nodes, idx, v, f, rays)
    assert_hits_identical(oh, om, h, mk)
    assert mk.all() and (h["t"] == 3.0).all()


def test_build_is_deterministic(c1_mesh):
    v, f = scenes.sphere(96, 48)
    _, n1, i1 = build(np.float32, v, f)
    _, n2, i2 = build(np.float32, v, f)
    assert n1.tobytes() == n2.tobytes() and np.array_equal(i1, i2)


def test_top_array_overflow_is_retried_with_the_same_result(monkeypatch):
    """Lopsided splits can outgrow the top-phase node array; the builder then starts over with a larger one.  The test hook
    gives the first attempt a 16-node array, so the retry path runs: same tree as an ordinary build, for both precisions."""
    v, f = scenes.sphere(96, 48)
    for real in (np.float32, np.float64):
        vv = v.astype(real)
        _, n1, i1 = build(real, vv, f)
        monkeypatch.setenv("NRT_ALLOW_ENV", "1")
        monkeypatch.setenv("NRT_BUILD_TINY_TOP", "1")
        a, n2, i2 = build(real, vv, f)
        monkeypatch.delenv("NRT_BUILD_TINY_TOP")
        assert n1.tobytes() == n2.tobytes() and np.array_equal(i1, i2)
        validate_bvh(n2, i2, vv, f, stats=a.GetStatistics())


def test_fp64_build(oracle, c1_mesh):
    v, f = c1_mesh
    v64 = v.astype(np.float64)
    a, nodes, idx = build(np.float64, v64, f)
    validate_bvh(nodes, idx, v64, f, stats=a.GetStatistics())
    rays = widen_rays(scenes.camera_rays(256, 256))
    h, mk = a.TraverseBatch(rays)
    oh, om = oracle.traverse(nodes, idx, v64, f, rays)
    assert_hits_identical(oh, om, h, mk)


def test_strided_vertices(oracle):
    rng = np.random.default_rng(3)
    vb = rng.uniform(-1, 1, size=(400, 7)).astype(np.float32)
    f = rng.integers(0, 400, size=(1500, 3), dtype=np.uint32)
    a = BVHAccel(np.float32)
    assert a.Build(1500, TriangleMesh(vb, f, 28))
    nodes, idx = a.GetTree()
    tight = np.ascontiguousarray(vb[:, :3])
    validate_bvh(nodes, idx, tight, f, stats=a.GetStatistics())


# ---- full-size configs of BASELINE.json ---------------------------------------------------------

def test_c2_sphere_full_frame_vs_reference_sample(oracle, golden_dir):
    v, f = scenes.sphere()
    a, nodes, idx = build(np.float32, v, f)
    validate_bvh(nodes, idx, v, f, stats=a.GetStatistics())
    rays = scenes.camera_rays(1920, 1080)
    h, mk = a.TraverseBatch(rays)
    s = np.load(os.path.join(golden_dir, "sphere_sample.npz"))
    st = int(s["stride"])
    onodes, oidx, _ = oracle.build(v, f)
    assert_hits_match(s["hits"], s["mask"], h[::st], mk[::st], oracle, onodes, oidx, v, f, rays[::st], max_ties=200)
    # the default walk is the reference's leaf sequence: every field identical to the restatement on the GPU-built tree
    assert is_reference_order_two_level_walk(a.LastKernelName()), a.LastKernelName()
    sub = slice(None, None, 5)
    oh, om = oracle.traverse(nodes, idx, v, f, rays[sub])
    assert_hits_identical(oh, om, h[sub], mk[sub])
    ties = check_opt_in_distance_order(a, rays, h, mk, oracle, nodes, idx, v, f)
    print("C2: %d exact-t ties renamed by the opt-in distance order on the full frame" % ties)


def test_c3_plane_1m_full_frame(oracle, golden_dir):
    """Config C3 at BASELINE.json's full size: 1 000 000 triangles, 1920x1080 primaries + bounce."""
    v, f = scenes.plane(1000, 500)
    a, nodes, idx = build(np.float32, v, f)
    m = validate_bvh(nodes, idx, v, f, stats=a.GetStatistics())
    assert m["max_depth"] < 64
    rays = scenes.camera_rays(1920, 1080)
    h, mk = a.TraverseBatch(rays)
    ka = json.load(open(os.path.join(golden_dir, "known_answers.json")))["KA4"]["wave_1920x1080"]
    # size-independent checksums against the reference's full frame: hit count and sum of t
    # (t is tie-independent, so the sums must agree exactly in double accumulation)
    assert int(mk.sum()) == ka["num_hits"] == 2055142
    assert float(h["t"][mk == 1].astype(np.float64).sum()) == ka["sum_t"]
    s = np.load(os.path.join(golden_dir, "c3_sample.npz"))
    st = int(s["stride"])
    onodes, oidx, _ = oracle.build(v, f)
    ties = assert_hits_match(s["hits"], s["mask"], h[::st], mk[::st], oracle, onodes, oidx, v, f, rays[::st], max_ties=2000)
    # the GPU traversal and the CPU restatement agree bit-for-bit on the GPU-built tree (subsample)
    assert is_reference_order_two_level_walk(a.LastKernelName()), a.LastKernelName()  # the shipped default: the reference's slot order
    sub = rays[::7]
    oh, om = oracle.traverse(nodes, idx, v, f, sub)
    assert_hits_identical(oh, om, h[::7], mk[::7])
    ties4 = check_opt_in_distance_order(a, rays, h, mk, oracle, nodes, idx, v, f)
    print("C3: %d exact-t ties renamed by the opt-in distance order on the full frame" % ties4)
    # properties: permuting the rays permutes the hits; tracing twice is idempotent
    perm = np.random.default_rng(0).permutation(rays.shape[0])
    hp, mp = a.TraverseBatch(rays[perm])
    assert hp.tobytes() == h[perm].tobytes() and np.array_equal(mp, mk[perm])
    # bounce wave from these hits: round trip through the wave-2 generator, CPU restatement on a subsample
    r2 = scenes.secondary_rays("bounce", v, f, rays, h, mk)
    h2, m2 = a.TraverseBatch(r2)
    oh2, om2 = oracle.traverse(nodes, idx, v, f, r2[::101])
    assert_hits_identical(oh2, om2, h2[::101], m2[::101])
    print("C3: %d ties on the %d-ray sample" % (ties, s["hits"].shape[0]))


def test_scale_by_two_doubles_t_exactly():
    """Linearity: scaling the mesh and the ray origins by 2 (a power of two) scales every t by exactly 2
    and leaves u, v, prim_id and the hit mask unchanged."""
    v, f = scenes.sphere(128, 64)
    rays = scenes.camera_rays(320, 180)
    a, _, _ = build(np.float32, v, f)
    h, m = a.TraverseBatch(rays)
    r2 = rays.copy()
    r2["org"] *= 2
    b, _, _ = build(np.float32, v * 2, f)
    h2, m2 = b.TraverseBatch(r2)
    assert np.array_equal(m, m2)
    hit = m == 1
    assert np.array_equal(h["t"][hit] * 2, h2["t"][hit]) and np.array_equal(h["prim_id"], h2["prim_id"])
    assert np.array_equal(h["u"][hit], h2["u"][hit]) and np.array_equal(h["v"][hit], h2["v"][hit])


def test_smoke_entry_point():
    import __graft_entry__

    __graft_entry__.smoke()


def test_morton_prepass_gives_a_valid_equal_quality_tree(oracle, monkeypatch):
    """NRT_MORTON=1: 30-bit Morton codes + stable wavefront radix sort of the primitive records before the
    top-down build (the north-star's pre-pass).  The tree must stay valid and of the same SAH quality; hit
    records do not depend on it."""
    v, f = scenes.sphere(160, 80)
    a0, n0, i0 = build(np.float32, v, f)
    monkeypatch.setenv("NRT_ALLOW_ENV", "1")  # (environment overrides are a debugging aid the process must opt into)
    monkeypatch.setenv("NRT_MORTON", "1")
    a1, n1, i1 = build(np.float32, v, f)
    m0 = validate_bvh(n0, i0, v, f, stats=a0.GetStatistics())
    m1 = validate_bvh(n1, i1, v, f, stats=a1.GetStatistics())
    assert abs(m1["sah_cost"] - m0["sah_cost"]) <= 1e-3 * m0["sah_cost"]
    rays = scenes.camera_rays(320, 180)
    h0, k0 = a0.TraverseBatch(rays)
    h1, k1 = a1.TraverseBatch(rays)
    onodes, oidx, _ = oracle.build(v, f)
    oh, om = oracle.traverse(onodes, oidx, v, f, rays)
    assert_hits_match(oh, om, h1, k1, oracle, onodes, oidx, v, f, rays, max_ties=100)
    assert np.array_equal(h0["t"], h1["t"]) and np.array_equal(k0, k1)
    # sorted-order build is deterministic too
    a2, n2, i2 = build(np.float32, v, f)
    assert n1.tobytes() == n2.tobytes() and np.array_equal(i1, i2)


def test_c5_fp64_plane_1m_full_frame(oracle, golden_dir):
    """Config C5: fp64 build + traverse, 1M triangles, 1920x1080 (checksums of the reference's fp64 frame)."""
    v, f = scenes.plane(1000, 500)
    v64 = v.astype(np.float64)
    a, nodes, idx = build(np.float64, v64, f)
    validate_bvh(nodes, idx, v64, f, stats=a.GetStatistics())
    rays = widen_rays(scenes.camera_rays(1920, 1080))
    h, mk = a.TraverseBatch(rays)
    ka = json.load(open(os.path.join(golden_dir, "known_answers.json")))["KA4"]["wave_1920x1080_f64"]
    assert int(mk.sum()) == ka["num_hits"]
    assert float(h["t"][mk == 1].sum()) == ka["sum_t"]
    sub = rays[::211]
    oh, om = oracle.traverse(nodes, idx, v64, f, sub)
    assert_hits_identical(oh, om, h[::211], mk[::211])


def test_c4_10m_triangles_4k_tile(oracle):
    """Config C4 sizing on one GPU: Plane(2500,2000) = 10 000 000 triangles; one 4096x512 tile of the 4096x4096
    frame (what each of 8 GPUs traces).  Valid tree, CPU restatement agrees on a subsample, permutation property."""
    v, f = scenes.plane(2500, 2000)
    assert f.shape[0] == 10_000_000
    a, nodes, idx = build(np.float32, v, f)
    m = validate_bvh(nodes, idx, v, f, stats=a.GetStatistics())
    assert m["max_depth"] < 64
    rays = scenes.camera_rays(4096, 4096, 1792, 2304)  # the central tile
    h, mk = a.TraverseBatch(rays)
    assert int(mk.sum()) > rays.shape[0] // 2
    assert is_reference_order_two_level_walk(a.LastKernelName()), a.LastKernelName()
    sub = slice(None, None, 53)
    oh, om = oracle.traverse(nodes, idx, v, f, rays[sub])
    assert_hits_identical(oh, om, h[sub], mk[sub])
    ties4 = check_opt_in_distance_order(a, rays, h, mk, oracle, nodes, idx, v, f)
    print("C4 tile: %d exact-t ties renamed by the opt-in distance order" % ties4)
    perm = np.random.default_rng(1).permutation(rays.shape[0])[:500000]
    hp, mp = a.TraverseBatch(rays[perm])
    assert hp.tobytes() == h[perm].tobytes() and np.array_equal(mp, mk[perm])
    print("C4: build %.2f ms, %d nodes, depth %d" % (a.LastBuildMs(), m["num_nodes"], m["max_depth"]))


def test_c4_full_frame_from_eight_row_interleaved_tiles(oracle):
    """Config C4 end to end on ONE GPU: the 4096x4096 frame over the 10 M-triangle plane traced as the eight row-interleaved
    tiles eight ranks would own (bench.py / nanort_amd.dist: rank r traces rows r, r + 8, ...), the 16-byte records
    reassembled by dist.assemble_image exactly as the root does with its gathered buffer — and that frame equals the frame
    traced as ONE 16.8 M-ray batch in every byte; a strided sample of it equals the restated reference on the GPU-built tree."""
    from nanort_amd import dist as nd
    from nanort_amd.wire import HIT_F32

    W = H = 4096
    world = 8
    v, f = scenes.plane(2500, 2000)
    a, nodes, idx = build(np.float32, v, f)
    tiles_h, tiles_m = [], []
    for r in range(world):
        rays = scenes.camera_rays_rows(W, H, r, world, nd.rows_per_rank(H, r, world))
        h, m = a.TraverseBatch(rays)
        tiles_h.append(h)
        tiles_m.append(m)
    gathered = np.concatenate(tiles_h)  # rank-major, as dist.gather_hit_records delivers it
    img = nd.assemble_image(gathered.view(np.uint8), W, H, world, HIT_F32)
    mask = np.empty((H, W), dtype=np.uint8)
    for r in range(world):
        mask[r::world] = tiles_m[r].reshape(-1, W)
    del tiles_h, gathered
    full = scenes.camera_rays(W, H)
    hf, mf = a.TraverseBatch(full)
    assert img.tobytes() == hf.tobytes() and np.array_equal(mask.reshape(-1), mf)
    sub = slice(None, None, 4099)
    oh, om = oracle.traverse(nodes, idx, v, f, full[sub])
    assert_hits_identical(oh, om, hf[sub], mf[sub])
    assert int(mf.sum()) > W * H // 2


def test_context_reuse_across_mesh_sizes(oracle):
    """One context, rebuilt over meshes of different sizes (grow-only workspace, no stale state): each tree is valid
    and traces like a fresh context's."""
    a = BVHAccel(np.float32)
    rays = scenes.camera_rays(160, 90)
    for nx, ny in ((20, 10), (300, 200), (8, 8), (120, 60), (300, 200)):
        v, f = scenes.plane(nx, ny)
        assert a.Build(f.shape[0], TriangleMesh(v, f))
        nodes, idx = a.GetTree()
        validate_bvh(nodes, idx, v, f, stats=a.GetStatistics())
        h, m = a.TraverseBatch(rays)
        oh, om = oracle.traverse(nodes, idx, v, f, rays)
        assert_hits_identical(oh, om, h, m)


def test_two_contexts_from_two_threads(oracle):
    """Distinct contexts may be driven from distinct host threads concurrently (ctypes drops the GIL)."""
    import threading

    meshes = [scenes.plane(150, 100), scenes.sphere(96, 48)]
    rays = scenes.camera_rays(320, 180)
    results = [None, None]
    errors = []

    def worker(k):
        try:
            v, f = meshes[k]
            a = BVHAccel(np.float32)
            out = []
            for _ in range(6):
                assert a.Build(f.shape[0], TriangleMesh(v, f))
                h, m = a.TraverseBatch(rays)
                out.append((h.tobytes(), m.tobytes()))
            nodes, idx = a.GetTree()
            results[k] = (out, nodes, idx)
        except Exception as e:  # pragma: no cover
            errors.append(e)

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    for k in range(2):
        out, nodes, idx = results[k]
        assert all(o == out[0] for o in out), "results changed between iterations"
        v, f = meshes[k]
        oh, om = oracle.traverse(nodes, idx, v, f, rays)
        assert out[0][0] == oh.tobytes() and out[0][1] == om.tobytes()


@pytest.mark.parametrize("kind", ["inf", "nan"])
@pytest.mark.parametrize("real", [np.float32, np.float64])
def test_non_finite_vertices_build_a_walkable_tree(oracle, real, kind):
    """+-Inf / NaN vertex components (the reference's Build() does not reject them): the GPU builder must stay inside its
    arrays and emit a structurally sound tree — a permutation of the primitives, children in range, leaf ranges tiling
    the index array — and the traversal over it must equal the restatement's on the same tree.  (A non-finite root box
    culls nearly every ray in the reference's slab test; the restatement over its own tree reports 1 hit of 20 000 too.)"""
    from helpers import assert_hits_identical
    from nanort_amd.wire import ray_dtype

    rng = np.random.default_rng(17)
    nv, nf = 4000, 9000
    v = rng.normal(size=(nv, 3)).astype(real) * 3
    if kind == "nan":
        v[rng.integers(0, nv, 25), rng.integers(0, 3, 25)] = np.nan
    v[rng.integers(0, nv, 25), rng.integers(0, 3, 25)] = np.inf
    v[rng.integers(0, nv, 25), rng.integers(0, 3, 25)] = -np.inf
    f = rng.integers(0, nv, size=(nf, 3)).astype(np.uint32)
    a = BVHAccel(real)
    assert a.Build(nf, TriangleMesh(v, f))
    nodes, idx = a.GetTree()
    assert sorted(idx.tolist()) == list(range(nf))
    leaf = nodes["flag"] == 1
    assert (nodes["data"][~leaf] < nodes.shape[0]).all() and (nodes["data"][~leaf] > 0).all()
    first, count = nodes["data"][leaf][:, 1].astype(np.int64), nodes["data"][leaf][:, 0].astype(np.int64)
    order = np.argsort(first)
    assert first[order][0] == 0 and np.array_equal(first[order][1:], (first + count)[order][:-1]) and (first + count).max() == nf
    n = 20000
    rays = np.zeros(n, dtype=ray_dtype(real))
    rays["org"] = rng.normal(size=(n, 3)) * 6
    d = rng.normal(size=(n, 3)) - rays["org"] * 0.3
    rays["dir"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays["max_t"] = 1e30
    h, m = a.TraverseBatch(rays)
    oh, om = oracle.traverse(nodes, idx, v, f, rays)
    assert np.array_equal(m, om)
    for k in ("t", "u", "v", "prim_id"):
        assert np.array_equal(h[k], oh[k], equal_nan=True), k


def _two_subtree_kernels(real, v, f, **opt):
    """The same mesh through the row form of the subtree phase (this process, the product library) and the one-node-per-step
    form — which only the profiling build of the library carries: a child process builds with it and dumps the tree."""
    import json
    import subprocess
    import sys
    import tempfile

    a = BVHAccel(real)
    o = default_build_options(real)
    for k, val in opt.items():
        o[k] = val
    assert a.Build(f.shape[0], TriangleMesh(v, f), o)
    n1, i1 = a.GetTree()
    s1 = a.GetStatistics()
    a.close()
    here = os.path.dirname(os.path.abspath(__file__))
    with tempfile.TemporaryDirectory() as tmp:
        np.savez(os.path.join(tmp, "in.npz"), v=v, f=f, real=np.dtype(real).name, tunables=json.dumps({"subtree_rows": 0}),
                 options=json.dumps({k: (int(x) if float(x).is_integer() else float(x)) for k, x in opt.items()}))
        r = subprocess.run([sys.executable, os.path.join(here, "checks", "build_dump.py"), os.path.join(tmp, "in.npz"), os.path.join(tmp, "out.npz")],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-1500:]
        d = np.load(os.path.join(tmp, "out.npz"))
        n0b, i0, s0 = d["nodes"].tobytes(), d["idx"], json.loads(str(d["stats"]))
    assert n1.tobytes() == n0b, "the two subtree kernels must emit the same node array"
    assert np.array_equal(i1, i0)
    for key in ("max_tree_depth", "num_leaf_nodes", "num_branch_nodes"):
        assert s1[key] == s0[key], key
    return n1, i1, s1


def _sliver_chain(n, ratio, real):
    """Tiny triangles whose centroids sit at ratio**i along x: every SAH split peels a handful of primitives off one end, a
    chain of lopsided splits far deeper than log2(n) inside ONE subtree task — the case the subtree kernels cut off with
    object-median splits once too many high-side children are pending (kSubStackSafe)."""
    x = ratio ** np.arange(n, dtype=np.float64)
    v = np.empty((3 * n, 3), dtype=np.float64)
    s = 1e-3 * x
    v[0::3] = np.stack([x, np.zeros(n), np.zeros(n)], 1)
    v[1::3] = np.stack([x + s, s, np.zeros(n)], 1)
    v[2::3] = np.stack([x, s, s], 1)
    f = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    return v.astype(real), f


@pytest.mark.parametrize("real,n,ratio,min_leaf", [
    (np.float64, 256, 0.3, 1), (np.float64, 256, 0.5, 1), (np.float64, 200, 0.2, 2), (np.float32, 100, 0.45, 1),
    (np.float64, 5000, 0.97, 1),
])
def test_the_two_subtree_kernels_agree_on_chains_of_lopsided_splits(real, n, ratio, min_leaf):
    v, f = _sliver_chain(n, ratio, real)
    nodes, idx, st = _two_subtree_kernels(real, v, f, min_leaf_primitives=min_leaf)
    validate_bvh(nodes, idx, v, f, min_leaf=min_leaf, stats=st)
    if n <= 256 and ratio <= 0.3:
        assert st["max_tree_depth"] > 36, "the chain must be deep enough to reach the pending-children guard"


@pytest.mark.parametrize("seed", range(6))
def test_the_two_subtree_kernels_agree_on_random_meshes_and_options(seed):
    rng = np.random.default_rng(1000 + seed)
    real = np.float32 if seed % 2 == 0 else np.float64
    n = int(rng.choice([3, 17, 64, 255, 256, 257, 700, 5000, 40000]))
    kind = seed % 3
    if kind == 0:    # soup
        c = rng.uniform(-1, 1, (n, 1, 3))
        tri = c + rng.normal(0, 0.03, (n, 3, 3))
    elif kind == 1:  # clusters of coincident centroids (median splits) next to spread-out ones
        c = np.repeat(rng.uniform(-1, 1, (max(1, n // 7), 1, 3)), 7, axis=0)[:n]
        if len(c) < n:
            c = np.concatenate([c, rng.uniform(-1, 1, (n - len(c), 1, 3))])
        d = rng.normal(0, 0.02, (n, 3, 3))
        tri = c + d - d.mean(axis=1, keepdims=True)
    else:            # flat along one axis, wildly different scales along another
        c = rng.uniform(-1, 1, (n, 1, 3)) * np.array([1.0, 1e-6, 0.0])
        tri = c + rng.normal(0, 0.01, (n, 3, 3)) * np.array([1.0, 1e-6, 1.0])
    v = tri.reshape(-1, 3).astype(real)
    f = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    opt = dict(min_leaf_primitives=int(rng.choice([1, 2, 4, 7, 16])), bin_size=int(rng.choice([2, 5, 16, 64])),
               max_tree_depth=int(rng.choice([3, 9, 20, 256])))
    nodes, idx, st = _two_subtree_kernels(real, v, f, **opt)
    validate_bvh(nodes, idx, v, f, min_leaf=opt["min_leaf_primitives"], max_depth=opt["max_tree_depth"], stats=st)
