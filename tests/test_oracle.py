"""Pin the CPU oracle (oracle/liboracle.so) — the checker everything else is judged by.

Against (a) the golden fixtures produced by running the UNMODIFIED reference
(oracle/gen_golden.py -> tests/golden/), (b) the known answers of SURVEY.md §8c,
and (c) the reference itself (oracle/_ref) when that prebuilt library is here.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from helpers import assert_hits_identical, trace_options
from nanort_amd import scenes
from nanort_amd.wire import RAY_F64, widen_rays
from oracle import bindings as ob


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def masked(nodes):
    n = nodes.copy()
    n["axis"][n["flag"] == 1] = 0
    return n


@pytest.fixture(scope="module")
def ka(golden_dir):
    return json.load(open(os.path.join(golden_dir, "known_answers.json")))


@pytest.fixture(scope="module")
def c1_ref(golden_dir):
    return np.load(os.path.join(golden_dir, "c1_ref.npz"))


def test_wire_sizes(oracle):
    assert [oracle.L.orc_sizeof(i) for i in range(6)] == [36, 72, 40, 64, 16, 32]


def test_ka1_regression_30(oracle, ka):
    """test/regression/possible-accuracy-problem-30/main.cc: fp64, one triangle; u=0.68 v=0.131201."""
    v = np.array([[1.0, 2.0, -3.0], [-1.0, 2.0, -3.0], [1.0, 2.0, 3.0]], dtype=np.float64)
    f = np.array([[0, 1, 2]], dtype=np.uint32)
    nodes, idx, st = oracle.build(v, f)
    assert len(nodes) == 1 and nodes[0]["flag"] == 1
    for label in ("normal", "tiny_dir0"):
        g = ka["KA1"][label]
        ray = np.zeros((1,), dtype=RAY_F64)
        ray["org"] = (-0.36, 7.93890843, 1.2160368)
        ray["dir"] = g["dir"]
        ray["max_t"] = 1.0e30
        h, m = oracle.traverse(nodes, idx, v, f, ray)
        assert m[0] == g["hit"] == 1
        assert h["u"][0] == g["u"] and h["v"][0] == g["v"] and h["t"][0] == g["t"]
        assert "%g" % h["u"][0] == "0.68" and "%g" % h["v"][0] == "0.131201"  # as the program prints them


def test_ka2_c1_tree_is_the_reference_tree(oracle, c1_mesh, c1_ref, ka):
    v, f = c1_mesh
    assert f.shape[0] == 980 and v.shape[0] == 531
    nodes, idx, st = oracle.build(v, f)
    assert st == ka["KA2"]["stats"] == {"max_tree_depth": 20, "num_leaf_nodes": 357, "num_branch_nodes": 356}
    assert masked(nodes).tobytes() == masked(c1_ref["nodes_f32"]).tobytes()
    assert np.array_equal(idx, c1_ref["indices_f32"])
    assert sha(masked(nodes)) == ka["KA2"]["sha256_nodes_masked"] and sha(idx) == ka["KA2"]["sha256_indices"]
    assert [float(x) for x in nodes[0]["bmin"]] == ka["KA2"]["bbox_min"]
    assert np.allclose(nodes[0]["bmin"], (-5.144927, -0.031673, -4.955276), atol=1e-6)
    assert np.allclose(nodes[0]["bmax"], (4.658843, 9.772833, 4.786486), atol=1e-6)


def test_ka2_c1_hits(oracle, c1_mesh, c1_ref, ka):
    v, f = c1_mesh
    nodes, idx, _ = oracle.build(v, f)
    rays = scenes.camera_rays(256, 256)
    h, m, cnt = oracle.traverse(nodes, idx, v, f, rays, count=True)
    assert_hits_identical(c1_ref["hits_256_f32"], c1_ref["mask_256_f32"], h, m)
    # SURVEY.md §8c KA2 (double accumulators, row-major)
    hit = m == 1
    assert int(hit.sum()) == 25472
    assert abs(h["t"][hit].astype(np.float64).sum() - 547835.902397) < 1e-5
    assert abs(h["u"][hit].astype(np.float64).sum() - 8518.003260) < 1e-5
    assert abs(h["v"][hit].astype(np.float64).sum() - 9528.415661) < 1e-5
    px = h[128 * 256 + 128]
    assert (float(px["t"]), int(px["prim_id"])) == (np.float32(24.5914974), 7)
    assert sha(h) == ka["KA2"]["wave_256"]["sha256_hits"]
    assert cnt[3] <= 512


def test_ka3_c1_512(oracle, c1_mesh, ka):
    v, f = c1_mesh
    nodes, idx, _ = oracle.build(v, f)
    h, m = oracle.traverse(nodes, idx, v, f, scenes.camera_rays(512, 512))
    assert int(m.sum()) == ka["KA3"]["num_hits"] == 101858
    assert abs(h["t"][m == 1].astype(np.float64).sum() - 2190412.321609) < 1e-4
    assert sha(h) == ka["KA3"]["sha256_hits"]


@pytest.mark.parametrize("name,opts", [
    ("cull", dict(cull=True)),
    ("skip7", dict(skip=7)),
    ("range12_500", dict(range_=(12, 500))),
])
def test_trace_options(oracle, c1_mesh, c1_ref, name, opts):
    """prim_ids_range half-open, skip_prim_id, cull_back_face — nanort.h:1055-1063, 1109-1116."""
    v, f = c1_mesh
    nodes, idx, _ = oracle.build(v, f)
    h, m = oracle.traverse(nodes, idx, v, f, scenes.camera_rays(256, 256), trace_options(**opts))
    assert_hits_identical(c1_ref["hits_256_" + name], c1_ref["mask_256_" + name], h, m)


def test_min_max_t_window(oracle, c1_mesh, c1_ref):
    """tt == t_best accepted, tt == min_t accepted, final hit iff t < max_t (strict) — nanort.h:1133-1139, 2552."""
    v, f = c1_mesh
    nodes, idx, _ = oracle.build(v, f)
    rays = scenes.camera_rays(256, 256)
    rays["min_t"] = 19.0
    rays["max_t"] = 24.5914974
    h, m = oracle.traverse(nodes, idx, v, f, rays)
    assert_hits_identical(c1_ref["hits_256_window"], c1_ref["mask_256_window"], h, m)
    assert m[128 * 256 + 128] == 0  # a hit at exactly max_t is reported as a miss


def test_fp64_instantiation(oracle, c1_mesh, c1_ref, ka):
    v, f = c1_mesh
    v64 = v.astype(np.float64)
    nodes, idx, st = oracle.build(v64, f)
    assert st == ka["KA2"]["stats_f64"]
    assert masked(nodes).tobytes() == masked(c1_ref["nodes_f64"]).tobytes()
    assert np.array_equal(idx, c1_ref["indices_f64"])
    h, m = oracle.traverse(nodes, idx, v64, f, widen_rays(scenes.camera_rays(256, 256)))
    assert_hits_identical(c1_ref["hits_256_f64"], c1_ref["mask_256_f64"], h, m)


def test_wave2_generators_and_hits(oracle, c1_mesh, c1_ref, golden_dir, ka):
    v, f = c1_mesh
    nodes, idx, _ = oracle.build(v, f)
    rays = scenes.camera_rays(256, 256)
    w2 = np.load(os.path.join(golden_dir, "c1_wave2.npz"))
    for kind in ("shadow", "bounce"):
        r2 = scenes.secondary_rays(kind, v, f, rays, c1_ref["hits_256_f32"], c1_ref["mask_256_f32"])
        assert sha(r2) == ka["KA2"]["wave2_" + kind]["sha256_rays"], "wave-2 generator is not byte-stable"
        h, m = oracle.traverse(nodes, idx, v, f, r2)
        assert_hits_identical(w2["hits_" + kind], w2["mask_" + kind], h, m)


def test_ka4_c3_plane_1m(oracle, golden_dir, ka):
    """Plane(1000,500): tree bit-equal to the reference's serial build (860 575 nodes, depth 185),
    every 13th primary ray's hit bit-equal to the reference's."""
    v, f = scenes.plane(1000, 500)
    assert sha(v) == ka["KA4"]["sha256_vertices"] and sha(f) == ka["KA4"]["sha256_faces"]
    nodes, idx, st = oracle.build(v, f)
    g = ka["KA4"]["serial"]
    assert len(nodes) == g["num_nodes"] == 860575 and st == g["stats"] and st["max_tree_depth"] == 185
    assert sha(masked(nodes)) == g["sha256_nodes_masked"] and sha(idx) == g["sha256_indices"]
    assert sha(idx) == ka["KA4"]["parallel"]["sha256_indices"]  # serial and parallel builds agree on indices_
    s = np.load(os.path.join(golden_dir, "c3_sample.npz"))
    stride = int(s["stride"])
    rays = scenes.camera_rays(1920, 1080)[::stride]
    h, m = oracle.traverse(nodes, idx, v, f, rays[::4])
    assert_hits_identical(s["hits"][::4], s["mask"][::4], h, m)
    assert ka["KA4"]["wave_1920x1080"]["num_hits"] == 2055142


def test_sphere_sample(oracle, golden_dir, ka):
    v, f = scenes.sphere()
    assert sha(v) == ka["C2_sphere"]["sha256_vertices"]
    nodes, idx, st = oracle.build(v, f)
    assert st == ka["C2_sphere"]["stats"]
    s = np.load(os.path.join(golden_dir, "sphere_sample.npz"))
    rays = scenes.camera_rays(1920, 1080)[:: int(s["stride"])]
    h, m = oracle.traverse(nodes, idx, v, f, rays)
    assert_hits_identical(s["hits"], s["mask"], h, m)


@pytest.mark.skipif(not ob.reference_available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("real", [np.float32, np.float64])
def test_oracle_equals_live_reference_on_random_soup(oracle, real):
    """Random triangle soup + random rays (incl. axis-aligned, zero components, strided vertices)."""
    rng = np.random.default_rng(1234)
    nv, nf = 900, 2500
    stride_elems = 5  # vertex stride > 3*sizeof(T): get_vertex_addr, nanort.h:467-472
    vbuf = rng.uniform(-1, 1, size=(nv, stride_elems)).astype(real)
    faces = rng.integers(0, nv, size=(nf, 3), dtype=np.uint32)
    stride = stride_elems * vbuf.dtype.itemsize
    R = ob.Reference(vbuf, faces, stride=stride)
    ok, st = R.build(parallel=False)
    rn, ri = R.tree()
    nodes, idx, ost = oracle.build(vbuf, faces, stride=stride)
    assert masked(nodes).tobytes() == masked(rn).tobytes() and np.array_equal(idx, ri)
    from nanort_amd.wire import ray_dtype
    n = 4000
    rays = np.zeros((n,), dtype=ray_dtype(real))
    rays["org"] = rng.uniform(-2, 2, size=(n, 3))
    d = rng.normal(size=(n, 3))
    d[:200, 0] = 0.0
    d[200:400, 1] = 0.0
    d[400:500, :2] = 0.0
    d[500:520] = (0.0, 0.0, -1.0)
    rays["dir"] = d
    rays["min_t"] = 0.0
    rays["max_t"] = rng.choice([1e30, 0.7, 2.0], size=n)
    for opts in (None, trace_options(cull=True), trace_options(skip=17, range_=(5, 2000))):
        rh, rm, _ = R.traverse(rays, opts)
        h, m = oracle.traverse(nodes, idx, vbuf, faces, rays, opts, stride=stride)
        assert_hits_identical(rh, rm, h, m)
