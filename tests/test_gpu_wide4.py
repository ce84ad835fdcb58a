"""Two tree levels per step (Wide4Node records, k_traverse_wide<..., WIDTH = 4>): the default walk of fp32 triangle
trees whose child boxes lie inside their parents'.  It must visit the same leaves in the same order as the one-level
walk (NRT_WIDE4=0) and as the reference loop, so every record is compared bit for bit: against the restatement on the
same node array, and against the one-level kernel on hostile inputs (degenerate rays, trace options that reject
primitives, trees with leaf children next to deep subtrees, reference-built deep trees)."""
import glob
import os

import numpy as np
import pytest

from helpers import assert_hits_identical
from nanort_amd import BVHAccel, TriangleMesh, scenes
from nanort_amd.wire import TRACE_OPTIONS

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def width(kernel_name):
    """Template arguments of k_traverse_wide: <T, STACK, STATS, KIND, PLAIN, CLOCK, WIDTH>."""
    return int(kernel_name.split("<")[1].rstrip(">").split(", ")[6])


def hostile_rays(v, n, seed):
    """Rays through the mesh's box from all around, plus the degenerate ones: zero direction components, origins on
    vertices, NaN / infinite origins, zero-length intervals."""
    rng = np.random.default_rng(seed)
    lo, hi = v.min(axis=0), v.max(axis=0)
    c, r = (lo + hi) / 2, np.linalg.norm(hi - lo)
    rays = np.zeros(n, dtype=scenes.camera_rays(2, 2).dtype)
    o = c + rng.normal(size=(n, 3)) * r
    t = lo + rng.uniform(size=(n, 3)) * (hi - lo)
    rays["org"] = o
    rays["dir"] = t - o
    rays["min_t"] = 0
    rays["max_t"] = np.finfo(np.float32).max
    k = n // 8
    rays["dir"][:k, rng.integers(0, 3, k)] = 0  # axis-parallel in one component
    rays["dir"][k:2 * k, :2] = 0  # along z only
    rays["org"][2 * k:3 * k] = v[rng.integers(0, v.shape[0], k)]  # start on a vertex
    rays["org"][3 * k:3 * k + 8] = np.nan
    rays["org"][3 * k + 8:3 * k + 16, 1] = np.inf
    rays["dir"][3 * k + 16:3 * k + 24] = 0
    rays["max_t"][4 * k:5 * k] = rng.uniform(0, 2, k) * r  # short rays
    rays["min_t"][5 * k:6 * k] = rng.uniform(0, 1, k) * r
    return rays


def both_widths(monkeypatch, make_accel, rays, opts=None):
    out = {}
    for w4 in ("0", "1"):
        monkeypatch.setenv("NRT_ALLOW_ENV", "1")  # (environment overrides are a debugging aid the process must opt into)
        monkeypatch.setenv("NRT_WIDE4", w4)  # read by nrtCreate
        a = make_accel()
        h, m = a.TraverseBatch(rays, opts) if opts is not None else a.TraverseBatch(rays)
        out[w4] = (h, m, a.LastKernelName())
    assert width(out["0"][2]) == 2 and width(out["1"][2]) == 4, (out["0"][2], out["1"][2])
    assert_hits_identical(out["0"][0], out["0"][1], out["1"][0], out["1"][1])
    return out["1"][0], out["1"][1]


@pytest.mark.parametrize("mesh", ["c1", "plane", "sphere", "tiny"])
def test_two_levels_per_step_equal_one_level_and_the_oracle(mesh, monkeypatch, oracle):
    if mesh == "c1":
        v, f = scenes.load_c1_mesh()
    elif mesh == "plane":
        v, f = scenes.plane(120, 77)
    elif mesh == "sphere":
        v, f = scenes.sphere(64, 40)
    else:  # a handful of triangles: the root's children are leaves, or one leaf and one branch
        v, f = scenes.plane(3, 2)
    keep = {}

    def make():
        a = BVHAccel(np.float32)
        assert a.Build(f.shape[0], TriangleMesh(v, f))
        keep["tree"] = a.GetTree()
        return a

    rays = hostile_rays(v, 60000, seed=7)
    h, m = both_widths(monkeypatch, make, rays)
    nodes, idx = keep["tree"]
    oh, om = oracle.traverse(nodes, idx, v, f, rays)
    assert_hits_identical(oh, om, h, m)


def test_trace_options_that_reject_primitives(monkeypatch, oracle):
    """prim_ids_range / skip_prim_id / cull_back_face: the variant with the id tests compiled in."""
    v, f = scenes.sphere(48, 32)
    keep = {}

    def make():
        a = BVHAccel(np.float32)
        assert a.Build(f.shape[0], TriangleMesh(v, f))
        keep["tree"] = a.GetTree()
        return a

    rays = hostile_rays(v, 30000, seed=11)
    for lo, hi, skip, cull in ((100, 2000, 0xFFFFFFFF, 0), (0, 0x7FFFFFFF, 777, 1)):
        opts = np.zeros(1, dtype=TRACE_OPTIONS)
        opts["prim_ids_range"] = (lo, hi)
        opts["skip_prim_id"] = skip
        opts["cull_back_face"] = cull
        h, m = both_widths(monkeypatch, make, rays, opts)
        nodes, idx = keep["tree"]
        oh, om = oracle.traverse(nodes, idx, v, f, rays, opts)
        assert_hits_identical(oh, om, h, m)


def test_reference_built_deep_tree(monkeypatch, oracle):
    """An adopted tree from the reference's builder (deeper and less balanced than the GPU builder's): same records,
    and the deeper stack (three pending entries per two levels) is sized for it."""
    v, f = scenes.plane(150, 100)
    nodes, idx, _ = oracle.build(v, f)

    def make():
        a = BVHAccel(np.float32)
        a.SetMesh(TriangleMesh(v, f))
        a.SetTree(nodes, idx)
        return a

    rays = hostile_rays(v, 50000, seed=3)
    h, m = both_widths(monkeypatch, make, rays)
    oh, om = oracle.traverse(nodes, idx, v, f, rays)
    assert_hits_identical(oh, om, h, m)


@pytest.mark.parametrize("case", sorted(glob.glob(os.path.join(GOLDEN, "fuzz_case_*f32*.npz")) +
                                        glob.glob(os.path.join(GOLDEN, "fuzz_case_r02_split.npz"))), ids=os.path.basename)
def test_saved_fuzz_cases(case, oracle):
    d = np.load(case)
    v, f, rays, opts, nodes, idx = d["v"], d["f"], d["rays"], d["opts"], d["nodes"], d["idx"]
    if v.dtype != np.float32:
        pytest.skip("fp64 trees walk one level per step")
    a = BVHAccel(np.float32)
    a.SetMesh(TriangleMesh(v, f))
    a.SetTree(nodes, idx)
    h, m = a.TraverseBatch(rays, opts)
    oh, om = oracle.traverse(nodes, idx, v, f, rays, opts)
    assert_hits_identical(oh, om, h, m)


def test_trees_with_boxes_sticking_out_walk_one_level_per_step(oracle):
    """The two-level step skips the intermediate box test, which is only implied when child boxes lie inside their
    parents': an adopted tree that breaks that is walked one level per step."""
    v, f = scenes.plane(40, 30)
    nodes, idx, _ = oracle.build(v, f)
    nodes = nodes.copy()
    branch = np.nonzero(nodes["flag"] == 0)[0]
    k = int(nodes["data"][branch[1]][0])  # a child of an inner branch: grow its box beyond the parent's
    nodes["bmax"][k] += 5.0
    a = BVHAccel(np.float32)
    a.SetMesh(TriangleMesh(v, f))
    a.SetTree(nodes, idx)
    rays = hostile_rays(v, 20000, seed=5)
    h, m = a.TraverseBatch(rays)
    assert width(a.LastKernelName()) == 2
    oh, om = oracle.traverse(nodes, idx, v, f, rays)
    assert_hits_identical(oh, om, h, m)


def test_64_bit_record_offsets_give_the_same_records(oracle):
    """Record arrays of 4 GiB and more (trees beyond ~110 M triangles) are walked by instantiations that address the records with
    64-bit offsets (template bit ORDER & 4; tools/big_mesh_probe.py runs a real one).  Forced here on an ordinary tree (tunable
    wide4_big = 2): every field of every record equals the 32-bit walk's and the restatement's, with and without leaf items, with
    rejecting trace options, for occlusion queries and several batches per launch."""
    from helpers import assert_hits_identical
    from nanort_amd.wire import default_trace_options

    v, f = scenes.plane(200, 150)
    a = BVHAccel(np.float32)
    assert a.Build(f.shape[0], TriangleMesh(v, f))
    rays1 = scenes.camera_rays(512, 288)
    h0, m0 = a.TraverseBatch(rays1)
    assert a.LastKernelName().endswith(", 4, 2>")
    bounce = scenes.secondary_rays("bounce", v, f, rays1, h0, m0)
    shadow = scenes.secondary_rays("shadow", v, f, rays1, h0, m0)
    b0, bm0 = a.TraverseBatch(bounce)
    o = default_trace_options()
    o["prim_ids_range"] = (500, 50000)
    o["skip_prim_id"] = 7000
    o["cull_back_face"] = 1
    c0, cm0 = a.TraverseBatch(bounce, o)
    occ0 = a.OccludedBatch(shadow)
    mb0 = a.TraverseBatches([(shadow, "occlusion"), bounce, rays1[:777]])
    a.SetTunable("wide4_big", 2)
    for leaf_compact, tail in ((1, ", 4, 6>"), (0, ", 4, 4>")):
        a.SetTunable("leaf_compact", leaf_compact)
        h1, m1 = a.TraverseBatch(rays1)
        assert a.LastKernelName().endswith("true, false" + tail), a.LastKernelName()
        assert_hits_identical(h0, m0, h1, m1)
        b1, bm1 = a.TraverseBatch(bounce)
        assert_hits_identical(b0, bm0, b1, bm1)
        c1, cm1 = a.TraverseBatch(bounce, o)
        assert a.LastKernelName().endswith("false, false" + tail), a.LastKernelName()
        assert_hits_identical(c0, cm0, c1, cm1)
        assert np.array_equal(a.OccludedBatch(shadow), occ0)
        mb1 = a.TraverseBatches([(shadow, "occlusion"), bounce, rays1[:777]])
        assert np.array_equal(mb0[0][1], mb1[0][1])
        for k in (1, 2):
            assert_hits_identical(mb0[k][0], mb0[k][1], mb1[k][0], mb1[k][1])
    nodes, idx = a.GetTree()
    oh, om = oracle.traverse(nodes, idx, v, f, bounce[::7])
    assert_hits_identical(oh, om, b1[::7], bm1[::7])
    # what the big arrays do not take: the opt-in distance order walks them one level per step
    a.SetTunable("order4", 1)
    a.TraverseBatch(rays1)
    assert ", 2, 0>" in a.LastKernelName(), a.LastKernelName()
