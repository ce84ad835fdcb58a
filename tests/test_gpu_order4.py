"""The OPT-IN two-level step with its four slots entered by ENTRY DISTANCE (tunable `order4` = 1, k_traverse_wide<..., WIDTH = 4,
ORDER = 1>; the default, order4 = 0, is the reference's order and is what the rest of the GPU suite pins bit for bit).  The leaf sequence is no longer the reference's, so the bar is the contract's, not same-tree
bit identity: hit flags and t bit-equal to the restatement walking the same node array, u / v / prim_id bit-equal except
at exact-t ties, where every differing ray is re-verified (helpers.assert_hits_match: the restatement restricted to the
reported primitive must reproduce the reported record bit for bit)."""
import numpy as np
import pytest

from helpers import assert_hits_identical, assert_hits_match, is_distance_order_two_level_walk, is_reference_order_two_level_walk
from nanort_amd import BVHAccel, TriangleMesh, scenes
from nanort_amd.wire import TRACE_OPTIONS
from test_gpu_wide4 import hostile_rays

pytestmark = pytest.mark.gpu


def targs(kernel_name):
    return kernel_name.split("<")[1].rstrip(">").split(", ")


def mesh_of(name):
    if name == "c1":
        return scenes.load_c1_mesh()
    if name == "plane":
        return scenes.plane(120, 77)
    if name == "sphere":
        return scenes.sphere(64, 40)
    return scenes.plane(3, 2)


def test_the_library_default_is_the_reference_order(monkeypatch, oracle):
    """The shipped default (round 5): the reference's own slot order — records bit-identical to the restatement on the same
    node array.  A stray NRT_ORDER4 in the environment does not change that (overrides need NRT_ALLOW_ENV=1)."""
    monkeypatch.setenv("NRT_ORDER4", "1")
    monkeypatch.delenv("NRT_ALLOW_ENV", raising=False)
    v, f = scenes.load_c1_mesh()
    a = BVHAccel(np.float32)
    assert a.GetTunable("order4") == 0
    assert a.Build(f.shape[0], TriangleMesh(v, f))
    rays = scenes.camera_rays(128, 128)
    h, m = a.TraverseBatch(rays)
    assert is_reference_order_two_level_walk(a.LastKernelName()), a.LastKernelName()
    nodes, idx = a.GetTree()
    oh, om = oracle.traverse(nodes, idx, v, f, rays)
    assert_hits_identical(oh, om, h, m)
    monkeypatch.setenv("NRT_ALLOW_ENV", "1")  # the debugging aid, opted into
    b = BVHAccel(np.float32)
    assert b.GetTunable("order4") == 1


@pytest.mark.parametrize("mesh", ["c1", "plane", "sphere", "tiny"])
def test_distance_order_matches_the_oracle_up_to_exact_ties(mesh, oracle):
    v, f = mesh_of(mesh)
    a = BVHAccel(np.float32)
    a.SetTunable("order4", 0)
    assert a.Build(f.shape[0], TriangleMesh(v, f))
    nodes, idx = a.GetTree()
    rays = np.concatenate([hostile_rays(v, 40000, seed=17), scenes.camera_rays(160, 120)])
    h0, m0 = a.TraverseBatch(rays)
    assert is_reference_order_two_level_walk(a.LastKernelName()), a.LastKernelName()
    a.SetTunable("order4", 1)
    h1, m1 = a.TraverseBatch(rays)
    assert is_distance_order_two_level_walk(a.LastKernelName()), a.LastKernelName()
    oh, om = oracle.traverse(nodes, idx, v, f, rays)
    assert_hits_identical(oh, om, h0, m0)  # the default walk: the reference's leaf sequence, every field
    ties = assert_hits_match(oh, om, h1, m1, oracle, nodes, idx, v, f, rays)
    assert ties <= rays.shape[0] // 3, ties  # (vertex-aimed hostile rays tie by construction; each was verified above)
    a.SetTunable("order4", 0)
    h2, m2 = a.TraverseBatch(rays)
    assert_hits_identical(h0, m0, h2, m2)


def test_distance_order_with_rejecting_trace_options_and_occlusion(oracle):
    v, f = scenes.sphere(48, 32)
    a = BVHAccel(np.float32)
    assert a.Build(f.shape[0], TriangleMesh(v, f))
    a.SetTunable("order4", 1)
    nodes, idx = a.GetTree()
    rays = hostile_rays(v, 30000, seed=23)
    for lo, hi, skip, cull in ((100, 2000, 0xFFFFFFFF, 0), (0, 0x7FFFFFFF, 777, 1)):
        opts = np.zeros(1, dtype=TRACE_OPTIONS)
        opts["prim_ids_range"] = (lo, hi)
        opts["skip_prim_id"] = skip
        opts["cull_back_face"] = cull
        h, m = a.TraverseBatch(rays, opts)
        assert targs(a.LastKernelName())[4] == "false" and is_distance_order_two_level_walk(a.LastKernelName())
        oh, om = oracle.traverse(nodes, idx, v, f, rays, opts)
        assert_hits_match(oh, om, h, m, oracle, nodes, idx, v, f, rays, base_opts=opts[0])
    # occlusion queries keep the reference's order (any-hit: the flags do not depend on the order)
    occ = a.OccludedBatch(rays)
    _, om = oracle.traverse(nodes, idx, v, f, rays)
    assert np.array_equal(occ, om)


def test_distance_order_on_a_reference_built_deep_tree(oracle):
    v, f = scenes.plane(150, 100)
    nodes, idx, _ = oracle.build(v, f)
    a = BVHAccel(np.float32)
    a.SetMesh(TriangleMesh(v, f))
    a.SetTree(nodes, idx)
    a.SetTunable("order4", 1)
    rays = hostile_rays(v, 50000, seed=29)
    h, m = a.TraverseBatch(rays)
    assert is_distance_order_two_level_walk(a.LastKernelName())
    oh, om = oracle.traverse(nodes, idx, v, f, rays)
    assert_hits_match(oh, om, h, m, oracle, nodes, idx, v, f, rays)
