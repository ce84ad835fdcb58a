"""Tunable leaf_compact (traverse.hip "leaf items", template bit ORDER & 2; the default since round 6): the two-level walk's leaf phase hands the records of
all lanes waiting at a leaf out over the whole wave (one record per lane and trip, the owner's ray constants fetched by
ds_bpermute) and the owners accept their items' results in record order through the reference's own rule.  Same tests on the same
operands in the same sequence: every field of every record must be BIT-IDENTICAL to the default walk's and to the restated
reference's on the same node array — including exact-t ties (coplanar duplicates: the later record wins), hostile rays (NaN / inf /
zero directions), rejecting trace options, occlusion queries and several batches per launch."""
import numpy as np
import pytest

from helpers import assert_hits_identical
from nanort_amd import BVHAccel, TriangleMesh, scenes
from nanort_amd.wire import default_trace_options

pytestmark = pytest.mark.gpu


def both_walks(a, rays, opt=None):
    a.SetTunable("leaf_compact", 0)
    h0, m0 = a.TraverseBatch(rays, opt)
    k0 = a.LastKernelName()
    a.SetTunable("leaf_compact", 1)
    h1, m1 = a.TraverseBatch(rays, opt)
    k1 = a.LastKernelName()
    return (h0, m0, k0), (h1, m1, k1)


def test_leaf_items_equal_the_default_walk_and_the_oracle(oracle):
    v, f = scenes.plane(300, 200)
    a = BVHAccel(np.float32)
    assert a.Build(f.shape[0], TriangleMesh(v, f))
    rays1 = scenes.camera_rays(640, 360)
    (h0, m0, k0), (h1, m1, k1) = both_walks(a, rays1)
    assert k0.endswith(", 4, 0>") and k1.endswith(", 4, 2>"), (k0, k1)
    assert_hits_identical(h0, m0, h1, m1)
    nodes, idx = a.GetTree()
    oh, om = oracle.traverse(nodes, idx, v, f, rays1)
    assert_hits_identical(oh, om, h1, m1)
    # incoherent rays: few lanes per leaf, items of one owner straddling two trips
    bounce = scenes.secondary_rays("bounce", v, f, rays1, h0, m0)
    (b0, bm0, _), (b1, bm1, _) = both_walks(a, bounce)
    assert_hits_identical(b0, bm0, b1, bm1)
    oh, om = oracle.traverse(nodes, idx, v, f, bounce[::5])
    assert_hits_identical(oh, om, b1[::5], bm1[::5])
    # rejecting options (the non-PLAIN instantiation): prim id range, skip id, back-face culling
    o = default_trace_options()
    o["prim_ids_range"] = (1000, 90000)
    o["skip_prim_id"] = 30000
    o["cull_back_face"] = 1
    (c0, cm0, ck0), (c1, cm1, ck1) = both_walks(a, bounce, o)
    assert "false, false, 4, 2>" in ck1 and "false, false, 4, 0>" in ck0, (ck0, ck1)
    assert_hits_identical(c0, cm0, c1, cm1)
    # distance order + leaf items (ORDER = 3): the opt-in walk's own records
    a.SetTunable("order4", 1)
    (d0, dm0, dk0), (d1, dm1, dk1) = both_walks(a, bounce)
    a.SetTunable("order4", 0)
    assert dk0.endswith(", 4, 1>") and dk1.endswith(", 4, 3>")
    assert_hits_identical(d0, dm0, d1, dm1)


def test_ties_duplicates_and_hostile_rays(oracle):
    """Every triangle three times (coplanar duplicates: exact-t ties inside one leaf and across leaves — the reference keeps the
    LAST accepted record) and the hostile ray set of the fuzz tests."""
    from test_gpu_wide4 import hostile_rays

    v, f = scenes.sphere(48, 24)
    f3 = np.concatenate([f, f[::-1], f]).astype(np.uint32)
    a = BVHAccel(np.float32)
    assert a.Build(f3.shape[0], TriangleMesh(v, f3))
    rays = np.concatenate([scenes.camera_rays(200, 120), hostile_rays(v, 20000, 7)])
    (h0, m0, _), (h1, m1, k1) = both_walks(a, rays)
    assert k1.endswith(", 4, 2>")
    assert_hits_identical(h0, m0, h1, m1)
    nodes, idx = a.GetTree()
    oh, om = oracle.traverse(nodes, idx, v, f3, rays)
    assert_hits_identical(oh, om, h1, m1)
    assert int((h1["prim_id"][m1 == 1] >= 2 * f.shape[0]).sum()) > 1000  # ties went to the last copy


def test_occlusion_and_batches_and_big_leaves():
    v, f = scenes.plane(120, 80)
    a = BVHAccel(np.float32)
    assert a.Build(f.shape[0], TriangleMesh(v, f))
    rays1 = scenes.camera_rays(320, 180)
    h, m = a.TraverseBatch(rays1)
    shadow = scenes.secondary_rays("shadow", v, f, rays1, h, m)
    bounce = scenes.secondary_rays("bounce", v, f, rays1, h, m)
    a.SetTunable("leaf_compact", 0)
    want_occ = a.OccludedBatch(shadow)
    want = a.TraverseBatches([(shadow, "occlusion"), bounce, rays1[:999]])
    a.SetTunable("leaf_compact", 1)
    assert np.array_equal(a.OccludedBatch(shadow), want_occ)
    got = a.TraverseBatches([(shadow, "occlusion"), bounce, rays1[:999]])
    assert a.LastKernelName().endswith(", 4, 2>")
    assert np.array_equal(got[0][1], want[0][1])
    for k in (1, 2):
        assert_hits_identical(want[k][0], want[k][1], got[k][0], got[k][1])
    # a tree with leaves of more than four records keeps the default leaf loop (the variant needs cnt <= 4)
    b = BVHAccel(np.float32)
    from nanort_amd.wire import default_build_options

    bo = default_build_options()
    bo["min_leaf_primitives"] = 8
    assert b.Build(f.shape[0], TriangleMesh(v, f), bo)
    b.SetTunable("leaf_compact", 1)
    hb, mb = b.TraverseBatch(rays1)
    assert b.LastKernelName().endswith(", 4, 0>")
    assert_hits_identical(h, m, hb, mb)
