"""Drain-time work splitting of the traversal kernel (k_traverse_wide<..., SPLIT>, enabled per context with
NRT_SPLIT=1): exact by construction + consistency flag (tests/test_split_model.py soaks the rule on the CPU); here the
kernel itself against the restatement on the same node array, bit for bit — on the saved fuzz cases (hostile
grid-aligned meshes: the flag fires and rays are re-run), on C1, and on a batch that ends in a long drain.
Also: adopted trees whose child boxes stick out of their parent's (the root box test the reference performs)."""
import glob
import os

import numpy as np
import pytest

from helpers import assert_hits_identical
from nanort_amd import BVHAccel, TriangleMesh, scenes

pytestmark = pytest.mark.gpu


def splitting(kernel_name):
    """Template arguments of k_traverse_wide: <T, STACK, STATS, KIND, PLAIN, SPLIT, CLOCK>."""
    return kernel_name.split("<")[1].rstrip(">").split(", ")[5] == "true"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture()
def split_on(monkeypatch):
    monkeypatch.setenv("NRT_SPLIT", "1")  # read by nrtCreate
    monkeypatch.setenv("NRT_DRAIN_STEPS", "1")  # a hand-out round on every trip
    monkeypatch.setenv("NRT_SPLIT_BUSY", "64")  # hand out at every opportunity


@pytest.mark.parametrize("case", sorted(glob.glob(os.path.join(GOLDEN, "fuzz_case_*.npz"))), ids=os.path.basename)
def test_saved_fuzz_cases_with_splitting(case, split_on):
    from oracle.bindings import Oracle

    d = np.load(case)
    v, f, rays, opts, nodes, idx = d["v"], d["f"], d["rays"], d["opts"], d["nodes"], d["idx"]
    a = BVHAccel(v.dtype.type)
    a.SetMesh(TriangleMesh(v, f))
    a.SetTree(nodes, idx)
    h, m = a.TraverseBatch(rays, opts)
    if v.dtype == np.float32:
        assert splitting(a.LastKernelName()), a.LastKernelName()  # the splitting variant ran
    oh, om = Oracle().traverse(nodes, idx, v, f, rays, opts)
    assert_hits_identical(oh, om, h, m)


def test_c1_with_splitting_equals_the_oracle(split_on):
    from oracle.bindings import Oracle

    v, f = scenes.load_c1_mesh()
    a = BVHAccel(np.float32)
    assert a.Build(f.shape[0], TriangleMesh(v, f))
    nodes, idx = a.GetTree()
    for w, h_ in ((64, 64), (300, 217)):  # small batches: every wave is out of rays at once and splits from the start
        rays = scenes.camera_rays(w, h_)
        h, m = a.TraverseBatch(rays)
        oh, om = Oracle().traverse(nodes, idx, v, f, rays)
        assert_hits_identical(oh, om, h, m)
    assert splitting(a.LastKernelName())


def test_splitting_gives_the_same_records_as_the_production_kernel_on_c3_bounce_rays(monkeypatch):
    v, f = scenes.plane(400, 250)
    mesh = TriangleMesh(v, f)
    rays1 = scenes.camera_rays(960, 540)
    base = BVHAccel(np.float32)
    assert base.Build(mesh.num_faces, mesh)
    h1, m1 = base.TraverseBatch(rays1)
    rays2 = scenes.secondary_rays("bounce", v, f, rays1, h1, m1)
    hb, mb = base.TraverseBatch(rays2)
    assert not splitting(base.LastKernelName())
    monkeypatch.setenv("NRT_SPLIT", "1")
    s = BVHAccel(np.float32)
    assert s.Build(mesh.num_faces, mesh)
    hs, ms = s.TraverseBatch(rays2)
    assert splitting(s.LastKernelName())
    assert np.array_equal(mb, ms) and hb.tobytes() == hs.tobytes()


def test_adopted_tree_whose_children_stick_out_of_the_root_box():
    """nrtSetTree accepts any well-formed node array.  The production kernel normally skips the root's own box test
    (a ray that misses node 0's box misses both children's when they lie inside it); for a tree that breaks that
    containment it must test node 0 first, as BVHAccel::Traverse does (nanort.h:2526-2533)."""
    from oracle.bindings import Oracle

    orc = Oracle()
    v, f = scenes.load_c1_mesh()
    nodes, idx, _ = orc.build(v, f)
    nodes = nodes.copy()
    mid = 0.5 * (nodes[0]["bmin"] + nodes[0]["bmax"])
    nodes[0]["bmax"][0] = mid[0]  # hand-edited: the root now covers only half of the scene in x
    rays = scenes.camera_rays(128, 128)
    oh, om = orc.traverse(nodes, idx, v, f, rays)
    full_h, full_m = orc.traverse(orc.build(v, f)[0], idx, v, f, rays)
    assert int(full_m.sum()) > int(om.sum()) > 0  # the edit really cuts rays off
    a = BVHAccel(np.float32)
    a.SetMesh(TriangleMesh(v, f))
    a.SetTree(nodes, idx)
    h, m = a.TraverseBatch(rays)
    assert_hits_identical(oh, om, h, m)
