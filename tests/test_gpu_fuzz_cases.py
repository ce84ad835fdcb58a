"""Saved fuzz cases (hostile grid-aligned meshes: exact ties, t one ulp below a box plane, NaN distances — found by
tests/checks/fuzz_parity.py and its CPU counterparts) replayed on the GPU against the restatement on the same node array,
bit for bit; and adopted trees whose child boxes stick out of their parent's (the root box test the reference performs)."""
import glob
import os

import numpy as np
import pytest

from helpers import assert_hits_identical
from nanort_amd import BVHAccel, TriangleMesh, scenes

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("case", sorted(glob.glob(os.path.join(GOLDEN, "fuzz_case_*.npz"))), ids=os.path.basename)
@pytest.mark.parametrize("tun", [dict(), dict(refill_min=1, trav_min=1, trav_min4=1, static_pct=0, chunk=16), dict(wide4=0, static_bands=1)], ids=["default", "eager", "one_level"])
def test_saved_fuzz_cases(case, tun):
    from oracle.bindings import Oracle

    d = np.load(case)
    v, f, rays, opts, nodes, idx = d["v"], d["f"], d["rays"], d["opts"], d["nodes"], d["idx"]
    a = BVHAccel(v.dtype.type)
    for k, val in tun.items():
        a.SetTunable(k, val)
    a.SetMesh(TriangleMesh(v, f))
    a.SetTree(nodes, idx)
    h, m = a.TraverseBatch(rays, opts)
    oh, om = Oracle().traverse(nodes, idx, v, f, rays, opts)
    assert_hits_identical(oh, om, h, m)


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
