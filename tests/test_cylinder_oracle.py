"""SURVEY §8f row 4 (cylinders), CPU side: the C restatement of the cylinder primitive traced through Traverse
(oracle/cylinder_oracle.c) against the golden fixture made from the unmodified reference example, and against the
live reference where it was built (oracle/_ref/libcylinder_ref.so)."""
import os

import numpy as np
import pytest

from nanort_amd import scenes
from oracle import bindings as ob
import sphere_fixture


def test_restatement_matches_golden_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "cylinders_ref.npz"))
    v, r = scenes.random_cylinders(sphere_fixture.N_CYLINDERS)
    rays = sphere_fixture.rays()
    O = ob.CylinderOracle()
    for key, kw in (("", {}), ("_nocap", {"test_cap": False}), ("_range", {"prim_ids_range": (500, 2500)})):
        h, m = O.traverse(g["nodes"], g["indices"], v, r, rays, **kw)
        assert np.array_equal(m, g["mask" + key]) and h.tobytes() == g["hits" + key].tobytes(), key
    h, m = g["hits"], g["mask"]
    hit = m == 1
    cap = hit & ((h["v"] == 0) | (h["v"] == 1)) & (h["u"] > 0)
    assert cap.sum() > 50 and (hit & ~cap).sum() > 5000  # both the caps and the side are exercised
    assert np.allclose(np.linalg.norm(h["normal"][hit], axis=1), 1.0, atol=1e-5)


@pytest.mark.skipif(not ob.have_cylinder_reference(), reason="oracle/_ref/libcylinder_ref.so not built")
@pytest.mark.parametrize("n", [1, 2, 5, 64, 3000])
def test_restatement_matches_live_reference(n):
    R = ob.CylinderReference()
    v, r = R.generate(n)
    mv, mr = scenes.random_cylinders(n)
    assert v.tobytes() == mv.tobytes() and r.tobytes() == mr.tobytes()  # the harness's generator == the example's
    nodes, idx, st = R.build(v, r)
    rays = sphere_fixture.rays()
    for cap in (True, False):
        h, m = R.traverse(rays, test_cap=cap)
        oh, om = ob.CylinderOracle().traverse(nodes, idx, v, r, rays, test_cap=cap)
        assert np.array_equal(m, om) and h.tobytes() == oh.tobytes()


@pytest.mark.skipif(not ob.have_cylinder_reference(), reason="oracle/_ref/libcylinder_ref.so not built")
def test_degenerate_cylinders_and_hostile_rays_match_live_reference():
    v, r = sphere_fixture.degenerate_cylinders()
    rays = sphere_fixture.hostile_rays()
    R = ob.CylinderReference()
    nodes, idx, _ = R.build(v, r)
    for cap in (True, False):
        h, m = R.traverse(rays, test_cap=cap)
        oh, om = ob.CylinderOracle().traverse(nodes, idx, v, r, rays, test_cap=cap)
        assert np.array_equal(m, om)
        for f in ("t", "u", "v", "prim_id", "normal"):
            assert np.array_equal(h[f], oh[f], equal_nan=True), f
