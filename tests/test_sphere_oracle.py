"""SURVEY §8f row 4, CPU side: the C restatement of the sphere ("particle") primitive traced through Traverse
(oracle/sphere_oracle.c) against the golden fixture made from the unmodified reference example, and against the
live reference where it was built (oracle/_ref/libsphere_ref.so)."""
import os

import numpy as np
import pytest

from nanort_amd import scenes
from oracle import bindings as ob
import sphere_fixture


def test_restatement_matches_golden_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "spheres_ref.npz"))
    c, r = sphere_fixture.scene()
    rays = sphere_fixture.rays()
    O = ob.SphereOracle()
    h, m = O.traverse(g["nodes"], g["indices"], c, r, rays)
    assert np.array_equal(m, g["mask"]) and h.tobytes() == g["hits"].tobytes()
    h2, m2 = O.traverse(g["nodes"], g["indices"], c, r, rays, (1000, 3000))
    assert np.array_equal(m2, g["mask_range"]) and h2.tobytes() == g["hits_range"].tobytes()
    assert 0 < int(m2.sum()) < int(m.sum())
    hit = m == 1
    assert np.all((h["u"][hit] >= 0) & (h["u"][hit] <= 1) & (h["v"][hit] >= 0) & (h["v"][hit] <= 1))
    assert np.all(h["prim_id"][~hit] == 0xFFFFFFFF)


@pytest.mark.skipif(not ob.have_sphere_reference(), reason="oracle/_ref/libsphere_ref.so not built")
@pytest.mark.parametrize("n", [1, 2, 5, 64, 3000])
def test_restatement_matches_live_reference(n):
    R = ob.SphereReference()
    c, r = R.generate(n)
    mine_c, mine_r = scenes.random_spheres(n)
    assert c.tobytes() == mine_c.tobytes() and r.tobytes() == mine_r.tobytes()  # the harness's generator == the example's
    nodes, idx, st = R.build(c, r)
    assert st["num_leaf_nodes"] + st["num_branch_nodes"] == nodes.shape[0]
    rays = sphere_fixture.rays()
    h, m = R.traverse(rays)
    oh, om = ob.SphereOracle().traverse(nodes, idx, c, r, rays)
    assert np.array_equal(m, om) and h.tobytes() == oh.tobytes()


@pytest.mark.skipif(not ob.have_sphere_reference(), reason="oracle/_ref/libsphere_ref.so not built")
def test_degenerate_spheres_and_hostile_rays_match_live_reference():
    c, r = sphere_fixture.degenerate_spheres()
    rays = sphere_fixture.hostile_rays()
    R = ob.SphereReference()
    nodes, idx, _ = R.build(c, r)
    h, m = R.traverse(rays)
    oh, om = ob.SphereOracle().traverse(nodes, idx, c, r, rays)
    assert np.array_equal(m, om) and h.tobytes() == oh.tobytes()
