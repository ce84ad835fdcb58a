"""SURVEY §8f row 3, CPU side: the C restatement of nanosg's two-level traversal (oracle/nanosg_oracle.c)
against the unmodified reference (oracle/_ref/libnanosg_ref.so) and the golden fixture made from it."""
import os

import numpy as np
import pytest

from nanort_amd import scenes
from oracle import bindings as ob
from scene_fixture import instances


def build_oracle_scene(oracle):
    O = ob.SceneOracle(oracle)
    for v, f, x in instances():
        O.add_node(v, f, x)
    assert O.commit()
    return O


def test_restatement_matches_golden_fixture(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "scene_ref.npz"))
    O = build_oracle_scene(oracle)
    rays = scenes.camera_rays(320, 180)
    h, m = O.traverse(rays)
    assert np.array_equal(m, g["mask"]) and h.tobytes() == g["hits"].tobytes()
    assert set(np.unique(h["node_id"][m == 1]).tolist()) == {0, 1, 2, 3}  # node 4 is hidden inside node 3
    for i in range(5):
        st = O.node_state(i)
        for k in ("xbmin", "xbmax", "inv_xform", "inv_xform33", "xform"):
            assert np.array_equal(st[k], g["node%d_%s" % (i, k)])


@pytest.mark.skipif(not ob.scene_reference_available(), reason="oracle/_ref/libnanosg_ref.so not built")
def test_restatement_matches_live_reference(oracle):
    R = ob.SceneReference()
    for v, f, x in instances():
        R.add_node(v, f, x)
    assert R.commit()
    O = build_oracle_scene(oracle)
    rng = np.random.default_rng(5)
    rays = scenes.camera_rays(200, 120)
    rays["org"] += rng.uniform(-0.5, 0.5, size=(rays.shape[0], 3)).astype(np.float32)
    rh, rm = R.traverse(rays)
    oh, om = O.traverse(rays)
    assert np.array_equal(rm, om) and rh.tobytes() == oh.tobytes()
