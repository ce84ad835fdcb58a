"""SURVEY §8f row 3 on the GPU: two-level (instanced) traversal through the C ABI (nrtScene*) vs the C restatement
of nanosg's Scene::Traverse and the golden fixture produced by the unmodified reference."""
import os

import numpy as np
import pytest

from nanort_amd import BVHAccel, Scene, TriangleMesh, scenes
from oracle import bindings as ob
from scene_fixture import instances

pytestmark = pytest.mark.gpu


def fields_equal(a, b, keys):
    return all(np.ascontiguousarray(a[k]).tobytes() == np.ascontiguousarray(b[k]).tobytes() for k in keys)


def test_same_local_trees_bit_identical(oracle, golden_dir):
    """Every node adopts the reference-built local tree (nrtSetTree): the GPU result must equal the reference's
    own two-level result in every field (t, u, v, prim_id, node_id, mask)."""
    g = np.load(os.path.join(golden_dir, "scene_ref.npz"))
    sc = Scene()
    accels = []
    for v, f, x in instances():
        a = BVHAccel(np.float32)
        a.SetMesh(TriangleMesh(v, f))
        nodes, idx, _ = oracle.build(v, f)
        a.SetTree(nodes, idx)
        accels.append(a)
        sc.AddNode(a, x)
    assert sc.Commit()
    rays = scenes.camera_rays(320, 180)
    h, m = sc.TraverseBatch(rays)
    assert np.array_equal(m, g["mask"])
    assert fields_equal(h, g["hits"], ("t", "u", "v", "prim_id", "node_id"))


def test_gpu_built_local_trees_match_up_to_ties(oracle):
    """Local trees built on the GPU: hit mask, world t and node_id equal the restatement's; prim_id/u/v may differ
    only where two triangles of a node are hit at exactly the same local t."""
    sc = Scene()
    O = ob.SceneOracle(oracle)
    keep = []
    for v, f, x in instances(sphere_res=(96, 48), plane_res=(200, 100)):
        a = BVHAccel(np.float32)
        assert a.Build(f.shape[0], TriangleMesh(v, f))
        keep.append(a)
        sc.AddNode(a, x)
        O.add_node(v, f, x)
    assert sc.Commit() and O.commit()
    rng = np.random.default_rng(11)
    rays = scenes.camera_rays(640, 360)
    rays["org"] += rng.uniform(-0.3, 0.3, size=(rays.shape[0], 3)).astype(np.float32)
    h, m = sc.TraverseBatch(rays)
    oh, om = O.traverse(rays)
    assert np.array_equal(m, om)
    assert fields_equal(h, oh, ("t", "node_id"))
    same = h["prim_id"] == oh["prim_id"]
    assert same.mean() > 0.999
    assert np.array_equal(h["u"][same], oh["u"][same]) and np.array_equal(h["v"][same], oh["v"][same])


def test_scene_error_paths_and_empty_scene(c1_mesh):
    from nanort_amd import NrtError

    sc = Scene()
    assert sc.Commit() is False  # the reference's Commit() returns false on an empty scene (nanosg.h:702-706)
    v, f = c1_mesh
    a = BVHAccel(np.float32)
    a.SetMesh(TriangleMesh(v, f))
    with pytest.raises(NrtError):
        sc.AddNode(a, np.eye(4))  # no tree yet
    a.Build(f.shape[0], TriangleMesh(v, f))
    sc.AddNode(a, np.eye(4))
    with pytest.raises(NrtError):
        sc.TraverseBatch(scenes.camera_rays(8, 8))  # not committed
    assert sc.Commit()
    h, m = sc.TraverseBatch(scenes.camera_rays(64, 64))
    # identity instance: the hit mask equals the single-level traversal's, and node_id is 0 on every hit
    h1, m1 = a.TraverseBatch(scenes.camera_rays(64, 64))
    assert np.array_equal(m, m1) and (h["node_id"][m == 1] == 0).all()
