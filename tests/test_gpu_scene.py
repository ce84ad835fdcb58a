"""SURVEY §8f row 3 on the GPU: two-level (instanced) traversal through the C ABI (nrtScene*) vs the C restatement
of nanosg's Scene::Traverse and the golden fixture produced by the unmodified reference."""
import os

import numpy as np
import pytest

from nanort_amd import BVHAccel, Scene, TriangleMesh, scenes
from oracle import bindings as ob
from scene_fixture import instances, xform

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
    # a handful of nodes is listed by a scan by default; the single-pass walk over a five-leaf top-level tree gives the same records
    sc.SetTunable("single_pass", 2)
    h2, m2 = sc.TraverseBatch(rays)
    assert sc.LastRedone() < len(rays) // 2
    assert np.array_equal(m2, g["mask"])
    assert fields_equal(h2, g["hits"], ("t", "u", "v", "prim_id", "node_id"))


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
    # a mesh context rebuilt after Commit: the scene's cached view of its tree is stale -> refused until committed again
    v2, f2 = scenes.sphere(32, 16)
    assert a.Build(f2.shape[0], TriangleMesh(v2, f2))
    with pytest.raises(NrtError, match="commit the scene again"):
        sc.TraverseBatch(scenes.camera_rays(8, 8))
    assert sc.Commit()
    h2, m2 = sc.TraverseBatch(scenes.camera_rays(64, 64))
    h3, m3 = a.TraverseBatch(scenes.camera_rays(64, 64))
    assert np.array_equal(m2, m3)


def test_node_state_matches_the_reference(golden_dir, oracle):
    """nrtSceneNodeState_f32: the matrices Node::Update derives, bit for bit as the unmodified nanosg computed them."""
    g = np.load(os.path.join(golden_dir, "scene_ref.npz"))
    sc = Scene()
    keep = []
    for v, f, x in instances():
        a = BVHAccel(np.float32)
        a.SetMesh(TriangleMesh(v, f))
        nodes, idx, _ = oracle.build(v, f)
        a.SetTree(nodes, idx)
        keep.append(a)
        sc.AddNode(a, x)
    assert sc.Commit()
    for i in range(5):
        st = sc.NodeState(i)
        for k in ("xform", "inv_xform", "inv_xform33"):
            assert np.array_equal(st[k], g["node%d_%s" % (i, k)]), (i, k)
        assert np.array_equal(st["inv_transpose_xform33"], g["node%d_inv_xform33" % i].T)


def test_nanosg_batch_tracer_addon(tmp_path):
    """include/nanosg_hip.h: BatchTracer over a NanoSG-shaped scene == the C ABI's compact records, plus the rest of the
    reference's Intersection record (P on the ray at distance t, normals through inv_transpose_xform33)."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc, libdir = os.path.join(root, "include"), os.path.join(root, "nanort_amd", "lib")
    exe = tmp_path / "nanosg_batch_check"
    r = subprocess.run(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-DNANORT_USE_HIP_BACKEND", "-I", inc,
                        os.path.join(root, "tests", "cpp", "nanosg_batch_check.cc"), "-o", str(exe), "-L", libdir, "-lnanort_hip",
                        "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    inst = instances()
    scene_path, rays_path, out_path = (str(tmp_path / n) for n in ("scene.bin", "rays.bin", "out.bin"))
    with open(scene_path, "wb") as fp:
        fp.write(np.array([len(inst)], dtype=np.uint32).tobytes())
        for v, f, x in inst:
            fp.write(np.array([v.shape[0], f.shape[0]], dtype=np.uint32).tobytes())
            fp.write(np.ascontiguousarray(v, dtype=np.float32).tobytes())
            fp.write(np.ascontiguousarray(f, dtype=np.uint32).tobytes())
            fp.write(np.ascontiguousarray(x, dtype=np.float32).tobytes())
    rays = scenes.camera_rays(320, 180)
    with open(rays_path, "wb") as fp:
        fp.write(np.array([rays.shape[0]], dtype=np.uint64).tobytes())
        fp.write(rays.tobytes())
    r = subprocess.run([str(exe), scene_path, rays_path, out_path], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    n = rays.shape[0]
    rec = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("prim_id", "<u4"), ("node_id", "<u4"), ("P", "<f4", 3),
                    ("Ns", "<f4", 3), ("Ng", "<f4", 3)])
    raw = open(out_path, "rb").read()
    out = np.frombuffer(raw, dtype=rec, count=n)
    mask = np.frombuffer(raw, dtype=np.uint8, count=n, offset=n * rec.itemsize)
    # the same scene through the Python mirror of the C ABI (GPU-built local trees: the builder is deterministic)
    sc = Scene()
    keep = []
    for v, f, x in inst:
        a = BVHAccel(np.float32)
        assert a.Build(f.shape[0], TriangleMesh(v, f))
        keep.append(a)
        sc.AddNode(a, x)
    assert sc.Commit()
    h, m = sc.TraverseBatch(rays)
    assert np.array_equal(mask, m) and int(m.sum()) > 1000
    hit = m == 1
    assert fields_equal(out[hit], h[hit], ("t", "u", "v", "prim_id", "node_id"))
    d = rays["dir"][hit] / np.linalg.norm(rays["dir"][hit], axis=1, keepdims=True)
    assert np.allclose(out["P"][hit], rays["org"][hit] + d * out["t"][hit, None], rtol=0, atol=2e-4)
    assert np.array_equal(out["Ns"][hit], out["Ng"][hit])
    # normals: the flat triangle normal through inv_transpose_xform33 of its node; check direction for node 0 (identity)
    n0 = hit & (out["node_id"] == 0)
    assert np.allclose(np.linalg.norm(out["Ng"][n0], axis=1), 1.0, atol=1e-5)
    assert set(np.unique(out["node_id"][hit]).tolist()) == {0, 1, 2, 3}


def test_device_resident_entry_point_matches_the_host_one():
    """nrtSceneTraverseBatchDevice_f32: rays and records in HBM, same bytes as the host entry point; the hit-flag array is
    optional."""
    import torch

    from nanort_amd.wire import SCENE_HIT_F32

    sc = Scene()
    keep = []
    for v, f, x in instances():
        a = BVHAccel(np.float32)
        assert a.Build(f.shape[0], TriangleMesh(v, f))
        keep.append(a)
        sc.AddNode(a, x)
    assert sc.Commit()
    rays = scenes.camera_rays(320, 180)
    h, m = sc.TraverseBatch(rays)
    d_rays = torch.from_numpy(rays.view(np.uint8)).cuda()
    d_hits = torch.zeros(rays.shape[0] * SCENE_HIT_F32.itemsize, dtype=torch.uint8, device="cuda")
    d_mask = torch.zeros(rays.shape[0], dtype=torch.uint8, device="cuda")
    sc.TraverseBatchDevice(d_rays, d_hits, d_mask)
    assert d_hits.cpu().numpy().tobytes() == h.tobytes() and np.array_equal(d_mask.cpu().numpy(), m)
    d_hits.zero_()
    sc.TraverseBatchDevice(d_rays, d_hits)
    assert d_hits.cpu().numpy().tobytes() == h.tobytes()
    bmin, bmax = sc.GetBoundingBox()
    assert np.all(bmin < bmax)


def test_ten_thousand_instances_through_the_top_level_bvh(oracle):
    """10 000 transformed copies of two small meshes: the listing walks the top-level BVH (built on the GPU over the
    nodes' world boxes), the trace is ONE launch.  Equal to the restatement (a scan over all 10 000 node boxes per ray, as
    nanosg's own top-level traversal is equivalent to) in every field."""
    from scene_fixture import xform

    rng = np.random.default_rng(5)
    sv, sf = scenes.sphere(16, 8)
    sv = sv - np.array([0, 5, 0], dtype=np.float32)
    pv, pf = scenes.plane(6, 4)
    pv = (pv - pv.mean(axis=0)).astype(np.float32)
    meshes = []
    for v, f in ((sv, sf), (pv, pf)):
        a = BVHAccel(np.float32)
        assert a.Build(f.shape[0], TriangleMesh(v, f))
        nodes, idx = a.GetTree()
        meshes.append((v, f, a, (nodes, idx)))
    sc = Scene()
    O = ob.SceneOracle(oracle)
    N = 10000
    for k in range(N):
        v, f, a, tree = meshes[k % 2]
        s = rng.uniform(0.004, 0.02, 3) if k % 2 == 0 else rng.uniform(0.008, 0.03, 3)
        x = xform(tuple(s), rng.uniform(0, 6.28), rng.uniform(0, 6.28), tuple(rng.uniform(-8, 8, 3) + np.array([0, 5, 0])))
        sc.AddNode(a, x)
        O.add_node(v, f, x, tree=tree)  # the same local node arrays: every field must agree
    assert sc.Commit() and O.commit()
    rays = scenes.camera_rays(256, 192)
    rays["org"] += rng.uniform(-0.5, 0.5, size=(rays.shape[0], 3)).astype(np.float32)
    h, m = sc.TraverseBatch(rays)
    oh, om = O.traverse(rays)
    assert 0.05 < om.mean() < 0.95
    assert np.array_equal(m, om)
    assert fields_equal(h, oh, ("t", "u", "v", "prim_id", "node_id"))


@pytest.mark.parametrize("prune", [False, True], ids=["plain_walk", "pruning_walk"])
def test_rays_that_enter_more_than_64_boxes_keep_the_64_nearest(oracle, monkeypatch, prune):
    """kMaxIntersections (nanosg.h:782): of the node boxes a ray enters only the 64 nearest by (entry distance, id) are
    considered.  200 small spheres strung along a line, rays shot down the line (entering all of them, from either end and
    from the middle), plus a camera wave: every field equals the restatement's."""
    from scene_fixture import xform

    if prune:  # the front-to-back walk that skips what lies beyond a full list (by default only on scenes of >= 32768 instances)
        monkeypatch.setenv("NRT_ALLOW_ENV", "1")  # (environment overrides are a debugging aid the process must opt into)
        monkeypatch.setenv("NRT_SCENE_PRUNE_MIN", "1")
    sv, sf = scenes.sphere(16, 8)
    sv = (sv - np.array([0, 5, 0], dtype=np.float32)).astype(np.float32)
    a = BVHAccel(np.float32)
    assert a.Build(sf.shape[0], TriangleMesh(sv, sf))
    tree = a.GetTree()
    sc = Scene()
    O = ob.SceneOracle(oracle)
    rng = np.random.default_rng(3)
    N = 200
    for k in range(N):
        x = xform((0.02, 0.02, 0.02), rng.uniform(0, 6.28), rng.uniform(0, 6.28), (0.25 * k - 25.0, 5.0 + 0.01 * rng.uniform(-1, 1), 0.0))
        sc.AddNode(a, x)
        O.add_node(sv, sf, x, tree=tree)
    assert sc.Commit() and O.commit()
    from nanort_amd.wire import RAY_F32

    m = 3000
    rays = np.zeros(m, dtype=RAY_F32)
    rays["org"][:, 0] = rng.choice([-30.0, 30.0, 0.1], m)
    rays["org"][:, 1] = 5.0 + rng.uniform(-0.2, 0.2, m)
    rays["org"][:, 2] = rng.uniform(-0.2, 0.2, m)
    tgt = np.stack([rng.uniform(-25, 25, m), 5.0 + rng.uniform(-0.25, 0.25, m), rng.uniform(-0.25, 0.25, m)], axis=1)
    d = tgt - rays["org"]
    rays["dir"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays["max_t"] = 1.0e30
    # exactly axis-parallel rays (zero direction components: the listing then prunes nothing), some with -0.0
    rays["dir"][:60] = 0.0
    rays["dir"][:60, 0] = np.where(rays["org"][:60, 0] > 0, -1.0, 1.0)
    rays["dir"][:20, 1] = -0.0
    rays["org"][:60, 1] = 5.0
    rays["org"][:60, 2] = 0.0
    rays = np.concatenate([rays, scenes.camera_rays(160, 90)])
    h, mk = sc.TraverseBatch(rays)
    oh, om = O.traverse(rays)
    assert int(mk.sum()) > 500
    assert np.array_equal(mk, om)
    assert fields_equal(h, oh, ("t", "u", "v", "prim_id", "node_id"))
    if not prune:  # the single-pass walk on the same material: rays that trace more than 64 instances go back to the listing path
        sc.SetTunable("single_pass", 2)
        h, mk = sc.TraverseBatch(rays)
        assert 0 < sc.LastRedone() < len(rays)
        assert np.array_equal(mk, om)
        assert fields_equal(h, oh, ("t", "u", "v", "prim_id", "node_id"))


@pytest.mark.parametrize("dir_scale", [1.0, 0.25, 4.0])
def test_single_pass_walk_equals_the_listing_path_and_the_restatement(oracle, dir_scale):
    """The default scene path (k_scene_walk: no per-ray list, top-level tree and instance trees on one stack, rays it cannot
    certify re-done by the listing path) against the listing path alone (tunable single_pass = 0) and the restatement, on
    material chosen to break it: a crowd of overlapping instances (rays enter dozens of boxes, some more than 64), perfectly
    flat planes (hits that round to the near side of their own box entry), coincident copies (equal distances: the lower id
    wins), direction vectors far from unit length (the reference culls by comparing a distance with a parameter, nanosg.h:795),
    axis-parallel rays (no subtree is ever skipped for them).  Every field must agree."""
    from scene_fixture import xform
    from nanort_amd.wire import RAY_F32

    rng = np.random.default_rng(21)
    sv, sf = scenes.sphere(12, 6)
    sv = (sv - np.array([0, 5, 0], dtype=np.float32)).astype(np.float32)
    pv, pf = scenes.plane(8, 8)
    pv = pv.copy()
    pv[:, 1] = 0.0
    meshes = []
    for v, f in ((sv, sf), (pv, pf)):
        a = BVHAccel(np.float32)
        assert a.Build(f.shape[0], TriangleMesh(v, f))
        meshes.append((v, f, a, a.GetTree()))
    sc = Scene()
    O = ob.SceneOracle(oracle)

    def add(which, x):
        v, f, a, tree = meshes[which]
        sc.AddNode(a, x)
        O.add_node(v, f, x, tree=tree)

    for k in range(300):
        s = rng.uniform(0.05, 0.35, 3)
        x = xform(tuple(s), rng.uniform(0, 6.28), rng.uniform(0, 6.28), tuple(rng.uniform(-4, 4, 3) + np.array([0, 5, 0])))
        add(0, x)
        if k % 10 == 0:
            add(0, x)  # a coincident copy
    for k in range(6):
        x = xform((0.5, 1, 0.5), 0.3 * (k % 2), 0, (0, 1.0 + k, 0))
        add(1, x)
        add(1, x)
    assert sc.Commit() and O.commit()
    n = 6000
    rays = np.zeros(n, dtype=RAY_F32)
    org = rng.uniform(-7, 7, size=(n, 3)) + np.array([0, 5, 0])
    tgt = rng.uniform(-4, 4, size=(n, 3)) + np.array([0, 5, 0])
    d = tgt - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["org"] = org.astype(np.float32)
    rays["dir"] = (d * dir_scale).astype(np.float32)
    rays["max_t"] = 3.0e38
    rays["dir"][:50, 0] = 0.0  # axis-parallel: such rays never skip a top-level subtree, only instances
    rays["dir"][50:80, 1] = -0.0
    rays = np.concatenate([rays, scenes.camera_rays(96, 64)])
    oh, om = O.traverse(rays)
    sc.SetTunable("single_pass", 2)  # (whatever the default size rule says)
    h1, m1 = sc.TraverseBatch(rays)
    redone = sc.LastRedone()
    assert sc.LastPath() == 1
    sc.SetTunable("single_pass", 0)
    h0, m0 = sc.TraverseBatch(rays)
    assert sc.LastRedone() == 0 and sc.LastPath() == 0
    assert np.array_equal(m0, om) and fields_equal(h0, oh, ("t", "u", "v", "prim_id", "node_id"))
    assert np.array_equal(m1, om) and fields_equal(h1, oh, ("t", "u", "v", "prim_id", "node_id"))
    assert redone < len(rays)  # the walk itself finished the bulk ...
    if dir_scale == 1.0:
        assert redone < 0.2 * len(rays)
    if dir_scale == 4.0:
        assert redone > 0  # ... and with the reference's cull firing late, rays that trace more than 64 instances are handed over
    if dir_scale == 0.25:  # the default rule backs off: a batch handed over above a threshold makes the next calls skip the walk
        assert redone * 100 > len(rays)
        sc.SetTunable("single_pass", 1)
        sc.SetTunable("walk_min", 2)
        sc.SetTunable("walk_backoff_pct", 1)
        seen = []
        for _ in range(3):
            h3, m3 = sc.TraverseBatch(rays)
            seen.append(sc.LastRedone())
            assert np.array_equal(m3, om) and fields_equal(h3, oh, ("t", "u", "v", "prim_id", "node_id"))
        assert seen[0] == redone and seen[1] == 0 and seen[2] == 0 and sc.LastPath() == 0
    # thresholds of the phases never change a record
    sc.SetTunable("single_pass", 2)
    for name, value in (("walk_trav_min", 1), ("walk_trav_min", 48), ("walk_refill_min", 1), ("walk_refill_min", 64), ("cand_min", 16)):
        sc.SetTunable(name, value)
        h2, m2 = sc.TraverseBatch(rays)
        assert np.array_equal(m2, om) and fields_equal(h2, oh, ("t", "u", "v", "prim_id", "node_id")), (name, value)


@pytest.mark.parametrize("count", [1, 3, 8, 20])
def test_small_scenes_every_listing_form_gives_the_restatements_records(oracle, count):
    """Scenes of a handful of nodes: the trace kernel lists a ray's instances itself (one launch; the default up to 8 nodes), a
    listing launch of its own (fuse_scan = 0), the scan raised past the node count (scan_max = 64: no top-level tree at 20
    nodes) and the tree listing (scan_max = 1: a top-level tree even for 3 nodes) — the same records, the restatement's, on the
    same local trees (reference-built, adopted), overlapping instances, ragged batch size."""
    rng = np.random.default_rng(300 + count)
    sv, sf = scenes.sphere(32, 16)
    sv = sv - np.array([0, 5, 0], dtype=np.float32)
    nodes, idx, _ = oracle.build(sv, sf)
    a = BVHAccel(np.float32)
    a.SetMesh(TriangleMesh(sv, sf))
    a.SetTree(nodes, idx)
    xs = [xform(tuple(rng.uniform(0.15, 0.5, 3)), rng.uniform(0, 6.28), rng.uniform(0, 6.28), tuple(rng.uniform(-4, 4, 3) + np.array([0, 5, 0])))
          for _ in range(count)]
    O = ob.SceneOracle(oracle)
    for x in xs:
        O.add_node(sv, sf, x)
    assert O.commit()
    rays = scenes.camera_rays(331, 187)
    rays["org"] += rng.uniform(-0.2, 0.2, size=(rays.shape[0], 3)).astype(np.float32)
    oh, om = O.traverse(rays)
    assert 0.02 < om.mean() < 0.98
    for tun in ({}, {"fuse_scan": 0}, {"scan_max": 64}, {"scan_max": 1}, {"scan_max": 64, "fuse_scan": 0}):
        sc = Scene()
        for k, v in tun.items():
            sc.SetTunable(k, v)
        for x in xs:
            sc.AddNode(a, x)
        assert sc.Commit()
        h, m = sc.TraverseBatch(rays)
        assert np.array_equal(m, om), tun
        assert fields_equal(h[om == 1], oh[om == 1], ("t", "u", "v", "prim_id", "node_id")), tun


def test_walk_opens_one_leaf_meshes_too(oracle):
    """Instances of a mesh whose tree is ONE leaf (a quad: two triangles) next to instances of a sphere: the single-pass walk
    tests node 0's box and goes straight to the triangles.  Every field equals the restatement; the walk (not the listing path)
    did the work."""
    from scene_fixture import xform

    rng = np.random.default_rng(33)
    qv = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], dtype=np.float32)
    qf = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32)
    sv, sf = scenes.sphere(12, 6)
    sv = (sv - np.array([0, 5, 0], dtype=np.float32)).astype(np.float32)
    meshes = []
    for v, f in ((qv, qf), (sv, sf)):
        a = BVHAccel(np.float32)
        assert a.Build(f.shape[0], TriangleMesh(v, f))
        meshes.append((v, f, a, a.GetTree()))
    assert meshes[0][3][0].shape[0] == 1  # the quad's tree: one node, a leaf
    sc = Scene()
    O = ob.SceneOracle(oracle)
    for k in range(400):
        v, f, a, tree = meshes[0 if k % 3 else 1]
        s = rng.uniform(0.1, 0.6, 3) if k % 3 else rng.uniform(0.03, 0.1, 3)
        x = xform(tuple(s), rng.uniform(0, 6.28), rng.uniform(0, 6.28), tuple(rng.uniform(-5, 5, 3) + np.array([0, 5, 0])))
        sc.AddNode(a, x)
        O.add_node(v, f, x, tree=tree)
    assert sc.Commit() and O.commit()
    rays = scenes.camera_rays(200, 120)
    rays["org"] += rng.uniform(-0.5, 0.5, size=(rays.shape[0], 3)).astype(np.float32)
    oh, om = O.traverse(rays)
    h, m = sc.TraverseBatch(rays)
    assert sc.LastPath() == 1 and sc.LastRedone() < len(rays) // 4  # 400 nodes: the walk's, and it certified the bulk
    assert 0.05 < om.mean() < 0.99
    assert np.array_equal(m, om) and fields_equal(h, oh, ("t", "u", "v", "prim_id", "node_id"))
    sc.SetTunable("single_pass", 0)
    h0, m0 = sc.TraverseBatch(rays)
    assert sc.LastPath() == 0
    assert np.array_equal(m0, om) and fields_equal(h0, oh, ("t", "u", "v", "prim_id", "node_id"))
