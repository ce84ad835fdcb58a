"""GPU parity, traversal: the HIP kernel through the C ABI vs the CPU oracle, SAME node array
=> every hit record bit-identical (t, u, v, prim_id, hit flag), fp32 and fp64."""
import os

import numpy as np
import pytest

from helpers import assert_hits_identical, trace_options, is_distance_order_two_level_walk, is_reference_order_two_level_walk
from nanort_amd import BVHAccel, TriangleMesh, scenes
from nanort_amd.wire import ray_dtype, widen_rays

pytestmark = pytest.mark.gpu


def gpu_on_tree(real, v, f, nodes, idx, stride=None):
    a = BVHAccel(real)
    a.SetMesh(TriangleMesh(v, f, stride))
    a.SetTree(nodes, idx)
    return a


@pytest.fixture(scope="module")
def c1(oracle, c1_mesh):
    v, f = c1_mesh
    nodes, idx, _ = oracle.build(v, f)
    return v, f, nodes, idx, gpu_on_tree(np.float32, v, f, nodes, idx)


def test_c1_256_bit_exact_and_counters(oracle, c1):
    v, f, nodes, idx, a = c1
    rays = scenes.camera_rays(256, 256)
    oh, om, cnt = oracle.traverse(nodes, idx, v, f, rays, count=True)
    h, m = a.TraverseBatch(rays)
    assert_hits_identical(oh, om, h, m)
    import torch

    c = a.TraverseCountDevice(torch.from_numpy(rays.view(np.uint8)).cuda())
    assert (c["nodes_visited"], c["leaves_tested"], c["tris_tested"], c["max_stack"]) == tuple(int(x) for x in cnt)


@pytest.mark.parametrize("opts", [dict(cull=True), dict(skip=7), dict(range_=(12, 500)), dict(range_=(3, 3))])
def test_trace_options(oracle, c1, opts):
    v, f, nodes, idx, a = c1
    rays = scenes.camera_rays(256, 256)
    o = trace_options(**opts)
    oh, om = oracle.traverse(nodes, idx, v, f, rays, o)
    h, m = a.TraverseBatch(rays, o)
    assert_hits_identical(oh, om, h, m)


def test_golden_fixture_without_the_oracle(c1, golden_dir):
    """Straight against the reference's own output committed under tests/golden/."""
    import os

    v, f, nodes, idx, a = c1
    g = np.load(os.path.join(golden_dir, "c1_ref.npz"))
    a2 = gpu_on_tree(np.float32, v, f, g["nodes_f32"], g["indices_f32"])
    h, m = a2.TraverseBatch(scenes.camera_rays(256, 256))
    assert_hits_identical(g["hits_256_f32"], g["mask_256_f32"], h, m)
    rw = scenes.camera_rays(256, 256)
    rw["min_t"] = 19.0
    rw["max_t"] = 24.5914974
    h, m = a2.TraverseBatch(rw)
    assert_hits_identical(g["hits_256_window"], g["mask_256_window"], h, m)


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 255, 257, 1000, 4099])
def test_ragged_batch_sizes(oracle, c1, n):
    v, f, nodes, idx, a = c1
    rays = scenes.camera_rays(128, 128)[5000:5000 + n]
    h, m = a.TraverseBatch(rays)
    assert h.shape[0] == n
    if n:
        oh, om = oracle.traverse(nodes, idx, v, f, rays)
        assert_hits_identical(oh, om, h, m)


def test_fp64(oracle, c1_mesh):
    v, f = c1_mesh
    v64 = v.astype(np.float64)
    nodes, idx, _ = oracle.build(v64, f)
    a = gpu_on_tree(np.float64, v64, f, nodes, idx)
    rays = widen_rays(scenes.camera_rays(256, 256))
    oh, om = oracle.traverse(nodes, idx, v64, f, rays)
    h, m = a.TraverseBatch(rays)
    assert_hits_identical(oh, om, h, m)
    # the fp64 walk fetches a record's plane rows by the ray's signs; node arrays of 4 GiB and more read whole records and
    # select (tunable f64_row_fetch = 0 forces that path): the same records
    a.SetTunable("f64_row_fetch", 0)
    h0, m0 = a.TraverseBatch(rays)
    assert_hits_identical(oh, om, h0, m0)


def test_deep_reference_tree_spills_past_the_lds_stack(oracle):
    """The reference's own tree on a grid mesh is deep (X-only binning): the per-lane stack
    overflows its 32 LDS entries and must continue in the global spill buffer."""
    v, f = scenes.plane(300, 150)
    nodes, idx, st = oracle.build(v, f)
    assert st["max_tree_depth"] > 40
    a = gpu_on_tree(np.float32, v, f, nodes, idx)
    rays = scenes.camera_rays(640, 360)
    oh, om, cnt = oracle.traverse(nodes, idx, v, f, rays, count=True)
    assert cnt[3] > 33, "fixture no longer exercises the spill path"
    h, m = a.TraverseBatch(rays)
    assert_hits_identical(oh, om, h, m)


@pytest.mark.parametrize("real", [np.float32, np.float64])
def test_random_soup_and_hostile_rays(oracle, real):
    """Random triangle soup (degenerate triangles included), strided vertices, rays with zero
    direction components / axis-aligned / tiny windows: NaN-discarding slab test, vsafe_inverse,
    fp64 edge fallback."""
    rng = np.random.default_rng(7)
    nv, nf, se = 1500, 5000, 4
    vbuf = rng.uniform(-1, 1, size=(nv, se)).astype(real)
    vbuf[:50, :3] = np.round(vbuf[:50, :3] * 4) / 4  # lattice points: exact edge hits
    faces = rng.integers(0, nv, size=(nf, 3), dtype=np.uint32)
    faces[:20, 1] = faces[:20, 0]  # degenerate
    stride = se * vbuf.dtype.itemsize
    nodes, idx, _ = oracle.build(vbuf, faces, stride=stride)
    a = gpu_on_tree(real, vbuf, faces, nodes, idx, stride)
    n = 20000
    rays = np.zeros((n,), dtype=ray_dtype(real))
    rays["org"] = rng.uniform(-2, 2, size=(n, 3))
    rays["org"][:3000] = np.round(rays["org"][:3000] * 4) / 4
    d = rng.normal(size=(n, 3))
    d[:1000, 0] = 0.0
    d[1000:2000, 1] = 0.0
    d[2000:2500, :2] = 0.0
    d[2500:3000] = np.round(d[2500:3000])
    d[np.all(d == 0, axis=1)] = (0, 0, 1)
    d[3000:3200, 2] = 1e-9
    rays["dir"] = d
    rays["max_t"] = rng.choice([1e30, 0.5, 3.0], size=n)
    rays["min_t"] = rng.choice([0.0, 1e-3, 0.4], size=n)
    for o in (None, trace_options(cull=True)):
        oh, om = oracle.traverse(nodes, idx, vbuf, faces, rays, o, stride=stride)
        h, m = a.TraverseBatch(rays, o)
        assert_hits_identical(oh, om, h, m)


def test_device_resident_entry_point_matches_host_entry_point(c1):
    import torch

    from nanort_amd.wire import HIT_F32

    v, f, nodes, idx, a = c1
    rays = scenes.camera_rays(200, 100)
    h, m = a.TraverseBatch(rays)
    d_rays = torch.from_numpy(rays.view(np.uint8)).cuda()
    d_hits = torch.zeros(rays.shape[0] * 16, dtype=torch.uint8, device="cuda")
    d_mask = torch.zeros(rays.shape[0], dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        a.TraverseBatchDevice(d_rays, d_hits, d_mask)
    s.synchronize()
    assert d_hits.cpu().numpy().view(HIT_F32).tobytes() == h.tobytes()
    assert np.array_equal(d_mask.cpu().numpy(), m)
    assert a.LastTraverseMs() > 0


def test_launches_on_several_streams_overlap_safely():
    """One context, launches issued back to back on more streams than there are launch slots, no host sync in
    between (each launch owns its cursors and overflow stacks): every result equals the serial one."""
    import torch

    v, f = scenes.plane(200, 100)
    a = BVHAccel(np.float32)
    assert a.Build(f.shape[0], TriangleMesh(v, f))
    rays1 = scenes.camera_rays(480, 270)
    h1, m1 = a.TraverseBatch(rays1)
    rays2 = scenes.secondary_rays("bounce", v, f, rays1, h1, m1)
    h2, m2 = a.TraverseBatch(rays2)
    batches = [(rays1, h1, m1), (rays2, h2, m2)]
    streams = [torch.cuda.Stream() for _ in range(6)]
    dev = []
    for k in range(18):
        r, h, m = batches[k % 2]
        dev.append((torch.from_numpy(r.view(np.uint8)).cuda(), torch.zeros(r.shape[0] * 16, dtype=torch.uint8, device="cuda"),
                    torch.zeros(r.shape[0], dtype=torch.uint8, device="cuda")))
    torch.cuda.synchronize()
    for k, (d_r, d_h, d_m) in enumerate(dev):
        with torch.cuda.stream(streams[k % len(streams)]):
            a.TraverseBatchDevice(d_r, d_h, d_m)
    torch.cuda.synchronize()
    for k, (d_r, d_h, d_m) in enumerate(dev):
        r, h, m = batches[k % 2]
        assert d_h.cpu().numpy().tobytes() == h.tobytes(), "launch %d" % k
        assert np.array_equal(d_m.cpu().numpy(), m)


def test_lean_launches_give_the_same_records_and_rebuilds_still_wait():
    """Default launches record NO event: the kernel's last wave publishes a completion record (sequence number + start /
    end stamps) in page-locked memory.  Same records, on more streams than launch slots (a taken-over slot waits for
    that slot's record), LastTraverseMs comes from the kernel's own stamps and agrees with the event-bracketed time of
    nrtSetLaunchTiming(ctx, 1), and a rebuild issued right behind launches in flight waits for them (the launches read
    the tree's buffers, which the rebuild reuses in place)."""
    import torch

    v, f = scenes.plane(200, 100)
    a = BVHAccel(np.float32)
    assert a.Build(f.shape[0], TriangleMesh(v, f))
    assert a.GetTunable("launch_timing") == 0
    rays = scenes.camera_rays(480, 270)
    h, m = a.TraverseBatch(rays)
    d_r = torch.from_numpy(rays.view(np.uint8)).cuda()
    d_o = torch.zeros(rays.shape[0] * 16, dtype=torch.uint8, device="cuda")
    stamp_ms = []
    for _ in range(5):
        a.TraverseBatchDevice(d_r, d_o)
        stamp_ms.append(a.LastTraverseMs())
    assert min(stamp_ms) > 0
    a.SetLaunchTiming(True)
    event_ms = []
    for _ in range(5):
        a.TraverseBatchDevice(d_r, d_o)
        event_ms.append(a.LastTraverseMs())
    a.SetLaunchTiming(False)
    # the events also bracket the dispatch; the stamps run from the first block's start to the last wave's end
    assert 0.3 * np.median(event_ms) < np.median(stamp_ms) <= 1.1 * np.median(event_ms), (stamp_ms, event_ms)
    streams = [torch.cuda.Stream() for _ in range(6)]
    outs = [(torch.zeros(rays.shape[0] * 16, dtype=torch.uint8, device="cuda"), torch.zeros(rays.shape[0], dtype=torch.uint8, device="cuda"))
            for _ in range(12)]
    torch.cuda.synchronize()
    for k, (d_h, d_m) in enumerate(outs):
        with torch.cuda.stream(streams[k % len(streams)]):
            a.TraverseBatchDevice(d_r, d_h, d_m)
    # rebuild over another mesh while those launches may still be running: must not disturb them
    v2, f2 = scenes.sphere(64, 32)
    assert a.Build(f2.shape[0], TriangleMesh(v2, f2))
    torch.cuda.synchronize()
    for d_h, d_m in outs:
        assert d_h.cpu().numpy().tobytes() == h.tobytes()
        assert np.array_equal(d_m.cpu().numpy(), m)
    h2, m2 = a.TraverseBatch(rays)  # the new tree answers now
    assert a.LastTraverseMs() > 0 and h2.tobytes() != h.tobytes()
    # many launches back to back on one stream, then destroy with the last ones possibly still in flight
    for _ in range(50):
        a.TraverseBatchDevice(d_r, d_o)
    a.close()
    torch.cuda.synchronize()


def test_work_distribution_tunables_cover_every_ray():
    """The static bands / dynamic chunks of the persistent kernel (traverse.hip, Claim) under odd settings: every ray of the
    batch is traced exactly once, whatever the split — the records equal the default configuration's."""
    v, f = scenes.plane(120, 60)
    a = BVHAccel(np.float32)
    assert a.Build(f.shape[0], TriangleMesh(v, f))
    for w, h_ in ((640, 361), (97, 13), (1, 1)):
        rays = scenes.camera_rays(w, h_)
        ref, refm = a.TraverseBatch(rays)
        for combo in (dict(static_pct=0), dict(static_pct=100), dict(static_bands=1), dict(static_bands=64, static_pct=90),
                      dict(chunk=16, parts=3), dict(chunk=1000, parts=16, static_pct=10), dict(blocks_per_cu=1, static_bands=3),
                      dict(refill_min=1, trav_min=1, trav_min4=1, leaf_min=1), dict(refill_min=64, trav_min=64, trav_min4=64, leaf_min=64)):
            saved = {k: a.GetTunable(k) for k in combo}
            for k, val in combo.items():
                a.SetTunable(k, val)
            h, m = a.TraverseBatch(rays)
            for k, val in saved.items():
                a.SetTunable(k, val)
            assert h.tobytes() == ref.tobytes() and np.array_equal(m, refm), (w, h_, combo)


def test_loaded_tree_with_unreachable_records(oracle, c1):
    """A loaded node array may carry records no path reaches (the reference's Load() takes any array,
    nanort.h:2219-2275); they must be ignored, whatever they contain."""
    v, f, nodes, idx, _ = c1
    junk = np.zeros(5, dtype=nodes.dtype)
    junk["flag"] = 0                      # look like branches
    junk["data"][:, 0] = 0xFFFFFFF0       # children far out of range
    junk["data"][:, 1] = 5
    junk["axis"] = 7
    padded = np.concatenate([nodes, junk])
    a = BVHAccel(np.float32)
    a.SetMesh(TriangleMesh(v, f))
    a.SetTree(padded, idx)
    rays = scenes.camera_rays(128, 128)
    h, m = a.TraverseBatch(rays)
    oh, om = oracle.traverse(nodes, idx, v, f, rays)
    assert_hits_identical(oh, om, h, m)


def test_error_paths(c1_mesh):
    from nanort_amd import NrtError

    v, f = c1_mesh
    a = BVHAccel(np.float32)
    with pytest.raises(NrtError):  # no tree yet
        a.SetMesh(TriangleMesh(v, f))
        a.TraverseBatch(scenes.camera_rays(8, 8))
    with pytest.raises(TypeError):  # precision mismatch caught on the host side
        a.SetMesh(TriangleMesh(v.astype(np.float64), f))
    from nanort_amd.wire import NODE_F32

    bad = np.zeros((1,), dtype=NODE_F32)
    bad["flag"] = 0
    bad["data"] = (5, 6)
    with pytest.raises(NrtError):  # child index out of range
        a.SetTree(bad, np.arange(f.shape[0], dtype=np.uint32))


def test_contexts_release_their_device_memory():
    """Create / use / destroy many contexts (triangles, spheres, cylinders, a two-level scene): free HBM returns to where
    it started (the contexts own grow-only buffers and launch slots; nrtDestroy must release all of them)."""
    import gc

    import torch

    from nanort_amd import CylinderGeometry, Scene, SphereGeometry

    v, f = scenes.plane(64, 32)
    c, r = scenes.random_spheres(2000)
    cv, cr = scenes.random_cylinders(500)
    rays = scenes.camera_rays(160, 90)
    prays = scenes.particle_camera_rays(64, 65)

    def cycle():
        a = BVHAccel(np.float32)
        assert a.Build(f.shape[0], TriangleMesh(v, f))
        a.TraverseBatch(rays)
        s = BVHAccel(np.float32)
        assert s.Build(2000, SphereGeometry(c, r))
        s.TraverseBatch(prays)
        y = BVHAccel(np.float32)
        assert y.Build(500, CylinderGeometry(cv, cr))
        y.TraverseBatch(prays)
        sc = Scene()
        sc.AddNode(a, np.eye(4, dtype=np.float32))
        assert sc.Commit()
        sc.TraverseBatch(rays)
        del sc
        for x in (a, s, y):
            x.close()

    cycle()
    gc.collect()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(25):
        cycle()
    gc.collect()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 << 20, "leaked %.1f MiB over 25 cycles" % ((free0 - free1) / 2**20)


@pytest.mark.parametrize("real", [np.float32, np.float64])
def test_occlusion_queries_equal_the_closest_hit_flags(real, c1_mesh):
    """Opt-in extension: nrtOccludedBatch* stops a ray at the first accepted primitive; its flags must be exactly the hit
    flags of the closest-hit traversal — camera, shadow and bounce rays, with and without restrictive trace options."""
    import torch

    for (v, f), (w, h) in ((c1_mesh, (256, 256)), (scenes.plane(200, 100), (480, 270))):
        a = BVHAccel(real)
        assert a.Build(f.shape[0], TriangleMesh(v.astype(real), f))
        rays1 = scenes.camera_rays(w, h)
        a32 = BVHAccel(np.float32)
        assert a32.Build(f.shape[0], TriangleMesh(v, f))
        h1, m1 = a32.TraverseBatch(rays1)
        sets = [rays1, scenes.secondary_rays("shadow", v, f, rays1, h1, m1), scenes.secondary_rays("bounce", v, f, rays1, h1, m1)]
        for rays in sets:
            if real == np.float64:
                rays = widen_rays(rays)
            for o in (None, trace_options(cull=True), trace_options(range_=(10, f.shape[0] // 2), skip=11)):
                _, m = a.TraverseBatch(rays, o)
                occ = a.OccludedBatch(rays, o)
                assert np.array_equal(occ, m)
        d = torch.from_numpy(np.ascontiguousarray(rays).view(np.uint8)).cuda()
        dm = torch.zeros(rays.shape[0], dtype=torch.uint8, device="cuda")
        a.OccludedBatchDevice(d, dm)
        torch.cuda.synchronize()
        assert np.array_equal(dm.cpu().numpy(), a.TraverseBatch(rays)[1])


def test_randomised_parity_soak_short():
    """tests/checks/fuzz_parity.py for a few seconds: random grid-aligned / flat / smooth meshes with duplicated and degenerate
    triangles, hostile rays, random trace options, fp32 and fp64, built and adopted trees, closest-hit and occlusion
    queries — GPU == oracle on the same node array, bit for bit.  (Round 1 ran it for 250 s: 11 140 rounds, 44.6 M rays.)"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "checks", "fuzz_parity.py"), "8", "7"], cwd=root, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "fuzz ok" in r.stdout, r.stdout[-2000:]


def test_randomised_primitive_and_scene_soak_short():
    """tests/checks/fuzz_prims_scenes.py for a few seconds: random sphere / cylinder sets (zero, negative, denormal and huge radii,
    zero-length and lattice-aligned cylinders, caps on and off, random build options and prim_ids_range) and random
    two-level scenes (rotated, mirrored, nearly flat, duplicated instances, up to 90 nodes) under hostile rays — GPU ==
    the restatements on the same node arrays.  (Round 1 ran it for 90 s: 5 573 rounds, 18.6 M rays.)"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "checks", "fuzz_prims_scenes.py"), "8", "5"], cwd=root, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "fuzz ok" in r.stdout, r.stdout[-2000:]


def test_randomised_lifecycle_soak_short():
    """tests/checks/fuzz_lifecycle.py for a few seconds: one context rebuilt over meshes of 4 ... 560 000 triangles while launches of
    its previous tree are in flight on six streams (four launch slots), tunables and walk variants flipping between launches,
    closest-hit / occlusion / multi-batch / host-buffer calls of ragged sizes mixed — every result equal to a fresh context's, byte
    for byte.  (Round 6 ran it for 150 s: 10 371 rounds, 4 692 rebuilds, 46 845 launches.)"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "checks", "fuzz_lifecycle.py"), "8", "3"], cwd=root, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "fuzz_lifecycle ok" in r.stdout, r.stdout[-2000:]


def test_host_entry_point_from_several_threads(c1):
    """nrtTraverseBatch (host buffers) issued from four host threads on ONE context: the calls share the context's staging
    buffers and are served one at a time — every result equals the single-threaded one."""
    import threading

    v, f, nodes, idx, a = c1
    batches = [scenes.camera_rays(200 + 8 * k, 100) for k in range(4)]
    want = [a.TraverseBatch(r) for r in batches]
    got = [None] * 4
    errors = []

    def work(k):
        try:
            for _ in range(6):
                got[k] = a.TraverseBatch(batches[k])
                assert got[k][0].tobytes() == want[k][0].tobytes() and np.array_equal(got[k][1], want[k][1])
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("real,mode", [(np.float32, "default"), (np.float32, "order4"), (np.float32, "one_level"), (np.float64, "default")])
def test_several_batches_in_one_launch_equal_separate_launches(real, mode):
    """nrtTraverseBatchesDevice: independent batches of unequal sizes (one empty, one without a mask) walked by ONE persistent
    launch — the records of each batch are exactly those of its own nrtTraverseBatchDevice call (fp64 contexts launch the
    batches one after the other: same contract)."""
    import torch

    from nanort_amd.wire import hit_dtype

    v, f = scenes.plane(200, 100)
    a = BVHAccel(real)
    if mode == "one_level":
        a.SetTunable("wide4", 0)
    assert a.Build(f.shape[0], TriangleMesh(v.astype(real), f))
    if mode == "order4":
        a.SetTunable("order4", 1)
    rays1 = scenes.camera_rays(480, 270)
    h1, m1 = a.TraverseBatch(rays1 if real == np.float32 else widen_rays(rays1))
    h1_32 = h1
    if real != np.float32:
        from nanort_amd.wire import HIT_F32

        h1_32 = np.zeros(h1.shape[0], dtype=HIT_F32)
        for k in ("t", "u", "v"):
            h1_32[k] = h1[k].astype(np.float32)
        h1_32["prim_id"] = h1["prim_id"]
    rays2 = scenes.secondary_rays("bounce", v, f, rays1, h1_32, m1)
    rays3 = scenes.secondary_rays("shadow", v, f, rays1, h1_32, m1)
    sets = [rays1, rays2[:1000], rays2[:0], rays3, rays2]
    if real != np.float32:
        sets = [widen_rays(r) for r in sets]
    HIT = hit_dtype(real)
    want = [a.TraverseBatch(r) if r.shape[0] else (np.zeros(0, HIT), np.zeros(0, np.uint8)) for r in sets]
    dev = []
    for k, r in enumerate(sets):
        d_r = torch.from_numpy(np.ascontiguousarray(r).view(np.uint8)).cuda() if r.shape[0] else torch.zeros(1, dtype=torch.uint8, device="cuda")
        d_h = torch.full((max(1, r.shape[0]) * HIT.itemsize,), 0xCD, dtype=torch.uint8, device="cuda")
        d_m = None if k == 3 else torch.full((max(1, r.shape[0]),), 0xCD, dtype=torch.uint8, device="cuda")
        dev.append((d_r, d_h, d_m, r.shape[0]))
    counts = a.TraverseBatchesDevice(dev)
    torch.cuda.synchronize()
    assert counts == [r.shape[0] for r in sets]
    for k, ((d_r, d_h, d_m, n), (h, m)) in enumerate(zip(dev, want)):
        got = d_h.cpu().numpy()[: n * HIT.itemsize].view(HIT)
        assert all(got[f_].tobytes() == h[f_].tobytes() for f_ in ("t", "u", "v", "prim_id")), "batch %d" % k  # (fp64 records carry 4 padding bytes)
        if d_m is not None and n:
            assert np.array_equal(d_m.cpu().numpy()[:n], m), "batch %d mask" % k
    # and again right behind it on the same stream (the slot's cursors and completion record are handed on)
    a.TraverseBatchesDevice(dev[:2])
    a.TraverseBatchDevice(dev[0][0], dev[0][1], dev[0][2])
    torch.cuda.synchronize()
    got = dev[0][1].cpu().numpy().view(HIT)
    assert all(got[f_].tobytes() == want[0][0][f_].tobytes() for f_ in ("t", "u", "v", "prim_id"))
    assert a.LastTraverseMs() > 0


@pytest.mark.parametrize("real", [np.float32, np.float64])
def test_host_batches_in_one_launch_equal_separate_host_calls(real):
    """nrtTraverseBatches: host batches of unequal sizes (one empty, one an occlusion query) uploaded together and walked by
    ONE launch return exactly the records / flags of separate nrtTraverseBatch / nrtOccludedBatch calls (fp64: the batches are
    launched one after the other behind the same entry point)."""
    from nanort_amd import NrtError

    v, f = scenes.plane(160, 90)
    a = BVHAccel(real)
    assert a.Build(f.shape[0], TriangleMesh(v.astype(real), f))
    rays1 = scenes.camera_rays(320, 180)
    h1, m1 = a.TraverseBatch(rays1 if real == np.float32 else widen_rays(rays1))
    h32 = h1
    if real != np.float32:
        from nanort_amd.wire import HIT_F32

        h32 = np.zeros(h1.shape[0], dtype=HIT_F32)
        for k in ("t", "u", "v"):
            h32[k] = h1[k].astype(np.float32)
        h32["prim_id"] = h1["prim_id"]
    bounce = scenes.secondary_rays("bounce", v, f, rays1, h32, m1)
    shadow = scenes.secondary_rays("shadow", v, f, rays1, h32, m1)
    sets = [shadow, bounce, bounce[:0], rays1[:777]]
    if real != np.float32:
        sets = [widen_rays(r) for r in sets]
    got = a.TraverseBatches([(sets[0], "occlusion"), sets[1], sets[2], sets[3]])
    assert got[0][0] is None and np.array_equal(got[0][1], a.OccludedBatch(sets[0]))
    for k in (1, 2, 3):
        if sets[k].shape[0] == 0:
            assert got[k][0].shape[0] == 0
            continue
        h, m = a.TraverseBatch(sets[k])
        assert all(got[k][0][f_].tobytes() == h[f_].tobytes() for f_ in ("t", "u", "v", "prim_id")), k
        assert np.array_equal(got[k][1], m)
    if real == np.float32:
        assert is_reference_order_two_level_walk(a.LastKernelName())
    with pytest.raises(NrtError):  # an occlusion batch needs its flag array (checked by the C entry point, not the binding)
        import ctypes

        nb = 1
        r = (ctypes.c_void_p * nb)(sets[0].ctypes.data)
        n = (ctypes.c_uint64 * nb)(sets[0].shape[0])
        fl = (ctypes.c_uint32 * nb)(1)
        hh = (ctypes.c_void_p * nb)(None)
        a._check(getattr(a._L, "nrtTraverseBatches_" + a._s)(a._h, nb, r, n, None, hh, None, fl))
    # ... but no record table at all when every batch is an occlusion query: host and device entry points
    import ctypes

    import torch

    occ = [sets[0], sets[0][::3].copy()]
    r = (ctypes.c_void_p * 2)(*[x.ctypes.data for x in occ])
    n = (ctypes.c_uint64 * 2)(*[x.shape[0] for x in occ])
    fl = (ctypes.c_uint32 * 2)(1, 1)
    outs = [np.full(x.shape[0], 0xCD, dtype=np.uint8) for x in occ]
    mm = (ctypes.c_void_p * 2)(*[x.ctypes.data for x in outs])
    a._check(getattr(a._L, "nrtTraverseBatches_" + a._s)(a._h, 2, r, n, None, None, mm, fl))
    want = [a.OccludedBatch(x) for x in occ]
    assert np.array_equal(outs[0], want[0]) and np.array_equal(outs[1], want[1])
    d_r = [torch.from_numpy(x.view(np.uint8).reshape(-1)).cuda() for x in occ]
    d_m = [torch.full((x.shape[0],), 0xCD, dtype=torch.uint8, device="cuda") for x in occ]
    r = (ctypes.c_void_p * 2)(*[x.data_ptr() for x in d_r])
    mm = (ctypes.c_void_p * 2)(*[x.data_ptr() for x in d_m])
    a._check(getattr(a._L, "nrtTraverseBatchesDevice_" + a._s)(a._h, 2, r, n, None, None, mm, fl, None))
    torch.cuda.synchronize()
    assert np.array_equal(d_m[0].cpu().numpy(), want[0]) and np.array_equal(d_m[1].cpu().numpy(), want[1])
