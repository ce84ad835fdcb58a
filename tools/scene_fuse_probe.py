"""Scenes of a handful of nodes: k_scene_trace listing a ray's instances itself (tunable fuse_scan = 1, one launch) against the
listing kernel in a launch of its own (fuse_scan = 0): the 5-node fixture of tests/scene_fixture.py and a 3-node scene at
1920x1080, interleaved rounds on one box; the records of the two forms must be identical.
    python tools/scene_fuse_probe.py [rounds]"""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from nanort_amd import BVHAccel, Scene, TriangleMesh, scenes
from nanort_amd.wire import SCENE_HIT_F32
from scene_fixture import instances, xform

ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def timed(fn, reps=9):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


rays = scenes.camera_rays(1920, 1080)
d = torch.from_numpy(rays.view(np.uint8)).cuda()
n = len(rays)
out = [torch.empty(n * SCENE_HIT_F32.itemsize, dtype=torch.uint8, device='cuda') for _ in range(2)]
msk = [torch.empty(n, dtype=torch.uint8, device='cuda') for _ in range(2)]


def run(name, sc):
    res = {0: [], 1: []}
    for r in range(ROUNDS):
        for fuse in (0, 1):
            sc.SetTunable("fuse_scan", fuse)
            res[fuse].append(timed(lambda: sc.TraverseBatchDevice(d, out[fuse], msk[fuse])))
    same = bool(torch.equal(out[0], out[1]) and torch.equal(msk[0], msk[1]))
    a, b = float(np.median(res[0])), float(np.median(res[1]))
    print("%-28s two launches %.3f ms = %.1f Mrays/s | fused %.3f ms = %.1f Mrays/s | x%.3f | records %s | hit fraction %.3f" % (
        name, a, n / a / 1e3, b, n / b / 1e3, a / b, "IDENTICAL" if same else "DIFFER", float(msk[1].float().mean())), flush=True)
    return same


ok = True
sc = Scene(); keep = []
for v, f, x in instances(sphere_res=(264, 132), plane_res=(1000, 500)):
    a = BVHAccel(np.float32); a.Build(f.shape[0], TriangleMesh(v, f)); keep.append(a); sc.AddNode(a, x)
sc.Commit()
ok &= run("5-node fixture", sc)
sv, sf = scenes.sphere(48, 24); sv = sv - np.array([0, 5, 0], dtype=np.float32)
a = BVHAccel(np.float32); a.Build(sf.shape[0], TriangleMesh(sv, sf)); keep.append(a)
rng = np.random.default_rng(7)
for N in (3, 8):
    sc = Scene()
    for k in range(N):
        sc.AddNode(a, xform(tuple(rng.uniform(0.2, 0.5, 3)), rng.uniform(0, 6.28), rng.uniform(0, 6.28), tuple(rng.uniform(-6, 6, 3) + np.array([0, 5, 0]))))
    sc.Commit()
    ok &= run("%d spheres of %d triangles" % (N, sf.shape[0]), sc)
sys.exit(0 if ok else 1)
