"""Scenes of 9 .. 63 nodes, 1920x1080 camera rays in HBM: the three ways to trace them —
  (a) listing kernel over the top-level tree + k_scene_trace (two launches; the default below walk_min = 64 nodes),
  (b) k_scene_trace testing every world box itself (tunable scan_max raised to the node count: one launch),
  (c) the single-pass walk k_scene_walk (tunable single_pass = 2: one launch).
Records of the three must be identical.    python tools/scene_small_probe.py [rounds]"""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from nanort_amd import BVHAccel, Scene, TriangleMesh, scenes
from nanort_amd.wire import SCENE_HIT_F32
from scene_fixture import xform

ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def timed(fn, reps=7):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


rays = scenes.camera_rays(1920, 1080)
d = torch.from_numpy(rays.view(np.uint8)).cuda()
n = len(rays)
outs = [torch.empty(n * SCENE_HIT_F32.itemsize, dtype=torch.uint8, device='cuda') for _ in range(3)]
msks = [torch.empty(n, dtype=torch.uint8, device='cuda') for _ in range(3)]
meshes = {}
for name, res in (("2208-tri sphere", (48, 24)), ("69168-tri sphere", (264, 132))):
    sv, sf = scenes.sphere(*res); sv = sv - np.array([0, 5, 0], dtype=np.float32)
    a = BVHAccel(np.float32); a.Build(sf.shape[0], TriangleMesh(sv, sf)); meshes[name] = a
ok = True
for name, a in meshes.items():
    for N in (12, 24, 48, 63):
        rng = np.random.default_rng(100 + N)
        xs = [xform(tuple(rng.uniform(0.1, 0.3, 3)), rng.uniform(0, 6.28), rng.uniform(0, 6.28), tuple(rng.uniform(-7, 7, 3) + np.array([0, 5, 0]))) for _ in range(N)]
        variants = []
        for label, tun in (("tree listing + trace", {}), ("scan in the trace kernel", {"scan_max": 64}), ("single-pass walk", {"single_pass": 2})):
            sc = Scene()
            for k, v in tun.items():
                sc.SetTunable(k, v)
            for x in xs:
                sc.AddNode(a, x)
            sc.Commit()
            variants.append((label, sc))
        res = [[] for _ in variants]
        for r in range(ROUNDS):
            for j, (label, sc) in enumerate(variants):
                res[j].append(timed(lambda: sc.TraverseBatchDevice(d, outs[j], msks[j])))
        same = all(bool(torch.equal(outs[0], outs[j]) and torch.equal(msks[0], msks[j])) for j in (1, 2))
        ok &= same
        print("%2d x %-16s %s | records %s | hit fraction %.3f" % (N, name, " | ".join("%s %.3f ms" % (variants[j][0], float(np.median(res[j]))) for j in range(3)),
                                                                  "IDENTICAL" if same else "DIFFER", float(msks[0].float().mean())), flush=True)
sys.exit(0 if ok else 1)
