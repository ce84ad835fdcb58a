"""Two-level (instanced) traversal throughput (nrtSceneTraverseBatchDevice: listing + ONE trace launch) next to the
single-level kernel: the 5-node fixture of tests/scene_fixture.py at 1920x1080 (its plane alone through
nrtTraverseBatchDevice for comparison) and a 10 000-instance scene."""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from nanort_amd import BVHAccel, Scene, TriangleMesh, scenes
from nanort_amd.wire import SCENE_HIT_F32
from scene_fixture import instances, xform

def timed(fn, reps=7):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3

COUNTS = [int(x) for x in sys.argv[1:] if x.isdigit()] or [1000, 10000, 100000]  # `python tools/scene_probe.py 10000 --no-fixture`
FIXTURE = "--no-fixture" not in sys.argv
rays = scenes.camera_rays(1920, 1080)
d = torch.from_numpy(rays.view(np.uint8)).cuda()
o = torch.empty(len(rays) * SCENE_HIT_F32.itemsize, dtype=torch.uint8, device='cuda'); m = torch.empty(len(rays), dtype=torch.uint8, device='cuda')
sc = Scene(); keep = []
ntri = 0
for v, f, x in (() if not FIXTURE else instances(sphere_res=(264, 132), plane_res=(1000, 500))):
    a = BVHAccel(np.float32); a.Build(f.shape[0], TriangleMesh(v, f)); keep.append(a); sc.AddNode(a, x); ntri += f.shape[0]
if FIXTURE:
    sc.Commit()
    ms = timed(lambda: sc.TraverseBatchDevice(d, o, m))
    o1 = torch.empty(len(rays) * 16, dtype=torch.uint8, device='cuda')
    ms1 = timed(lambda: keep[0].TraverseBatchDevice(d, o1))
    print("5-node fixture (%d triangles in 2 meshes, 1920x1080): scene %.3f ms = %.1f Mrays/s | its plane alone, single-level kernel: %.3f ms = %.1f Mrays/s | ratio %.2f" % (
        ntri, ms, len(rays) / ms / 1e3, ms1, len(rays) / ms1 / 1e3, ms / ms1), flush=True)
rng = np.random.default_rng(5)
sv, sf = scenes.sphere(48, 24); sv = sv - np.array([0, 5, 0], dtype=np.float32)
a = BVHAccel(np.float32); a.Build(sf.shape[0], TriangleMesh(sv, sf))
for N in COUNTS:
    sc = Scene()
    for k in range(N):
        sc.AddNode(a, xform(tuple(rng.uniform(0.01, 0.04, 3)), rng.uniform(0, 6.28), rng.uniform(0, 6.28), tuple(rng.uniform(-9, 9, 3) + np.array([0, 5, 0]))))
    t0 = time.perf_counter(); sc.Commit(); tc = (time.perf_counter() - t0) * 1e3
    ms = timed(lambda: sc.TraverseBatchDevice(d, o, m), reps=5)
    print("%d instances of a %d-triangle mesh: commit %.1f ms, scene %.3f ms = %.1f Mrays/s, hit fraction %.3f, re-done by the listing path %d" % (N, sf.shape[0], tc, ms, len(rays) / ms / 1e3, float(m.float().mean()), sc.LastRedone()), flush=True)
    for kv in [x for x in sys.argv[1:] if "=" in x]:  # name=v0,v1,...: the same query under each value of a scene tunable, records compared with the first
        name, vals = kv.split("=")
        ref = None
        for v in vals.split(","):
            sc.SetTunable(name, int(v))
            msv = timed(lambda: sc.TraverseBatchDevice(d, o, m), reps=7)
            rec = (o.clone(), m.clone())
            same = True if ref is None else bool(torch.equal(rec[0], ref[0]) and torch.equal(rec[1], ref[1]))
            ref = ref or rec
            print("    %s = %s: %.3f ms = %.1f Mrays/s  records identical to the first: %s" % (name, v, msv, len(rays) / msv / 1e3, same), flush=True)
    if "--ab" in sys.argv:
        sc.SetTunable("single_pass", 0)
        ms0 = timed(lambda: sc.TraverseBatchDevice(d, o, m), reps=5)
        print("    listing path alone: %.3f ms = %.1f Mrays/s" % (ms0, len(rays) / ms0 / 1e3), flush=True)
