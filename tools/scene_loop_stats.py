"""Where a wave of the single-pass scene walk (k_scene_walk) spends its trips: loop counters of the profiling build
(libnanort_hip_prof.so, nrtSceneDebugCounters) on the instanced scenes of tools/scene_probe.py.
    python tools/scene_loop_stats.py 10000 100000 [--fixture]"""
import ctypes
import os
import sys

os.environ["NRT_USE_PROF_LIB"] = "1"  # the profiling build of the library (include/nanort_hip_prof.h)
import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import torch

from nanort_amd import BVHAccel, Scene, TriangleMesh, capi, scenes
from nanort_amd.wire import SCENE_HIT_F32
from scene_fixture import instances, xform

L = capi.lib()
L.nrtSceneDebugCounters.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
L.nrtSceneDebugCounters.restype = ctypes.c_int


def report(name, sc, d, o, m, nrays):
    for k, v in [x.split("=") for x in sys.argv[1:] if "=" in x]:
        sc.SetTunable(k, int(v))
    sc.TraverseBatchDevice(d, o, m)
    sc.SetTunable("count_loops", 1)
    sc.TraverseBatchDevice(d, o, m)
    c = np.zeros(16, dtype=np.uint64)
    assert L.nrtSceneDebugCounters(sc._h, c.ctypes.data, 16) == 0
    sc.SetTunable("count_loops", 0)
    c = c.astype(np.float64)
    ticks = c[10] + c[11] + c[12]
    print("%s: per ray: top-level steps %.1f, instance steps %.1f, top-level leaves reached %.2f, instances opened %.2f, leaf-phase lane trips %.1f" % (
        name, c[5] / nrays, (c[4] - c[5]) / nrays, c[9] / nrays, c[8] / nrays, c[7] / nrays))
    print("    lanes per wave trip: level changes %.1f of 64 (%.0f blocks), inner phase %.1f (%.0f trips of 2 rounds), leaf phase %.1f (%.0f trips)" % (
        c[2] / max(c[1], 1), c[1], c[4] / max(2 * c[3], 1), c[3], c[7] / max(c[6], 1), c[6]))
    print("    share of the waves' clock: level changes %.2f, inner phase %.2f, leaf phase %.2f; re-done by the listing path: %d" % (
        c[10] / ticks, c[11] / ticks, c[12] / ticks, sc.LastRedone()), flush=True)


rays = scenes.camera_rays(1920, 1080)
d = torch.from_numpy(rays.view(np.uint8)).cuda()
o = torch.empty(len(rays) * SCENE_HIT_F32.itemsize, dtype=torch.uint8, device="cuda")
m = torch.empty(len(rays), dtype=torch.uint8, device="cuda")
if "--fixture" in sys.argv:
    sc, keep = Scene(), []
    for v, f, x in instances(sphere_res=(264, 132), plane_res=(1000, 500)):
        a = BVHAccel(np.float32)
        a.Build(f.shape[0], TriangleMesh(v, f))
        keep.append(a)
        sc.AddNode(a, x)
    sc.Commit()
    report("5-node fixture", sc, d, o, m, len(rays))
rng = np.random.default_rng(5)
sv, sf = scenes.sphere(48, 24)
sv = sv - np.array([0, 5, 0], dtype=np.float32)
a = BVHAccel(np.float32)
a.Build(sf.shape[0], TriangleMesh(sv, sf))
for N in [int(x) for x in sys.argv[1:] if x.isdigit()]:
    sc = Scene()
    for k in range(N):
        sc.AddNode(a, xform(tuple(rng.uniform(0.01, 0.04, 3)), rng.uniform(0, 6.28), rng.uniform(0, 6.28), tuple(rng.uniform(-9, 9, 3) + np.array([0, 5, 0]))))
    sc.Commit()
    report("%d instances" % N, sc, d, o, m, len(rays))
