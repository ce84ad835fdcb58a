"""Joint sweep of the single-pass scene walk's phase thresholds on the 10 000-instance scene of tools/scene_probe.py (round 6: after leaf
items went into the walk): walk_trav_min x walk_refill_min x cand_min x cand_busy_max, records compared with the defaults'.

    python tools/scene_sweep3.py [instances]
"""
import itertools
import sys
import time

import numpy as np

sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch  # noqa: E402

from nanort_amd import BVHAccel, Scene, TriangleMesh, scenes  # noqa: E402
from nanort_amd.wire import SCENE_HIT_F32  # noqa: E402
from scene_fixture import xform  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
rays = scenes.camera_rays(1920, 1080)
d = torch.from_numpy(rays.view(np.uint8)).cuda()
o = torch.empty(len(rays) * SCENE_HIT_F32.itemsize, dtype=torch.uint8, device="cuda")
m = torch.empty(len(rays), dtype=torch.uint8, device="cuda")
rng = np.random.default_rng(5)
sv, sf = scenes.sphere(48, 24)
sv = sv - np.array([0, 5, 0], dtype=np.float32)
a = BVHAccel(np.float32)
a.Build(sf.shape[0], TriangleMesh(sv, sf))
sc = Scene()
for k in range(N):
    sc.AddNode(a, xform(tuple(rng.uniform(0.01, 0.04, 3)), rng.uniform(0, 6.28), rng.uniform(0, 6.28), tuple(rng.uniform(-9, 9, 3) + np.array([0, 5, 0]))))
sc.Commit()


def timed(reps=7):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sc.TraverseBatchDevice(d, o, m)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


for _ in range(20):
    sc.TraverseBatchDevice(d, o, m)
base = timed()
ref = (o.clone(), m.clone())
print("%d instances, defaults (walk_trav_min 24, walk_refill_min 24, cand_min 1, cand_busy_max 64): %.3f ms = %.1f Mrays/s" % (N, base, len(rays) / base / 1e3), flush=True)
res = []
for tm, rm, cm, cb in itertools.product((16, 24, 32, 44), (16, 24, 36, 48), (1, 12, 24), (24, 64)):
    if cm == 1 and cb == 24:
        continue
    for k, v in (("walk_trav_min", tm), ("walk_refill_min", rm), ("cand_min", cm), ("cand_busy_max", cb)):
        sc.SetTunable(k, v)
    ms = timed(5)
    same = bool(torch.equal(o, ref[0]) and torch.equal(m, ref[1]))
    res.append((ms, tm, rm, cm, cb, same))
res.sort()
for ms, tm, rm, cm, cb, same in res[:12] + res[-3:]:
    print("  walk_trav_min %2d walk_refill_min %2d cand_min %2d cand_busy_max %2d: %.3f ms = %.1f Mrays/s (%+.1f %%) same=%s" % (tm, rm, cm, cb, ms, len(rays) / ms / 1e3, (base / ms - 1) * 100, same), flush=True)
print("all records identical to the defaults':", all(r[5] for r in res))
