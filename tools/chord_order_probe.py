"""Can a cheap per-ray predictor stand in for the unknown ray costs when ordering a launch?  Key = length of the ray's
segment inside the scene's bounding box.  For each C3 wave: launch time in original order, with the top p% by key
first, and with the top 1% by TRUE cost first (costs from the STATS kernel) — all with fully dynamic distribution —
plus what the key captures of the expensive rays."""
import os, sys
import numpy as np
sys.path.insert(0, '.')
import torch
from nanort_amd import BVHAccel, TriangleMesh, scenes
v, f = scenes.plane(1000, 500); mesh = TriangleMesh(v, f)
rays1 = scenes.camera_rays(1920, 1080)
a = BVHAccel(np.float32); a.Build(mesh.num_faces, mesh)
h1, m1 = a.TraverseBatch(rays1)
rays2 = scenes.secondary_rays("bounce", v, f, rays1, h1, m1)
bmin, bmax = v.min(axis=0), v.max(axis=0)

def chord(rays):
    o, d = rays['org'].astype(np.float64), rays['dir'].astype(np.float64)
    with np.errstate(divide='ignore', invalid='ignore'):
        inv = 1.0 / d
        t0 = (bmin - o) * inv; t1 = (bmax - o) * inv
    lo = np.nanmax(np.minimum(t0, t1), axis=1); hi = np.nanmin(np.maximum(t0, t1), axis=1)
    lo = np.maximum(lo, rays['min_t']); hi = np.minimum(hi, rays['max_t'])
    return np.maximum(hi - lo, 0) * np.linalg.norm(d, axis=1)

os.environ['NRT_DEBUG'] = '96'
b = BVHAccel(np.float32); b.Build(mesh.num_faces, mesh)
costs = {}
for name, rays in (("primary", rays1), ("bounce", rays2)):
    h, m = b.TraverseBatch(rays); costs[name] = (h['u'] + h['v']).astype(np.float64), h['u'].copy()
del os.environ['NRT_DEBUG']
def timeit(acc, rays, tag):
    d = torch.from_numpy(np.ascontiguousarray(rays).view(np.uint8)).cuda(); o = torch.empty(len(rays) * 16, dtype=torch.uint8, device='cuda')
    ts = []
    for _ in range(9):
        acc.TraverseBatchDevice(d, o); ts.append(acc.LastTraverseMs())
    print("  %-44s %.3f ms" % (tag, float(np.median(ts))), flush=True)
    return float(np.median(ts))
os.environ['NRT_STATIC_PCT'] = '0'
c = BVHAccel(np.float32); c.Build(mesh.num_faces, mesh)
for name, rays in (("primary", rays1), ("bounce", rays2)):
    cost, steps = costs[name]
    key = chord(rays)
    n = len(rays)
    print(name, "rays", n, "corr(steps, chord) %.3f" % np.corrcoef(steps, key)[0, 1], "max steps", steps.max())
    timeit(a, rays, "default distribution, original order")
    timeit(c, rays, "dynamic only, original order")
    by_cost = np.argsort(-cost, kind='stable')
    k = n // 100
    rest = np.ones(n, bool); rest[by_cost[:k]] = False
    timeit(c, rays[np.concatenate([by_cost[:k], np.nonzero(rest)[0]])], "dynamic only, top 1% by TRUE cost first")
    by_key = np.argsort(-key, kind='stable')
    for pct in (1, 3, 6, 12):
        k = n * pct // 100
        rest = np.ones(n, bool); rest[by_key[:k]] = False
        left = steps[rest]
        tag = "dynamic only, top %d%% by chord first" % pct
        t = timeit(c, rays[np.concatenate([by_key[:k], np.nonzero(rest)[0]])], tag)
        print("      (longest ray left behind: %d steps; 99.9th pct of the rest %d)" % (left.max(), np.percentile(left, 99.9)))
