"""Is the bounce launch's fixed cost the heavy tail of per-ray step counts?  Fully dynamic distribution
(NRT_STATIC_PCT=0) with the rays in original order, longest-first and shortest-first (costs from the STATS kernel)."""
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
os.environ['NRT_DEBUG'] = '96'
b = BVHAccel(np.float32); b.Build(mesh.num_faces, mesh)
h, m = b.TraverseBatch(rays2)
cost = h['u'] + h['v']
del os.environ['NRT_DEBUG']
def timeit(acc, rays, tag):
    d = torch.from_numpy(np.ascontiguousarray(rays).view(np.uint8)).cuda(); o = torch.empty(len(rays) * 16, dtype=torch.uint8, device='cuda')
    ts = []
    for _ in range(9):
        acc.TraverseBatchDevice(d, o); ts.append(acc.LastTraverseMs())
    print("%-40s %.3f ms" % (tag, float(np.median(ts))), flush=True)
timeit(a, rays2, "default distribution, original order")
os.environ['NRT_STATIC_PCT'] = '0'
c = BVHAccel(np.float32); c.Build(mesh.num_faces, mesh)
timeit(c, rays2, "dynamic only, original order")
timeit(c, rays2[np.argsort(-cost, kind='stable')], "dynamic only, longest first")
timeit(c, rays2[np.argsort(cost, kind='stable')], "dynamic only, shortest first")
# longest 1% first, the rest in original order (keeps coherence of the bulk)
k = len(rays2) // 100
idx = np.argsort(-cost, kind='stable')
head = idx[:k]; rest = np.setdiff1d(np.arange(len(rays2)), head)
timeit(c, rays2[np.concatenate([head, rest])], "dynamic only, longest 1% first")
