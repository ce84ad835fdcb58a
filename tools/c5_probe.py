import sys, numpy as np
sys.path.insert(0, '.')
import torch
from nanort_amd import BVHAccel, TriangleMesh, scenes
from nanort_amd.wire import widen_rays
v, f = scenes.plane(1000, 500)
for real in (np.float32, np.float64):
    vv = v.astype(real); m = TriangleMesh(vv, f)
    a = BVHAccel(real)
    ts = []
    for _ in range(4):
        a.Build(m.num_faces, m); ts.append(a.LastBuildMs())
    rays = scenes.camera_rays(1920, 1080)
    if real == np.float64: rays = widen_rays(rays)
    d = torch.from_numpy(rays.view(np.uint8)).cuda()
    o = torch.empty(len(rays) * (16 if real == np.float32 else 32), dtype=torch.uint8, device='cuda')
    tt = []
    for _ in range(6):
        a.TraverseBatchDevice(d, o); tt.append(a.LastTraverseMs())
    c = a.TraverseCountDevice(d)
    print(np.dtype(real).name, 'build ms %.3f' % np.median(ts[1:]), 'primary ms %.3f = %.0f Mrays/s' % (np.median(tt), len(rays) / np.median(tt) / 1e3), 'nodes/ray %.1f' % (c['nodes_visited'] / len(rays)), flush=True)
