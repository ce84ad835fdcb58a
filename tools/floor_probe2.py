import os, sys
import numpy as np
sys.path.insert(0, '.')
import torch
from nanort_amd import BVHAccel, TriangleMesh, scenes
v, f = scenes.plane(1000, 500); mesh = TriangleMesh(v, f)
rays = scenes.camera_rays(1920, 1080)
n = len(rays)
d = torch.from_numpy(rays.view(np.uint8)).cuda(); o = torch.empty(n * 16, dtype=torch.uint8, device='cuda'); m = torch.empty(n, dtype=torch.uint8, device='cuda')
for cfg in [dict(NRT_DEBUG=6), dict(NRT_DEBUG=0)]:
    for k, val in cfg.items(): os.environ[k] = str(val)
    a = BVHAccel(np.float32); a.Build(mesh.num_faces, mesh)
    for label, mask in (('no mask', None), ('mask', m)):
        ts = []
        for _ in range(7):
            a.TraverseBatchDevice(d, o, mask); ts.append(a.LastTraverseMs())
        print(cfg, label, '%.4f ms' % np.median(ts), ['%.3f' % t for t in ts], flush=True)
