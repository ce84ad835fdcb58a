import os, sys
import numpy as np
sys.path.insert(0, '.')
import torch
from nanort_amd import BVHAccel, TriangleMesh, scenes
v, f = scenes.plane(1000, 500); mesh = TriangleMesh(v, f)
rays = scenes.camera_rays(1920, 1080)
d = torch.from_numpy(rays.view(np.uint8)).cuda(); o = torch.empty(len(rays) * 16, dtype=torch.uint8, device='cuda')
for cfg in [dict(NRT_DEBUG=6), dict(NRT_DEBUG=0)]:
    for k, val in cfg.items(): os.environ[k] = str(val)
    a = BVHAccel(np.float32); a.Build(mesh.num_faces, mesh)
    for n in (64, 4096, 65536, 262144, 1048576, 2073600):
        ts = []
        for _ in range(5):
            a.TraverseBatchDevice(d[: n * 36], o[: n * 16]); ts.append(a.LastTraverseMs())
        print(cfg, n, 'rays: %.4f ms' % np.median(ts), flush=True)
