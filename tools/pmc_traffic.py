"""Workload for the HBM-traffic PMC passes: C3 primary + bounce launches of the production kernel, plus
calibration launches of the same kernel in stream-only mode (NRT_DEBUG=6: read every ray, write every hit,
no traversal) whose byte count is known exactly (instantiated with a different LDS stack depth so that the
kernel NAME tells the two apart in the counter CSV)."""
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
d1 = torch.from_numpy(rays1.view(np.uint8)).cuda(); d2 = torch.from_numpy(rays2.view(np.uint8)).cuda()
o1 = torch.empty(len(rays1) * 16, dtype=torch.uint8, device='cuda'); o2 = torch.empty(len(rays2) * 16, dtype=torch.uint8, device='cuda')
mk1 = torch.empty(len(rays1), dtype=torch.uint8, device='cuda'); mk2 = torch.empty(len(rays2), dtype=torch.uint8, device='cuda')
for _ in range(6):
    a.TraverseBatchDevice(d1, o1, mk1); a.TraverseBatchDevice(d2, o2, mk2)
torch.cuda.synchronize()
os.environ['NRT_DEBUG'] = '6'; os.environ['NRT_WIDE_STACK'] = '16'
c = BVHAccel(np.float32); c.Build(mesh.num_faces, mesh)
for _ in range(6):
    c.TraverseBatchDevice(d1, o1, mk1)
torch.cuda.synchronize()
print('rays', len(rays1), len(rays2))
