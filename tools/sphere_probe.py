import sys, numpy as np
sys.path.insert(0, '.')
import torch, hashlib
from nanort_amd import BVHAccel, SphereGeometry, scenes
n = 1000000
c, r = scenes.random_spheres(n)
rays = scenes.particle_camera_rays(1920, 1080)
a = BVHAccel(np.float32)
assert a.Build(n, SphereGeometry(c, r))
d = torch.from_numpy(rays.view(np.uint8)).cuda(); o = torch.empty(len(rays) * 16, dtype=torch.uint8, device='cuda'); m = torch.empty(len(rays), dtype=torch.uint8, device='cuda')
ts = []
for _ in range(10):
    a.TraverseBatchDevice(d, o, m); ts.append(a.LastTraverseMs())
print("spheres 1M: traverse %.4f ms (median of 8), records %s" % (float(np.median(ts[2:])), hashlib.md5(o.cpu().numpy().tobytes()).hexdigest()[:10]))
