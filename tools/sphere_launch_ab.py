import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from nanort_amd import BVHAccel, SphereGeometry, scenes
rays = scenes.particle_camera_rays(1920, 1080)
d = torch.from_numpy(rays.view(np.uint8)).cuda()
n = 1000000
g = SphereGeometry(*scenes.random_spheres(n))
a = BVHAccel(np.float32); assert a.Build(n, g)
o = torch.zeros(len(rays) * 16, dtype=torch.uint8, device="cuda"); m = torch.zeros(len(rays), dtype=torch.uint8, device="cuda")
for rnd in range(3):
    for lt in (1, 0):
        a.SetTunable("launch_timing", lt)
        ts = []
        for _ in range(7):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); a.TraverseBatchDevice(d, o, m); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        # back to back
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): a.TraverseBatchDevice(d, o, m)
        e1.record(); torch.cuda.synchronize()
        print("launch_timing=%d: single %.4f ms, back-to-back %.4f ms per launch, LastTraverseMs %.4f" % (lt, float(np.median(ts)), e0.elapsed_time(e1) / 10, a.LastTraverseMs()), flush=True)
