"""One or two tree levels per step for the sphere and cylinder kinds (tunable wide4, applied at Build): kernel time of the
particle / cylinder example workloads at 1920x1080 and whether the records are identical."""
import sys

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402

from nanort_amd import BVHAccel, CylinderGeometry, SphereGeometry, scenes  # noqa: E402

rays = scenes.particle_camera_rays(1920, 1080)
d = torch.from_numpy(rays.view(np.uint8)).cuda()
for name, geom, n, rec in (("spheres", lambda n: SphereGeometry(*scenes.random_spheres(n)), 1000000, 16),
                           ("cylinders", lambda n: CylinderGeometry(*scenes.random_cylinders(n)), 20000, 28)):
    g = geom(n)
    outs = {}
    for rnd in range(2):
        for w4 in (0, 1):
            a = BVHAccel(np.float32)
            a.SetTunable("wide4", w4)
            assert a.Build(n, g)
            o = torch.zeros(len(rays) * rec, dtype=torch.uint8, device="cuda")
            m = torch.zeros(len(rays), dtype=torch.uint8, device="cuda")
            ts = []
            for _ in range(5):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                a.TraverseBatchDevice(d, o, m)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            outs[w4] = (o.cpu().numpy().tobytes(), m.cpu().numpy().tobytes())
            print("%s n=%d wide4=%d: %.4f ms incl. the post pass = %.1f Mrays/s (%s)" % (name, n, w4, float(np.median(ts)), len(rays) / float(np.median(ts)) / 1e3, a.LastKernelName()), flush=True)
            a.close()
    print("%s: records identical between the two walks: %s" % (name, outs[0] == outs[1]), flush=True)
