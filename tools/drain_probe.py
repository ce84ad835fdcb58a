"""When do the waves of a traversal launch run dry and finish?  (NRT_DEBUG bit 8192: per-wave realtime stamps,
100 MHz.)  Prints, for the C3 primary and bounce waves: launch span, when the first / median / last wave ran out of rays,
mean and longest drain (out of rays -> done), and how many lane-microseconds the drain wastes."""
import ctypes, os, sys
import numpy as np
os.environ["NRT_DEBUG"] = str(int(os.environ.get("NRT_DEBUG", "0")) | 8192)
sys.path.insert(0, '.')
import torch
from nanort_amd import BVHAccel, TriangleMesh, scenes

v, f = scenes.plane(1000, 500); mesh = TriangleMesh(v, f)
rays1 = scenes.camera_rays(1920, 1080)
a = BVHAccel(np.float32); a.Build(mesh.num_faces, mesh)
h1, m1 = a.TraverseBatch(rays1)
rays2 = scenes.secondary_rays("bounce", v, f, rays1, h1, m1)
a._L.nrtDebugWaveClocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
a._L.nrtDebugWaveClocks.restype = ctypes.c_long
for name, rays in (("primary", rays1), ("bounce", rays2)):
    d = torch.from_numpy(rays.view(np.uint8)).cuda(); o = torch.empty(len(rays) * 16, dtype=torch.uint8, device='cuda')
    for rep in range(3):
        a.TraverseBatchDevice(d, o); torch.cuda.synchronize()
        buf = np.zeros((8192, 3), dtype=np.uint64)
        n = a._L.nrtDebugWaveClocks(a._h, buf.ctypes.data_as(ctypes.c_void_p), 8192)
        c = buf[:n].astype(np.float64) / 100.0  # microseconds
        t0 = c[:, 0].min()
        dry, end = c[:, 1] - t0, c[:, 2] - t0
        q = lambda x, p: float(np.percentile(x, p))
        print("%s (%s): span %.1f us (event %.1f) | start spread %.1f | dry: first %.1f p10 %.1f median %.1f p90 %.1f last %.1f | done: p10 %.1f median %.1f p90 %.1f last %.1f | drain mean %.1f longest %.1f | idle wave-time before the end %.1f%%" % (
            name, a.LastKernelName()[-19:], end.max(), a.LastTraverseMs() * 1e3, (c[:, 0] - t0).max(), dry.min(), q(dry, 10), q(dry, 50), q(dry, 90), dry.max(),
            q(end, 10), q(end, 50), q(end, 90), end.max(), (end - dry).mean(), (end - dry).max(), 100.0 * (end.max() - end).sum() / (end.max() * n)), flush=True)
