"""Per-ray cost distribution of the C3 waves (STATS kernel, NRT_DEBUG=96: u <- WideNode steps, v <- triangle tests)."""
import os, sys
import numpy as np
sys.path.insert(0, '.')
import torch
from nanort_amd import BVHAccel, TriangleMesh, scenes
from nanort_amd.wire import HIT_F32
v, f = scenes.plane(1000, 500); mesh = TriangleMesh(v, f)
rays1 = scenes.camera_rays(1920, 1080)
a = BVHAccel(np.float32); a.Build(mesh.num_faces, mesh)
h1, m1 = a.TraverseBatch(rays1)
rays2 = scenes.secondary_rays("bounce", v, f, rays1, h1, m1)
os.environ['NRT_DEBUG'] = '96'
b = BVHAccel(np.float32); b.Build(mesh.num_faces, mesh)
for name, rays in (("primary", rays1), ("bounce", rays2)):
    h, m = b.TraverseBatch(rays)
    steps = h['u']; tris = h['v']
    q = [50, 90, 99, 99.9, 99.99, 100]
    print(name, "steps mean %.1f" % steps.mean(), "percentiles", dict(zip(q, np.percentile(steps, q).round(0))),
          "| tris mean %.2f max %d" % (tris.mean(), tris.max()), flush=True)
    if name == "bounce":
        dz = np.abs(rays['dir'][:, 2]) / np.linalg.norm(rays['dir'], axis=1)
        order = np.argsort(-steps)[:10]
        print(" top-10 rays: steps", steps[order], "|dir.z|", dz[order].round(3))
        print(" corr(steps, 1/|dz|) = %.3f" % np.corrcoef(steps, 1.0 / np.maximum(dz, 1e-3))[0, 1])
