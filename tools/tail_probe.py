"""How much of a launch is start-up/drain tail?  Times the C3 waves at 1x, 2x and 4x the ray count (the ray
array repeated) — with no tail the time per ray would not change."""
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
for name, rays in (("primary", rays1), ("bounce", rays2)):
    for rep in (1, 2, 4):
        r = np.concatenate([rays] * rep)
        d = torch.from_numpy(r.view(np.uint8)).cuda(); o = torch.empty(len(r) * 16, dtype=torch.uint8, device='cuda')
        ts = []
        for _ in range(7):
            a.TraverseBatchDevice(d, o); ts.append(a.LastTraverseMs())
        ms = float(np.median(ts))
        print("%s x%d: %.3f ms, %.1f Mrays/s" % (name, rep, ms, len(r) / ms / 1e3), flush=True)
both = np.concatenate([rays1, rays2])
for tag, r in (("primary+bounce, one launch", both), ("interleaved rows of both", np.concatenate([np.stack([rays1[:len(rays2)], rays2], axis=1).reshape(-1), rays1[len(rays2):]]))):
    d = torch.from_numpy(np.ascontiguousarray(r).view(np.uint8)).cuda(); o = torch.empty(len(r) * 16, dtype=torch.uint8, device='cuda')
    for env in ({}, {"NRT_STATIC_PCT": "50"}, {"NRT_STATIC_PCT": "0"}, {"NRT_CHUNK": "128"}):
        os.environ.update(env)
        c = BVHAccel(np.float32); c.Build(mesh.num_faces, mesh)
        ts = []
        for _ in range(7):
            c.TraverseBatchDevice(d, o); ts.append(c.LastTraverseMs())
        ms = float(np.median(ts))
        print("%s %s: %.3f ms, %.1f Mrays/s" % (tag, env, ms, len(r) / ms / 1e3), flush=True)
        for k in env: del os.environ[k]
