import os, sys, ctypes
import numpy as np
sys.path.insert(0, '.')
os.environ['NRT_DEBUG'] = '32'
for a in sys.argv[1:]:
    k, v = a.split('='); os.environ[k] = v
import torch
from nanort_amd import BVHAccel, TriangleMesh, scenes, capi
v, f = scenes.plane(1000, 500); mesh = TriangleMesh(v, f)
rays1 = scenes.camera_rays(1920, 1080)
a = BVHAccel(np.float32); a.Build(mesh.num_faces, mesh)
h1, m1 = a.TraverseBatch(rays1)
rays2 = scenes.secondary_rays("bounce", v, f, rays1, h1, m1)
L = capi.lib(); L.nrtDebugCounters.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
for name, rays in (('primary', rays1), ('bounce', rays2)):
    d = torch.from_numpy(rays.view(np.uint8)).cuda(); o = torch.empty(len(rays) * 16, dtype=torch.uint8, device='cuda')
    a.TraverseBatchDevice(d, o); ms = a.LastTraverseMs()
    c = np.zeros(8, dtype=np.uint64); L.nrtDebugCounters(a._h, c.ctypes.data_as(ctypes.c_void_p))
    it1, act1, idle2, it2, act2, refills, refilled, ent2 = [int(x) for x in c[:8]]
    n = len(rays)
    print(name, 'ms %.3f' % ms, 'phase1: wave-iters/ray-group %.1f' % (it1 / (n / 64)), 'avg active lanes %.1f' % (act1 / it1),
          '| phase2: wave-iters/ray-group %.1f avg active %.1f' % (it2 / (n / 64), act2 / max(1, it2)), 'entries/group %.2f idle lanes at entry %.1f' % (ent2 / (n / 64), idle2 / max(1, ent2)), '| refills/group %.2f lanes/refill %.1f' % (refills / (n / 64), refilled / max(1, refills)))
