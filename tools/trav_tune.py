"""Sweep the traversal kernel's tunables on C3 (primary + bounce waves); prints kernel ms per config."""
import itertools, os, sys, hashlib
import numpy as np
sys.path.insert(0, '.')
import torch
from nanort_amd import BVHAccel, TriangleMesh, scenes
from nanort_amd.wire import HIT_F32

v, f = scenes.plane(1000, 500)
mesh = TriangleMesh(v, f)
rays1 = scenes.camera_rays(1920, 1080)
base = BVHAccel(np.float32); base.Build(mesh.num_faces, mesh)
h1, m1 = base.TraverseBatch(rays1)
rays2 = scenes.secondary_rays("bounce", v, f, rays1, h1, m1)
d1 = torch.from_numpy(rays1.view(np.uint8)).cuda(); d2 = torch.from_numpy(rays2.view(np.uint8)).cuda()
o1 = torch.empty(len(rays1) * 16, dtype=torch.uint8, device='cuda'); o2 = torch.empty(len(rays2) * 16, dtype=torch.uint8, device='cuda')
ref = None
def run(cfg, reps=5):
    global ref
    for k, val in cfg.items(): os.environ[k] = str(val)
    a = BVHAccel(np.float32); a.Build(mesh.num_faces, mesh)
    t1, t2 = [], []
    for _ in range(reps):
        a.TraverseBatchDevice(d1, o1); t1.append(a.LastTraverseMs())
        a.TraverseBatchDevice(d2, o2); t2.append(a.LastTraverseMs())
    hsh = hashlib.md5(o1.cpu().numpy().tobytes() + o2.cpu().numpy().tobytes()).hexdigest()
    if ref is None: ref = hsh
    return float(np.median(t1)), float(np.median(t2)), hsh == ref
grid = sys.argv[1] if len(sys.argv) > 1 else 'coarse'
if grid == 'coarse':
    combos = [dict(NRT_LDS_STACK=s, NRT_REFILL_MIN=r, NRT_TRAV_MIN=t) for s in (32, 24, 16) for r in (1, 16, 32, 48) for t in (1, 8, 16, 32, 48)]
else:
    combos = [eval(x) for x in sys.argv[1:]]
for c in combos:
    a, b, ok = run(c)
    print("%s  primary %.3f ms  bounce %.3f ms  sum %.3f  same=%s" % (c, a, b, a + b, ok), flush=True)
