import sys, time, numpy as np
sys.path.insert(0, '.')
from nanort_amd import BVHAccel, TriangleMesh, scenes
from oracle.bindings import Oracle
orc = Oracle()
for name, (v, f), (W, H) in [('c1', scenes.load_c1_mesh(), (256, 256)), ('plane100x50', scenes.plane(100, 50), (512, 288))]:
    rays = scenes.camera_rays(W, H)
    nodes, idx, st = orc.build(v, f)
    oh, om, cnt = orc.traverse(nodes, idx, v, f, rays, count=True)
    a = BVHAccel(np.float32)
    a.SetMesh(TriangleMesh(v, f))
    a.SetTree(nodes, idx)
    h, m = a.TraverseBatch(rays)
    print(name, 'mask eq', (m == om).all(), 'hits eq', h.tobytes() == oh.tobytes(), 'kernel ms', a.LastTraverseMs(), 'oracle counters', cnt)
    if h.tobytes() != oh.tobytes():
        bad = np.nonzero((h['t'] != oh['t']) | (h['prim_id'] != oh['prim_id']) | (h['u'] != oh['u']) | (h['v'] != oh['v']))[0]
        print(' mismatches', len(bad), bad[:10]); print(h[bad[:5]], oh[bad[:5]])
    import torch
    d_rays = torch.from_numpy(rays.view(np.uint8)).cuda()
    print(' gpu counters', a.TraverseCountDevice(d_rays))
