import sys, os, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from nanort_amd import BVHAccel, TriangleMesh, scenes
from bvh_check import validate_bvh
import torch
for name, (v, f) in [('plane1M', scenes.plane(1000, 500)), ('sphere70k', scenes.sphere())]:
    rays = scenes.camera_rays(1920, 1080)
    d = torch.from_numpy(rays.view(np.uint8)).cuda(); o = torch.empty(len(rays) * 16, dtype=torch.uint8, device='cuda')
    for mort in (0, 1):
        os.environ['NRT_MORTON'] = str(mort)
        a = BVHAccel(np.float32); m = TriangleMesh(v, f)
        ts = []
        for _ in range(6):
            a.Build(m.num_faces, m); ts.append(a.LastBuildMs())
        nodes, idx = a.GetTree()
        r = validate_bvh(nodes, idx, v, f, stats=a.GetStatistics())
        tt = []
        for _ in range(5):
            a.TraverseBatchDevice(d, o); tt.append(a.LastTraverseMs())
        print(name, 'morton', mort, 'build ms', ['%.3f' % t for t in ts], 'sah %.3f nodes %d depth %d' % (r['sah_cost'], r['num_nodes'], r['max_depth']), 'trace ms %.3f' % np.median(tt), flush=True)
