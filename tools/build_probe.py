import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from nanort_amd import BVHAccel, TriangleMesh, scenes
from bvh_check import validate_bvh
for name, (v, f) in [('plane1M', scenes.plane(1000, 500)), ('sphere70k', scenes.sphere())]:
    a = BVHAccel(np.float32); m = TriangleMesh(v, f)
    ts = []
    for _ in range(6):
        a.Build(m.num_faces, m); ts.append(a.LastBuildMs())
    nodes, idx = a.GetTree()
    r = validate_bvh(nodes, idx, v, f, stats=a.GetStatistics())
    print(name, 'build ms', ['%.3f' % t for t in ts], 'sah %.3f nodes %d depth %d' % (r['sah_cost'], r['num_nodes'], r['max_depth']), flush=True)
