import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from nanort_amd import BVHAccel, TriangleMesh, scenes
from oracle.bindings import Oracle
from bvh_check import validate_bvh, sah_cost
import torch
orc = Oracle()
cases = [('c1', scenes.load_c1_mesh(), (256, 256)), ('plane100x50', scenes.plane(100, 50), (512, 288)),
         ('sphere', scenes.sphere(), (640, 360)), ('plane1000x500', scenes.plane(1000, 500), (1920, 1080))]
if len(sys.argv) > 1: cases = [c for c in cases if c[0] in sys.argv[1:]]
for name, (v, f), (W, H) in cases:
    rays = scenes.camera_rays(W, H)
    a = BVHAccel(np.float32)
    t0 = time.time(); ok = a.Build(len(f), TriangleMesh(v, f)); t1 = time.time()
    st = a.GetStatistics()
    print(name, 'build ok', ok, 'ms(dev)', a.LastBuildMs(), 'wall ms', (t1-t0)*1e3, st)
    for _ in range(3):
        a.Build(len(f), TriangleMesh(v, f)); print('   rebuild ms', a.LastBuildMs())
    nodes, idx = a.GetTree()
    m = validate_bvh(nodes, idx, v, f, stats=st)
    print('   valid', m)
    h, mk = a.TraverseBatch(rays)
    print('   traverse kernel ms', a.LastTraverseMs(), 'Mrays/s', len(rays)/a.LastTraverseMs()/1e3)
    d_rays = torch.from_numpy(rays.view(np.uint8)).cuda()
    print('   counters', a.TraverseCountDevice(d_rays))
    if len(f) <= 100000:
        onodes, oidx, ost = orc.build(v, f)
        print('   ref tree: nodes', len(onodes), 'depth', ost['max_tree_depth'], 'sah', sah_cost(onodes))
        oh, om = orc.traverse(onodes, oidx, v, f, rays)
        eq = h.tobytes() == oh.tobytes() and (mk == om).all()
        print('   hits == oracle(ref tree):', eq)
        if not eq:
            bad = np.nonzero((h['t'] != oh['t']) | (h['prim_id'] != oh['prim_id']) | (h['u'] != oh['u']) | (h['v'] != oh['v']))[0]
            print('   mismatches', len(bad), bad[:8], h[bad[:4]], oh[bad[:4]])
        oh2, om2 = orc.traverse(nodes, idx, v, f, rays)
        print('   hits == oracle(gpu tree):', h.tobytes() == oh2.tobytes())
    else:
        g = np.load('tests/golden/c3_sample.npz'); s = int(g['stride'])
        print('   hits == golden sample:', h[::s].tobytes() == g['hits'].tobytes(), (mk[::s] == g['mask']).all())
