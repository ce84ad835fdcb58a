"""C3 waves under different BVHBuildOptions::min_leaf_primitives (the reference's default is 4): build ms, tree size,
node visits / triangle tests per ray, kernel ms.  A tuning probe: hit records do not depend on the leaf size."""
import sys, hashlib
import numpy as np
sys.path.insert(0, '.')
import torch
from nanort_amd import BVHAccel, TriangleMesh, scenes
from nanort_amd.wire import default_build_options

v, f = scenes.plane(1000, 500); mesh = TriangleMesh(v, f)
rays1 = scenes.camera_rays(1920, 1080)
base = BVHAccel(np.float32); base.Build(mesh.num_faces, mesh)
h1, m1 = base.TraverseBatch(rays1)
rays2 = scenes.secondary_rays("bounce", v, f, rays1, h1, m1)
d1 = torch.from_numpy(rays1.view(np.uint8)).cuda(); d2 = torch.from_numpy(rays2.view(np.uint8)).cuda()
o1 = torch.empty(len(rays1) * 16, dtype=torch.uint8, device='cuda'); o2 = torch.empty(len(rays2) * 16, dtype=torch.uint8, device='cuda')
ref_t = None
for ml in [int(x) for x in (sys.argv[1:] or ["1", "2", "3", "4", "6", "8"])]:
    o = default_build_options(np.float32); o["min_leaf_primitives"] = ml
    a = BVHAccel(np.float32)
    bms = []
    for _ in range(3):
        a.Build(mesh.num_faces, mesh, o); bms.append(a.LastBuildMs())
    st = a.GetStatistics()
    t1, t2 = [], []
    for _ in range(7):
        a.TraverseBatchDevice(d1, o1); t1.append(a.LastTraverseMs())
        a.TraverseBatchDevice(d2, o2); t2.append(a.LastTraverseMs())
    c1 = a.TraverseCountDevice(d1); c2 = a.TraverseCountDevice(d2)
    tt = hashlib.md5(np.ascontiguousarray(o1.cpu().numpy().view(np.float32).reshape(-1, 4)[:, 2]).tobytes()).hexdigest()
    ref_t = ref_t or tt
    print("min_leaf %d: build %.3f ms, %d nodes depth %d | primary %.3f ms (%.1f nodes %.2f tris per ray) bounce %.3f ms (%.1f, %.2f) sum %.3f | same t: %s" % (
        ml, float(np.median(bms)), int(st["num_leaf_nodes"] + st["num_branch_nodes"]), int(st["max_tree_depth"]), float(np.median(t1)),
        c1["nodes_visited"] / len(rays1), c1["tris_tested"] / len(rays1), float(np.median(t2)), c2["nodes_visited"] / len(rays2),
        c2["tris_tested"] / len(rays2), float(np.median(t1)) + float(np.median(t2)), tt == ref_t), flush=True)
