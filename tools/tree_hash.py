"""Fingerprints of GPU-built trees (node array + index permutation) over a spread of meshes, precisions and build options:
a builder refactor that is meant to change nothing must reproduce every line.  Usage: python tools/tree_hash.py"""
import hashlib, sys
import numpy as np
sys.path.insert(0, '.')
from nanort_amd import BVHAccel, TriangleMesh, SphereGeometry, scenes
from nanort_amd.wire import default_build_options
rng = np.random.default_rng(3)
cases = [("plane1M", scenes.plane(1000, 500)), ("sphere70k", scenes.sphere()), ("plane300x150", scenes.plane(300, 150))]
v = rng.normal(size=(20000, 3)).astype(np.float32); f = rng.integers(0, 20000, size=(60000, 3)).astype(np.uint32)
cases.append(("soup60k", (v, f)))
g = rng.integers(-6, 7, size=(3000, 3)).astype(np.float32); gf = rng.integers(0, 3000, size=(9000, 3)).astype(np.uint32)
cases.append(("grid9k", (g, gf)))
for name, (v, f) in cases:
    for real in (np.float32, np.float64):
        for ml, bins, md in ((4, 64, 256), (1, 8, 256), (16, 200, 12)):
            if name == "plane1M" and (real is np.float64 or ml != 4):
                continue
            bo = default_build_options(real); bo["min_leaf_primitives"] = ml; bo["bin_size"] = bins; bo["max_tree_depth"] = md
            a = BVHAccel(real); assert a.Build(f.shape[0], TriangleMesh(v.astype(real), f), bo)
            nodes, idx = a.GetTree()
            print(name, real.__name__, ml, bins, md, nodes.shape[0], hashlib.md5(nodes.tobytes() + idx.tobytes()).hexdigest(), flush=True)
c, r = scenes.random_spheres(50000)
a = BVHAccel(np.float32); assert a.Build(50000, SphereGeometry(c, r)); nodes, idx = a.GetTree()
print("spheres50k", nodes.shape[0], hashlib.md5(nodes.tobytes() + idx.tobytes()).hexdigest())
