"""Replay a case saved by fuzz_parity.py (gpurun_out/fuzz_fail_*.npz or tests/golden/fuzz_case_*.npz) on the GPU over
the SAME node array and print every ray whose record differs from the restatement's.
Usage: python tests/checks/replay_fuzz_case.py case.npz"""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from nanort_amd import BVHAccel, TriangleMesh
from oracle.bindings import Oracle

d = np.load(sys.argv[1])
v, f, rays, opts, nodes, idx = d['v'], d['f'], d['rays'], d['opts'], d['nodes'], d['idx']
a = BVHAccel(v.dtype.type)
a.SetMesh(TriangleMesh(v, f)); a.SetTree(nodes, idx)
h, m = a.TraverseBatch(rays, opts)
oh, om = Oracle().traverse(nodes, idx, v, f, rays, opts)
bad = np.nonzero((m != om) | (h['t'].view(np.uint32 if v.dtype == np.float32 else np.uint64) != oh['t'].view(np.uint32 if v.dtype == np.float32 else np.uint64))
                 | (h['prim_id'] != oh['prim_id']) | (h['u'] != oh['u']) | (h['v'] != oh['v']))[0]
print("kernel", a.LastKernelName(), "rays", len(rays), "differing", len(bad))
for i in bad[:12]:
    print(i, "dir", rays[i]['dir'], "gpu", m[i], h[i], "oracle", om[i], oh[i])
