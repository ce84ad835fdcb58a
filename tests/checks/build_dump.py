"""Child process of tests/test_gpu_build.py: build one mesh with given options / tunables on the PROFILING build of the library
(libnanort_hip_prof.so: the only one that carries the builder's one-node-per-step subtree kernel, tunable subtree_rows = 0) and
dump the tree.   python tests/checks/build_dump.py in.npz out.npz"""
import json
import os
import sys

os.environ["NRT_USE_PROF_LIB"] = "1"
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nanort_amd import BVHAccel, TriangleMesh  # noqa: E402
from nanort_amd.wire import default_build_options  # noqa: E402

d = np.load(sys.argv[1], allow_pickle=False)
real = np.float32 if str(d["real"]) == "float32" else np.float64
v, f = d["v"].astype(real), d["f"]
a = BVHAccel(real)
for k, val in json.loads(str(d["tunables"])).items():
    a.SetTunable(k, val)
o = default_build_options(real)
for k, val in json.loads(str(d["options"])).items():
    o[k] = val
assert a.Build(f.shape[0], TriangleMesh(v, f), o)
nodes, idx = a.GetTree()
st = a.GetStatistics()
np.savez(sys.argv[2], nodes=np.frombuffer(nodes.tobytes(), dtype=np.uint8), idx=idx,
         stats=json.dumps({k: np.asarray(st[k]).reshape(-1)[0].item() for k in st.dtype.names}))
