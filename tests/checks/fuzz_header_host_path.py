"""CPU-only soak of include/nanort.h's HOST path (no GPU backend): its own builder + the per-ray Traverse() with the
restated TriangleIntersector, driven through tests/cpp/host_check.cc, against the CPU restatement of the reference
walking the header's own node array — every record bit for bit — on the hostile generators of fuzz_parity.py (integer
grids, flat sheets, duplicated / degenerate triangles; axis-parallel, zero, NaN and infinite ray components), fp32 and
fp64; the tree's structure is validated as well.  Usage: python tests/checks/fuzz_header_host_path.py [seconds] [seed]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bvh_check import validate_bvh  # noqa: E402
from nanort_amd.wire import HIT_F32, HIT_F64, NODE_F32, NODE_F64, ray_dtype  # noqa: E402
from oracle.bindings import Oracle  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
orc = Oracle()
d = tempfile.mkdtemp()
exe = os.path.join(d, "host_check")
subprocess.run(["g++", "-std=c++11", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "host_check.cc"), "-o", exe], check=True)
t_end = time.time() + budget
rounds = total = 0
while time.time() < t_end:
    f64 = rng.random() < 0.4
    real = np.float64 if f64 else np.float32
    n = int(rng.choice([1, 2, 3, 5, 17, 64, 257, 1500, 6000]))
    kind = rng.integers(0, 3)
    nv = max(3, n)
    if kind == 0:
        v = rng.integers(-4, 5, size=(nv, 3)).astype(real)
    elif kind == 1:
        v = rng.normal(size=(nv, 3)).astype(real) * 3
    else:
        v = np.column_stack([rng.uniform(-5, 5, nv), rng.uniform(-5, 5, nv), rng.integers(-2, 3, nv)]).astype(real)
    f = rng.integers(0, v.shape[0], size=(n, 3)).astype(np.uint32)
    if n > 10:
        f[: n // 10] = f[n // 10: 2 * (n // 10)]
        f[-1] = f[-1][[0, 0, 1]]
    m = 2000
    rays = np.zeros(m, dtype=ray_dtype(real))
    rays["org"] = rng.integers(-6, 7, size=(m, 3)).astype(real) if rng.random() < 0.5 else rng.normal(size=(m, 3)).astype(real) * 6
    tgt = v[rng.integers(0, v.shape[0], m)] + (rng.integers(-1, 2, size=(m, 3)) * (rng.random((m, 1)) < 0.3)).astype(real)
    dd = tgt - rays["org"]
    dd[: m // 8] = rng.integers(-1, 2, size=(m // 8, 3))
    dd[m // 8: m // 8 + 20, 0] = np.nan
    dd[m // 8 + 20: m // 8 + 40, 1] = np.inf
    rays["dir"] = dd.astype(real)
    rays["min_t"] = rng.choice([0.0, 0.0, 1e-3, 0.5], m).astype(real)
    rays["max_t"] = rng.choice([1e30, 1e30, 2.0, 1.0, 0.0, -1.0], m).astype(real)
    mesh, rp, out = os.path.join(d, "mesh.bin"), os.path.join(d, "rays.bin"), os.path.join(d, "out.bin")
    with open(mesh, "wb") as fp:
        fp.write(np.array([v.shape[0], f.shape[0]], dtype=np.uint32).tobytes() + np.ascontiguousarray(v).tobytes() + f.tobytes())
    with open(rp, "wb") as fp:
        fp.write(np.array([m], dtype=np.uint64).tobytes() + rays.tobytes())
    r = subprocess.run([exe, "trace", "f64" if f64 else "f32", mesh, rp, out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    tag = "round %d seed %d %s n %d kind %d" % (rounds, seed, real.__name__, n, kind)
    assert r.returncode == 0, tag + ": " + r.stdout[-500:]
    hd, nd = (HIT_F64, NODE_F64) if f64 else (HIT_F32, NODE_F32)
    raw = open(out, "rb").read()
    hits = np.frombuffer(raw, dtype=hd, count=m)
    o = m * hd.itemsize
    mask = np.frombuffer(raw, dtype=np.uint8, count=m, offset=o)
    o += m
    nn = int(np.frombuffer(raw, dtype=np.uint64, count=1, offset=o)[0])
    o += 8
    nodes = np.frombuffer(raw, dtype=nd, count=nn, offset=o)
    idx = np.frombuffer(raw, dtype=np.uint32, count=n, offset=o + nn * nd.itemsize)
    if kind == 1:
        validate_bvh(nodes, idx, v, f)
    assert sorted(idx.tolist()) == list(range(n)), "indices are not a permutation: " + tag
    oh, om = orc.traverse(nodes, idx, v, f, rays)
    assert np.array_equal(mask, om), "hit flags differ: " + tag
    hit = om != 0
    for k in ("t", "u", "v", "prim_id"):
        assert hits[k][hit].tobytes() == oh[k][hit].tobytes(), "%s differs: %s" % (k, tag)
    rounds += 1
    total += m
print("header host path == restatement: %d rounds, %d rays, seed %d" % (rounds, total, seed))
