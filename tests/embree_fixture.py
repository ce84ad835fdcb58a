"""The scene and ray set shared by the Embree-API tests and oracle/gen_golden_embree.py: the five nodes of
tests/scene_fixture.py with their transforms baked into the vertices (the Embree shim has no instancing, reference
examples/embree-api/nanort-embree.cc:648-680), serialised the way tests/cpp/embree_check.cc reads it."""
import struct

import numpy as np

from nanort_amd import scenes

from scene_fixture import instances

RTC_RAY = np.dtype(
    [("org", "<f4", 3), ("align0", "<f4"), ("dir", "<f4", 3), ("align1", "<f4"), ("tnear", "<f4"), ("tfar", "<f4"),
     ("time", "<f4"), ("mask", "<u4"), ("Ng", "<f4", 3), ("align2", "<f4"), ("u", "<f4"), ("v", "<f4"),
     ("geomID", "<u4"), ("primID", "<u4"), ("instID", "<u4"), ("pad", "<u4", 3)]
)
assert RTC_RAY.itemsize == 96
INVALID = 0xFFFFFFFF


def meshes():
    out = []
    for verts, faces, xf in instances():
        v = verts.astype(np.float32)
        w = np.empty_like(v)  # nanosg's row-vector convention: p' = p . M[:3,:3] + M[3,:3], in float32
        for k in range(3):
            w[:, k] = v[:, 0] * xf[0, k] + v[:, 1] * xf[1, k] + v[:, 2] * xf[2, k] + xf[3, k]
        out.append((np.ascontiguousarray(w, dtype=np.float32), np.ascontiguousarray(faces, dtype=np.uint32)))
    return out


def scene_bytes(ms=None):
    ms = meshes() if ms is None else ms
    b = [struct.pack("<I", len(ms))]
    for v, f in ms:
        b.append(struct.pack("<II", v.shape[0], f.shape[0]))
        b.append(v.tobytes())
        b.append(f.tobytes())
    return b"".join(b)


def rays(seed=11):
    """Camera rays over the scene + random segments with finite tfar / positive tnear (the interval only gates the
    node boxes in nanosg's traversal, which the test must see reproduced)."""
    cam = scenes.camera_rays(160, 90)
    n = cam.shape[0]
    r = np.zeros((n + 6000, 8), dtype=np.float32)
    r[:n, 0:3] = cam["org"]
    r[:n, 3:6] = cam["dir"]
    r[:n, 6] = 0.0
    r[:n, 7] = 1.0e30
    rng = np.random.default_rng(seed)
    m = 6000
    org = rng.uniform([-8, 0, -2], [8, 12, 14], size=(m, 3))
    tgt = rng.uniform([-6, 0, 0], [6, 10, 8], size=(m, 3))
    d = tgt - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r[n:, 0:3] = org
    r[n:, 3:6] = d
    r[n:, 6] = np.where(rng.random(m) < 0.3, rng.uniform(0, 6, m), 0.0)
    r[n:, 7] = np.where(rng.random(m) < 0.5, rng.uniform(1, 20, m), 1.0e30)
    return np.ascontiguousarray(r)


def rays_bytes(r):
    return struct.pack("<Q", r.shape[0]) + r.tobytes()


def parse_out(blob, num_meshes, num_rays):
    bounds = np.frombuffer(blob, dtype="<f4", count=8)
    ids = np.frombuffer(blob, dtype="<u4", count=num_meshes, offset=32)
    out = np.frombuffer(blob, dtype=RTC_RAY, count=num_rays, offset=32 + 4 * num_meshes)
    return bounds, ids, out
