"""TEST INFRASTRUCTURE.  Generates tests/golden/embree_ref.npz: what the reference's Embree-2 shim returns for the
scene and rays of tests/embree_fixture.py.

The reference's shim (examples/embree-api/nanort-embree.cc) does not compile at this revision (its call at :326 lacks
the template arguments nanosg.h:779-781 requires), so it cannot be run.  What it does is a thin mapping around
nanosg::Scene::Traverse, and this script executes exactly that with the UNMODIFIED nanosg.h + nanort.h
(oracle/_ref/libnanosg_ref.so):

  scene   :321-332  every triangle mesh becomes a root node (identity transform), in geometry-id order
  ray     :518-533  org, dir, min_t = tnear, max_t = tfar
  query   :535-539  Scene::Traverse(ray, &isect, cull_back_face = false)
  hit     :541-548  tfar = t, u, v, geomID = node_id, primID = prim_id, instID = INVALID
  miss    :549-553  geomID = primID = instID = INVALID; nothing else written
  bounds  :505-513  Scene::GetBoundingBox
  ids     :226-246, :560-598  rtcNewTriangleMesh returns 1, 2, 3, ...

Run in the build container (needs /root/reference through oracle/_ref):  python oracle/gen_golden_embree.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from embree_fixture import INVALID, meshes, rays  # noqa: E402

from nanort_amd.wire import RAY_F32  # noqa: E402
from oracle.bindings import SceneReference  # noqa: E402


def main():
    ms = meshes()
    r = rays()
    ref = SceneReference()
    ident = np.eye(4, dtype=np.float32)
    for v, f in ms:
        ref.add_node(v, f, ident)
    assert ref.commit()
    nr = np.zeros((r.shape[0],), dtype=RAY_F32)
    nr["org"] = r[:, 0:3]
    nr["dir"] = r[:, 3:6]
    nr["min_t"] = r[:, 6]
    nr["max_t"] = r[:, 7]
    hits, mask = ref.traverse(nr, cull_back_face=False)
    hit = mask != 0
    out = {
        "hit": mask,
        "tfar": np.where(hit, hits["t"], r[:, 7]).astype(np.float32),
        "u": hits["u"][hit],
        "v": hits["v"][hit],
        "geomID": np.where(hit, hits["node_id"], INVALID).astype(np.uint32),
        "primID": np.where(hit, hits["prim_id"], INVALID).astype(np.uint32),
        "ids": np.arange(1, len(ms) + 1, dtype=np.uint32),
    }
    bmin, bmax = ref.bounds()
    out["bounds"] = np.array([bmin[0], bmin[1], bmin[2], 0, bmax[0], bmax[1], bmax[2], 0], dtype=np.float32)
    path = os.path.join(ROOT, "tests", "golden", "embree_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d rays, %d hits, geomIDs hit %s, bounds %s" % (
        path, r.shape[0], int(hit.sum()), np.unique(out["geomID"][hit]).tolist(), out["bounds"].tolist()))


if __name__ == "__main__":
    main()
