#!/usr/bin/env python3
"""Generate tests/golden/* by RUNNING THE UNMODIFIED REFERENCE — test infrastructure.

Runs only where /root/reference exists (the build container):

    make -C oracle ref && python oracle/gen_golden.py

It drives oracle/_ref/libnanort_ref.so (the reference header behind a C shim)
and oracle/_ref/obj2mesh (the reference example's own OBJ loader) and writes
small fixtures that travel with the repo, so that the oracle (liboracle.so) and
the HIP path can be pinned on machines where the reference is absent.

Fixtures (all little-endian, numpy .npz / .json):
  c1_mesh.npz        Cornell box + Suzanne flattened as examples/objrender does
  c1_ref.npz         reference tree (serial build) + full hit records, 256x256
                     camera wave, fp32 and fp64; trace-option variants
  c1_wave2.npz       reference hits for the shadow / bounce waves built from
                     the 256x256 primaries
  c3_sample.npz      Plane(1000,500) 1920x1080: every 13th primary ray's hit
  sphere_sample.npz  lumpy sphere (C2 stand-in) 1920x1080: every 13th ray
  known_answers.json KA1-KA4 of SURVEY.md §8c as measured here + checksums
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from nanort_amd import scenes  # noqa: E402
from nanort_amd.wire import RAY_F64, default_trace_options, widen_rays  # noqa: E402
from oracle.bindings import Reference  # noqa: E402

REFERENCE = os.environ.get("REFERENCE", "/root/reference")
GOLDEN = os.path.join(ROOT, "tests", "golden")
STRIDE = 13


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def masked_nodes(nodes):
    n = nodes.copy()
    n["axis"][n["flag"] == 1] = 0  # leaf .axis is uninitialised in the reference (nanort.h:501)
    return n


def sums(hits, mask):
    m = mask == 1
    return {
        "num_hits": int(m.sum()),
        "sum_t": float(hits["t"][m].astype(np.float64).sum()),
        "sum_u": float(hits["u"][m].astype(np.float64).sum()),
        "sum_v": float(hits["v"][m].astype(np.float64).sum()),
        "sha256_hits": sha(hits),
        "sha256_prim_id": sha(hits["prim_id"]),
    }


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    ka = {"_generator": "oracle/gen_golden.py", "_source": "unmodified reference via oracle/_ref"}

    # ---- C1 mesh through the example's own loader --------------------------
    objdir = os.path.join(REFERENCE, "examples", "objrender")
    tmp = "/tmp/_c1_mesh.bin"
    subprocess.check_call([os.path.join(HERE, "_ref", "obj2mesh"), "cornellbox_suzanne.obj", tmp], cwd=objdir)
    raw = open(tmp, "rb").read()
    nv, nf = np.frombuffer(raw[:8], dtype=np.uint32)
    verts = np.frombuffer(raw[8 : 8 + 12 * nv], dtype=np.float32).reshape(-1, 3).copy()
    faces = np.frombuffer(raw[8 + 12 * nv :], dtype=np.uint32).reshape(-1, 3).copy()
    np.savez_compressed(os.path.join(GOLDEN, "c1_mesh.npz"), vertices=verts, faces=faces)

    # ---- KA1: regression #30 (fp64, one triangle, cache_bbox) --------------
    v30 = np.array([[1.0, 2.0, -3.0], [-1.0, 2.0, -3.0], [1.0, 2.0, 3.0]], dtype=np.float64)
    f30 = np.array([[0, 1, 2]], dtype=np.uint32)
    ka["KA1"] = {}
    for label, d0 in (("normal", 0.0), ("tiny_dir0", -5.30287619e-17)):
        d = np.array([d0, -8.66025404e-01, -0.5])
        d = d / np.sqrt((d * d).sum())
        ray = np.zeros((1,), dtype=RAY_F64)
        ray["org"] = (-0.36, 7.93890843, 1.2160368)
        ray["dir"] = d
        ray["min_t"] = 0.0
        ray["max_t"] = 1.0e30
        R = Reference(v30, f30)
        ok, st = R.build(cache_bbox=True)
        h, m, _ = R.traverse(ray)
        ka["KA1"][label] = {
            "dir": [float(x) for x in d],
            "hit": int(m[0]),
            "u": float(h["u"][0]),
            "v": float(h["v"][0]),
            "t": float(h["t"][0]),
            "prim_id": int(h["prim_id"][0]),
        }

    # ---- KA2 / KA3: C1 -----------------------------------------------------
    c1 = {}
    R = Reference(verts, faces)
    ok, st = R.build(parallel=False)
    nodes, indices = R.tree()
    bmin, bmax = R.bounding_box()
    ka["KA2"] = {
        "stats": {k: int(v) for k, v in st.items() if k != "build_secs"},
        "num_nodes": int(nodes.shape[0]),
        "bbox_min": [float(x) for x in bmin],
        "bbox_max": [float(x) for x in bmax],
        "sha256_nodes_masked": sha(masked_nodes(nodes)),
        "sha256_indices": sha(indices),
    }
    c1["nodes_f32"] = nodes
    c1["indices_f32"] = indices
    rays256 = scenes.camera_rays(256, 256)
    h256, m256, _ = R.traverse(rays256)
    ka["KA2"]["wave_256"] = sums(h256, m256)
    ka["KA2"]["pixels"] = {
        "%d,%d" % (x, y): [float(h256[y * 256 + x]["t"]), float(h256[y * 256 + x]["u"]),
                           float(h256[y * 256 + x]["v"]), int(h256[y * 256 + x]["prim_id"])]
        for (x, y) in [(128, 128), (64, 64), (192, 64), (128, 192)]
    }
    c1["hits_256_f32"] = h256
    c1["mask_256_f32"] = m256
    rays512 = scenes.camera_rays(512, 512)
    h512, m512, _ = R.traverse(rays512)
    ka["KA3"] = sums(h512, m512)

    # trace-option variants (nanort.h:1055-1063, 1109-1116)
    o = default_trace_options()
    o["cull_back_face"] = 1
    hc, mc, _ = R.traverse(rays256, o)
    c1["hits_256_cull"], c1["mask_256_cull"] = hc, mc
    o = default_trace_options()
    o["skip_prim_id"] = 7
    hs, ms, _ = R.traverse(rays256, o)
    c1["hits_256_skip7"], c1["mask_256_skip7"] = hs, ms
    o = default_trace_options()
    o["prim_ids_range"] = (12, 500)
    hr, mr, _ = R.traverse(rays256, o)
    c1["hits_256_range12_500"], c1["mask_256_range12_500"] = hr, mr
    # min_t / max_t windows: equality is accepted at both ends of Intersect, and
    # the final predicate is strict (nanort.h:1133-1139, 2552)
    rw = rays256.copy()
    rw["min_t"] = 19.0
    rw["max_t"] = 24.5914974
    hw, mw, _ = R.traverse(rw)
    c1["hits_256_window"], c1["mask_256_window"] = hw, mw

    # fp64 instantiation on the same mesh
    v64 = verts.astype(np.float64)
    R64 = Reference(v64, faces)
    ok, st64 = R64.build(parallel=False)
    n64, i64 = R64.tree()
    c1["nodes_f64"] = n64
    c1["indices_f64"] = i64
    h64, m64, _ = R64.traverse(widen_rays(rays256))
    c1["hits_256_f64"], c1["mask_256_f64"] = h64, m64
    ka["KA2"]["wave_256_f64"] = sums(h64, m64)
    ka["KA2"]["stats_f64"] = {k: int(v) for k, v in st64.items() if k != "build_secs"}
    np.savez_compressed(os.path.join(GOLDEN, "c1_ref.npz"), **c1)

    # wave 2 on C1
    w2 = {}
    for kind in ("shadow", "bounce"):
        r2 = scenes.secondary_rays(kind, verts, faces, rays256, h256, m256)
        h2, m2, _ = R.traverse(r2)
        w2["hits_" + kind], w2["mask_" + kind] = h2, m2
        ka["KA2"]["wave2_" + kind] = dict(sums(h2, m2), num_rays=int(r2.shape[0]), sha256_rays=sha(r2))
    np.savez_compressed(os.path.join(GOLDEN, "c1_wave2.npz"), **w2)

    # ---- KA4: C3 plane, 1920x1080 -------------------------------------------
    pv, pf = scenes.plane(1000, 500)
    ka["KA4"] = {"sha256_vertices": sha(pv), "sha256_faces": sha(pf)}
    R3 = Reference(pv, pf)
    ok, st3 = R3.build(parallel=False)
    n3, i3 = R3.tree()
    ka["KA4"]["serial"] = {
        "stats": {k: int(v) for k, v in st3.items() if k != "build_secs"},
        "num_nodes": int(n3.shape[0]),
        "sha256_nodes_masked": sha(masked_nodes(n3)),
        "sha256_indices": sha(i3),
    }
    rays = scenes.camera_rays(1920, 1080)
    h3, m3, secs = R3.traverse(rays)
    ka["KA4"]["wave_1920x1080"] = sums(h3, m3)
    ok, st3p = R3.build(parallel=True)
    n3p, i3p = R3.tree()
    ka["KA4"]["parallel"] = {
        "stats": {k: int(v) for k, v in st3p.items() if k != "build_secs"},
        "num_nodes": int(n3p.shape[0]),
        "sha256_indices": sha(i3p),
    }
    np.savez_compressed(
        os.path.join(GOLDEN, "c3_sample.npz"),
        stride=np.array(STRIDE),
        hits=h3[::STRIDE],
        mask=m3[::STRIDE],
    )
    for kind in ("shadow", "bounce"):
        r2 = scenes.secondary_rays(kind, pv, pf, rays, h3, m3)
        h2, m2, _ = R3.traverse(r2)
        ka["KA4"]["wave2_" + kind] = dict(sums(h2, m2), num_rays=int(r2.shape[0]), sha256_rays=sha(r2))
    # fp64 (config C5)
    R5 = Reference(pv.astype(np.float64), pf)
    ok, st5 = R5.build(parallel=True)
    h5, m5, _ = R5.traverse(widen_rays(rays))
    ka["KA4"]["wave_1920x1080_f64"] = sums(h5, m5)

    # ---- C2 stand-in: lumpy sphere -------------------------------------------
    sv, sf = scenes.sphere()
    RS = Reference(sv, sf)
    ok, sts = RS.build(parallel=False)
    hs_, ms_, _ = RS.traverse(rays)
    ka["C2_sphere"] = {
        "num_faces": int(sf.shape[0]),
        "sha256_vertices": sha(sv),
        "stats": {k: int(v) for k, v in sts.items() if k != "build_secs"},
        "wave_1920x1080": sums(hs_, ms_),
    }
    np.savez_compressed(
        os.path.join(GOLDEN, "sphere_sample.npz"), stride=np.array(STRIDE), hits=hs_[::STRIDE], mask=ms_[::STRIDE]
    )

    # ---- two-level scene (nanosg), SURVEY §8f row 3 ---------------------------------
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from scene_fixture import instances
    from oracle.bindings import SceneReference

    RS2 = SceneReference()
    for v_, f_, x_ in instances():
        RS2.add_node(v_, f_, x_)
    assert RS2.commit()
    srays = scenes.camera_rays(320, 180)
    sh, sm = RS2.traverse(srays)
    sc = {"hits": sh, "mask": sm}
    for i in range(5):
        for k_, val in RS2.node_state(i).items():
            sc["node%d_%s" % (i, k_)] = val
    np.savez_compressed(os.path.join(GOLDEN, "scene_ref.npz"), **sc)
    ka["scene"] = {"num_hits": int(sm.sum()), "sha256_hits": sha(sh)}

    with open(os.path.join(GOLDEN, "known_answers.json"), "w") as f:
        json.dump(ka, f, indent=1, sort_keys=True)
    print(json.dumps(ka, indent=1, sort_keys=True)[:3000])


if __name__ == "__main__":
    main()
