"""bench_rows.py — the SURVEY §8(f) "next" rows as numbers in the driver's line (`next_rows` of bench.py): one figure
and one same-run parity sample each, outside the timed region.

  scene_fixture / scene_10k / scene_100k   two-level (instanced) traversal, nanosg::Scene::Traverse (reference examples/nanosg/nanosg.h:778-870)
  spheres_1m                  the particle example's primitive (reference examples/particle_primitive/main.cc:161-291)
  cylinders                   the cylinder example's primitive (reference examples/cylinder_primitive/main.cc:237-343)
  embree_stream               rtcIntersect1M of the Embree-2 shim (reference examples/embree-api/nanort-embree.cc:454-693)
  wavefront_frame             the device-shaded wavefront path tracer (examples/wavefront_path_tracer_gpu)

Parity samples use the CPU restatements under oracle/ as the CHECKER (the product never touches them): the GPU result on
a strided sample of the same rays against the restatement walking the SAME node arrays — every field bit for bit.
"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

W, H = 1920, 1080
BIN = os.path.join(ROOT, "tools", "bin")
INC, LIBDIR = os.path.join(ROOT, "include"), os.path.join(ROOT, "nanort_amd", "lib")


def _timed(fn, reps=5):
    import torch

    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


def _fields_equal(a, b, keys):
    return bool(all(np.ascontiguousarray(a[k]).tobytes() == np.ascontiguousarray(b[k]).tobytes() for k in keys))


def scene_rows():
    """5-node fixture (2 meshes, 1.07 M triangles), 10 000 and 100 000 instances of a small mesh, 1920x1080 camera rays resident in HBM."""
    import torch

    from nanort_amd import BVHAccel, Scene, TriangleMesh, scenes
    from nanort_amd.wire import SCENE_HIT_F32
    from oracle import bindings as ob
    from scene_fixture import instances, xform

    rays = scenes.camera_rays(W, H)
    d = torch.from_numpy(rays.view(np.uint8)).cuda()
    o = torch.empty(len(rays) * SCENE_HIT_F32.itemsize, dtype=torch.uint8, device="cuda")
    m = torch.empty(len(rays), dtype=torch.uint8, device="cuda")
    out = {}
    # --- the fixture
    sc, O, keep, ntri = Scene(), ob.SceneOracle(), [], 0
    for v, f, x in instances(sphere_res=(264, 132), plane_res=(1000, 500)):
        a = BVHAccel(np.float32)
        assert a.Build(f.shape[0], TriangleMesh(v, f))
        keep.append(a)
        sc.AddNode(a, x)
        O.add_node(v, f, x, tree=a.GetTree())
        ntri += f.shape[0]
    assert sc.Commit() and O.commit()
    ms = _timed(lambda: sc.TraverseBatchDevice(d, o, m))
    gh, gm = o.cpu().numpy().view(SCENE_HIT_F32), m.cpu().numpy()
    step = 173  # (co-prime with the image width: the sample covers every column)
    oh, om = O.traverse(rays[::step])
    out["scene_fixture"] = {
        "workload": "5 nodes (a displaced plane + 4 transformed spheres), %d triangles in the nodes' meshes, %dx%d camera rays in HBM" % (ntri, W, H),
        "value": round(len(rays) / ms / 1e3, 1), "unit": "Mrays/s", "ms": round(ms, 4),
        "parity": {"kind": "port (oracle/nanosg_oracle.c over the GPU-built local trees)", "rays": int(oh.shape[0]),
                   "bit_identical": bool(np.array_equal(om, gm[::step])) and _fields_equal(oh, gh[::step], ("t", "u", "v", "prim_id", "node_id"))}}
    del sc, O, keep
    # --- 10 000 and 100 000 instances (the single-pass walk; the listing path alone timed beside it)
    sv, sf = scenes.sphere(48, 24)
    sv = sv - np.array([0, 5, 0], dtype=np.float32)
    a = BVHAccel(np.float32)
    assert a.Build(sf.shape[0], TriangleMesh(sv, sf))
    tree = a.GetTree()
    for key, N, step in (("scene_10k", 10000, 4001), ("scene_100k", 100000, 16001)):
        rng = np.random.default_rng(5)
        sc, O = Scene(), ob.SceneOracle()
        for _ in range(N):
            x = xform(tuple(rng.uniform(0.01, 0.04, 3)), rng.uniform(0, 6.28), rng.uniform(0, 6.28), tuple(rng.uniform(-9, 9, 3) + np.array([0, 5, 0])))
            sc.AddNode(a, x)
            O.add_node(sv, sf, x, tree=tree)
        t0 = time.perf_counter()
        assert sc.Commit()
        commit_ms = (time.perf_counter() - t0) * 1e3
        ms = _timed(lambda: sc.TraverseBatchDevice(d, o, m), reps=3)
        redone = sc.LastRedone()
        gh, gm = o.cpu().numpy().view(SCENE_HIT_F32), m.cpu().numpy()
        assert O.commit()
        oh, om = O.traverse(rays[::step])  # (the restatement scans all the boxes per ray: keep the sample small)
        sc.SetTunable("single_pass", 0)
        ms_list = _timed(lambda: sc.TraverseBatchDevice(d, o, m), reps=2)
        lh, lm = o.cpu().numpy().view(SCENE_HIT_F32), m.cpu().numpy()
        out[key] = {
            "workload": "%d instances of a %d-triangle mesh, %dx%d camera rays in HBM" % (N, sf.shape[0], W, H),
            "value": round(len(rays) / ms / 1e3, 1), "unit": "Mrays/s", "ms": round(ms, 4), "commit_ms": round(commit_ms, 2),
            "path": "single-pass walk (k_scene_walk); rays handed to the listing path: %d" % redone,
            "listing_path_alone": {"value": round(len(rays) / ms_list / 1e3, 1), "ms": round(ms_list, 4),
                                   "records_identical": bool(np.array_equal(lm, gm)) and _fields_equal(lh, gh, ("t", "u", "v", "prim_id", "node_id"))},
            "hit_fraction": round(float(gm.mean()), 4),
            "parity": {"kind": "port (oracle/nanosg_oracle.c over the GPU-built local tree)", "rays": int(oh.shape[0]),
                       "bit_identical": bool(np.array_equal(om, gm[::step])) and _fields_equal(oh, gh[::step], ("t", "u", "v", "prim_id", "node_id"))}}
        del sc, O
    return out


def spheres_row(n=1000000):
    """The particle example at scale: n random spheres, the example's camera; the u/v pass is part of the launch."""
    import torch

    from nanort_amd import BVHAccel, SphereGeometry, scenes
    from nanort_amd.wire import HIT_F32
    from oracle import bindings as ob

    c, r = scenes.random_spheres(n)
    rays = scenes.particle_camera_rays(W, H)
    a = BVHAccel(np.float32)
    bms = []
    for _ in range(3):
        assert a.Build(n, SphereGeometry(c, r))
        bms.append(a.LastBuildMs())
    d = torch.from_numpy(rays.view(np.uint8)).cuda()
    o = torch.empty(len(rays) * 16, dtype=torch.uint8, device="cuda")
    m = torch.empty(len(rays), dtype=torch.uint8, device="cuda")
    ms = _timed(lambda: a.TraverseBatchDevice(d, o, m))
    gh, gm = o.cpu().numpy().view(HIT_F32), m.cpu().numpy()
    nodes, idx = a.GetTree()
    step = 97
    oh, om = ob.SphereOracle().traverse(nodes, idx, c, r, rays[::step])
    return {"workload": "particle example: %d random spheres, %dx%d camera, rays in HBM; traversal + the u/v pass" % (n, W, H),
            "value": round(len(rays) / ms / 1e3, 1), "unit": "Mrays/s", "ms": round(ms, 4), "build_ms": round(float(np.median(bms)), 3),
            "parity": {"kind": "port (oracle/sphere_oracle.c over the GPU-built tree)", "rays": int(oh.shape[0]),
                       "mask_t_prim_bit_identical": bool(np.array_equal(om, gm[::step])) and _fields_equal(oh, gh[::step], ("t", "prim_id")),
                       "max_abs_du_dv": float(max(np.abs(oh["u"] - gh["u"][::step]).max(), np.abs(oh["v"] - gh["v"][::step]).max()))}}


def cylinders_row(n=20000):
    """The cylinder example's workload: n random box-spanning cylinders, the example's camera; the normal pass is part of the launch."""
    import torch

    from nanort_amd import BVHAccel, CylinderGeometry, scenes
    from nanort_amd.wire import CYL_HIT_F32
    from oracle import bindings as ob

    v, r = scenes.random_cylinders(n)
    rays = scenes.particle_camera_rays(W, H)
    a = BVHAccel(np.float32)
    bms = []
    for _ in range(3):
        assert a.Build(n, CylinderGeometry(v, r))
        bms.append(a.LastBuildMs())
    d = torch.from_numpy(rays.view(np.uint8)).cuda()
    o = torch.empty(len(rays) * 28, dtype=torch.uint8, device="cuda")
    m = torch.empty(len(rays), dtype=torch.uint8, device="cuda")
    ms = _timed(lambda: a.TraverseBatchDevice(d, o, m), reps=3)
    gh, gm = o.cpu().numpy().view(CYL_HIT_F32), m.cpu().numpy()
    nodes, idx = a.GetTree()
    step = 1009
    oh, om = ob.CylinderOracle().traverse(nodes, idx, v, r, rays[::step])
    return {"workload": "cylinder example: %d random cylinders, %dx%d camera, rays in HBM; traversal + the normal pass" % (n, W, H),
            "value": round(len(rays) / ms / 1e3, 1), "unit": "Mrays/s", "ms": round(ms, 4), "build_ms": round(float(np.median(bms)), 3),
            "parity": {"kind": "port (oracle/cylinder_oracle.c over the GPU-built tree)", "rays": int(oh.shape[0]),
                       "bit_identical": bool(np.array_equal(om, gm[::step])) and bool(oh.tobytes() == np.ascontiguousarray(gh[::step]).tobytes())}}


def _binary(name, build_cmd):
    """A helper program under tools/bin/ (built by __graft_entry__.build(); built here when missing)."""
    exe = os.path.join(BIN, name)
    if not os.path.exists(exe):
        os.makedirs(BIN, exist_ok=True)
        subprocess.run(build_cmd + ["-o", exe], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    return exe


def embree_check_cmd():
    return ["g++", "-std=c++11", "-O2", "-I", INC, os.path.join(ROOT, "tests", "cpp", "embree_check.cc"), "-L", LIBDIR, "-lnanort_embree",
            "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"]


def wf_gpu_cmd():
    return ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off", "-DNANORT_USE_HIP_BACKEND", "-I", INC,
            os.path.join(ROOT, "examples", "wavefront_path_tracer_gpu", "main.hip"), "-L", LIBDIR, "-lnanort_hip", "-Wl,-rpath," + LIBDIR]


def build_host_cmd():
    return ["g++", "-std=c++11", "-O2", "-I", INC, os.path.join(ROOT, "tools", "build_host.cc"), "-L", LIBDIR, "-lnanort_hip", "-lnrt_scenes",
            "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-lpthread"]


def embree_row():
    """rtcIntersect1M over 96-byte host RTCRay records (2 M triangles in 5 meshes, 1920x1080): the stream query of the Embree-2
    API.  Parity sample: the same scene through nrtScene* directly (identity transforms) on a strided sample of the rays."""
    import embree_fixture as ef
    import scene_fixture

    from nanort_amd import scenes

    exe = _binary("embree_check", embree_check_cmd())
    orig = scene_fixture.instances
    saved = ef.instances
    ef.instances = lambda: orig(sphere_res=(512, 256), plane_res=(1000, 500))
    try:
        ms_ = ef.meshes()
    finally:
        ef.instances = saved
    cam = scenes.camera_rays(W, H)
    r = np.zeros((cam.shape[0], 8), dtype=np.float32)
    r[:, 0:3], r[:, 3:6], r[:, 7] = cam["org"], cam["dir"], 1.0e30
    d = tempfile.mkdtemp(prefix="nrt_embree_", dir="/tmp")
    open(os.path.join(d, "scene.bin"), "wb").write(ef.scene_bytes(ms_))
    open(os.path.join(d, "rays.bin"), "wb").write(ef.rays_bytes(r))
    p = subprocess.run([exe, os.path.join(d, "scene.bin"), os.path.join(d, "rays.bin"), os.path.join(d, "out.bin"), "stream"],
                       env=dict(os.environ, EMBREE_CHECK_TIMING="4"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    rates = [float(l.split(",")[-1].split()[0]) for l in p.stdout.splitlines() if "Mrays/s" in l]
    if p.returncode != 0 or not rates:
        return {"error": p.stdout[-500:]}
    out = {"workload": "rtcIntersect1M: %d triangles in %d meshes, %dx%d host RTCRay records (96 B) in and out" % (sum(f.shape[0] for _, f in ms_), len(ms_), W, H),
           "value": round(float(np.median(rates[1:] or rates)), 1), "unit": "Mrays/s"}
    # parity sample: the Embree records against nrtScene* over the same meshes (identity instances): hit flag, tfar, geomID (0-based node), primID, u, v
    from nanort_amd import BVHAccel, Scene, TriangleMesh

    _, _, recs = ef.parse_out(open(os.path.join(d, "out.bin"), "rb").read(), len(ms_), r.shape[0])
    sc, keep = Scene(), []
    for v, f in ms_:
        a = BVHAccel(np.float32)
        assert a.Build(f.shape[0], TriangleMesh(v, f))
        keep.append(a)
        sc.AddNode(a, np.eye(4, dtype=np.float32))
    assert sc.Commit()
    step = 211
    sh, sm = sc.TraverseBatch(cam[::step])
    e = recs[::step]
    hit = sm == 1
    same = bool(np.array_equal(e["geomID"] != ef.INVALID, hit)) and bool(np.array_equal(e["tfar"][hit], sh["t"][hit])) and \
        bool(np.array_equal(e["geomID"][hit], sh["node_id"][hit])) and bool(np.array_equal(e["primID"][hit], sh["prim_id"][hit])) and \
        bool(np.array_equal(e["u"][hit], sh["u"][hit])) and bool(np.array_equal(e["v"][hit], sh["v"][hit]))
    out["parity"] = {"kind": "the same scene through nrtSceneTraverseBatch_f32 (the layer the shim sits on)",
                     "rays": int(e.shape[0]), "hit_tfar_geomID_primID_u_v_equal": same}
    # ... and against an INDEPENDENT checker: the restated nanosg::Scene::Traverse (oracle/nanosg_oracle.c, pinned to the unmodified
    # nanosg.h) over the same per-mesh trees the GPU built, on a strided sample of the very records the stream call returned
    try:
        from oracle import bindings as ob

        O = ob.SceneOracle()
        for (v, f), a in zip(ms_, keep):
            O.add_node(v, f, np.eye(4, dtype=np.float32), tree=a.GetTree())
        O.commit()
        step2 = 1031
        oh, om = O.traverse(cam[::step2])
        e2 = recs[::step2]
        ohit = om == 1
        out["parity"]["vs_restated_nanosg"] = {
            "rays": int(e2.shape[0]),
            "hit_tfar_geomID_primID_u_v_equal": bool(np.array_equal(e2["geomID"] != ef.INVALID, ohit) and np.array_equal(e2["tfar"][ohit], oh["t"][ohit]) and
                                                     np.array_equal(e2["geomID"][ohit], oh["node_id"][ohit]) and np.array_equal(e2["primID"][ohit], oh["prim_id"][ohit]) and
                                                     np.array_equal(e2["u"][ohit], oh["u"][ohit]) and np.array_equal(e2["v"][ohit], oh["v"][ohit]))}
    except Exception as ex:  # pragma: no cover
        out["parity"]["vs_restated_nanosg"] = {"error": repr(ex)}
    return out


def wavefront_row():
    """The device-shaded wavefront path tracer: 1920x1080, 2 samples per pixel, depth 3, 1 M triangles — every wave through the
    header's device entry points, only the image crosses PCIe.  Default mode: ONE stream, a depth's shadow query and the next
    path wave in ONE launch (BVHAccel::TraverseBatchesDevice); beside it the same frame with separate launches on one stream and
    with the shadow queries on a second stream.  Its parity is the GPU suite's (image == host-shaded image; the one-launch and
    separate-launch images are bit-identical)."""
    exe = _binary("wf_gpu", wf_gpu_cmd())
    d = tempfile.mkdtemp(prefix="nrt_wf_", dir="/tmp")

    def run(extra):
        p = subprocess.run([exe, "--size", str(W), str(H), "--spp", "2", "--depth", "3", "--grid", "1000", "500"] + extra,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        line = [l for l in p.stdout.splitlines() if "Mray_slots_per_s" in l]
        if p.returncode != 0 or not line:
            return None, p.stdout[-500:]
        tok = line[-1].split()
        return {tok[i]: tok[i + 1] for i in range(0, len(tok) - 1, 2)}, None

    kv, err = run(["--out", os.path.join(d, "img.f32")])
    if kv is None:
        return {"error": err}
    kv1, _ = run(["--streams", "1"])
    kv2, _ = run(["--streams", "2"])
    return {"workload": "device-shaded wavefront path tracer: %s triangles, %s, spp %s, depth %s; one ray slot per pixel and wave; one stream, a "
                        "depth's shadow query and the next path wave in one launch" % (kv.get("triangles"), kv.get("image"), kv.get("spp"), kv.get("depth")),
        "value": round(float(kv["Mray_slots_per_s"]), 1), "unit": "M ray slots/s end to end", "frame_ms": float(kv["frame_ms"]),
        "separate_launches_one_stream": {"value": round(float(kv1["Mray_slots_per_s"]), 1), "frame_ms": float(kv1["frame_ms"])} if kv1 else None,
        "shadow_queries_on_a_second_stream": {"value": round(float(kv2["Mray_slots_per_s"]), 1), "frame_ms": float(kv2["frame_ms"])} if kv2 else None,
        "image_sum": float(kv["image_sum"]),
        "parity": _wavefront_parity(exe, d)}


def wf_host_cmd():
    return ["g++", "-std=c++11", "-O2", "-fopenmp", "-DNANORT_USE_HIP_BACKEND", "-I", INC, os.path.join(ROOT, "examples", "wavefront_path_tracer", "main.cc"),
            "-L", LIBDIR, "-lnanort_hip", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"]


def _wavefront_parity(gpu_exe, d):
    """In-run check of the wavefront row: the device-shaded image against the HOST-shaded example's (every wave through
    BVHAccel::TraverseBatch / TraverseBatches, shading in plain C++) at 480x270 — equal up to the device's sinf / cosf — and the
    host-shaded example's own --verify (its GPU-traced image == its per-ray CPU Traverse() image in every float)."""
    try:
        host = _binary("wf_host", wf_host_cmd())
        args = ["--size", "480", "270", "--spp", "2", "--depth", "3", "--grid", "400", "200"]
        a = subprocess.run([gpu_exe] + args + ["--out", os.path.join(d, "g.f32")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        b = subprocess.run([host] + args + ["--out", os.path.join(d, "h.ppm"), "--raw", os.path.join(d, "h.f32"), "--verify"], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=600)
        if a.returncode != 0 or b.returncode != 0:
            return {"in_run": True, "error": (a.stdout + b.stdout)[-400:]}
        g = np.fromfile(os.path.join(d, "g.f32"), dtype=np.float32)
        h = np.fromfile(os.path.join(d, "h.f32"), dtype=np.float32)
        diff = np.abs(g - h)
        return {"in_run": True, "kind": "device-shaded image vs the host-shaded example's (480x270, spp 2, depth 3)",
                "pixels_components_within_2e-3": round(float((diff <= 2e-3).mean()), 5), "mean_abs_diff": float(diff.mean()),
                "host_example_gpu_traced_image_equals_its_cpu_traced_image": "verify: 0 differing float components" in b.stdout}
    except Exception as ex:  # pragma: no cover
        return {"in_run": True, "error": repr(ex)}


# ---------------------------------------------------------------------------
# hardware counters of a row's dominant kernel (outside every timed region): bench_rows.py --pmc-row NAME is re-run under
# rocprofv3 --pmc (kernel trace only; one pass per counter set, as MI355X_MICROARCH.md prescribes)
# ---------------------------------------------------------------------------
ROW_KERNELS = {"scene_10k": "k_scene_walk", "scene_fixture": "k_scene_trace", "spheres_1m": "k_traverse_wide", "cylinders": "k_traverse_wide"}


def pmc_row_child(name):
    """The row's workload without its parity sample: set-up, then a few launches of the query (the rocprofv3 child)."""
    import torch

    from nanort_amd import BVHAccel, CylinderGeometry, Scene, SphereGeometry, TriangleMesh, scenes

    if name in ("scene_10k", "scene_fixture"):
        from nanort_amd.wire import SCENE_HIT_F32
        from scene_fixture import instances, xform

        rays = scenes.camera_rays(W, H)
        sc, keep = Scene(), []
        if name == "scene_fixture":
            for v, f, x in instances(sphere_res=(264, 132), plane_res=(1000, 500)):
                a = BVHAccel(np.float32)
                assert a.Build(f.shape[0], TriangleMesh(v, f))
                keep.append(a)
                sc.AddNode(a, x)
        else:
            rng = np.random.default_rng(5)
            sv, sf = scenes.sphere(48, 24)
            sv = sv - np.array([0, 5, 0], dtype=np.float32)
            a = BVHAccel(np.float32)
            assert a.Build(sf.shape[0], TriangleMesh(sv, sf))
            keep.append(a)
            for _ in range(10000):
                sc.AddNode(a, xform(tuple(rng.uniform(0.01, 0.04, 3)), rng.uniform(0, 6.28), rng.uniform(0, 6.28), tuple(rng.uniform(-9, 9, 3) + np.array([0, 5, 0]))))
        assert sc.Commit()
        d = torch.from_numpy(rays.view(np.uint8)).cuda()
        o = torch.empty(len(rays) * SCENE_HIT_F32.itemsize, dtype=torch.uint8, device="cuda")
        m = torch.empty(len(rays), dtype=torch.uint8, device="cuda")
        run = lambda: sc.TraverseBatchDevice(d, o, m)  # noqa: E731
    else:
        rays = scenes.particle_camera_rays(W, H)
        a = BVHAccel(np.float32)
        if name == "spheres_1m":
            c, r = scenes.random_spheres(1000000)
            assert a.Build(1000000, SphereGeometry(c, r))
            rec = 16
        else:
            v, r = scenes.random_cylinders(20000)
            assert a.Build(20000, CylinderGeometry(v, r))
            rec = 28
        d = torch.from_numpy(rays.view(np.uint8)).cuda()
        o = torch.empty(len(rays) * rec, dtype=torch.uint8, device="cuda")
        m = torch.empty(len(rays), dtype=torch.uint8, device="cuda")
        run = lambda: a.TraverseBatchDevice(d, o, m)  # noqa: E731
    for _ in range(4):
        run()
    torch.cuda.synchronize()
    print("pmc_row_child", name, len(rays), flush=True)


def row_counters(name, keep_dir=None):
    """{hbm_frac, l2_hit_rate, lane_util, wait_frac, l1_frac, ...} of the row's dominant kernel, mean per launch."""
    import csv
    import glob
    import shutil

    import bench

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    tmp = keep_dir or tempfile.mkdtemp(prefix="nrt_rowpmc_", dir="/tmp")
    key = ROW_KERNELS[name]
    acc, durs, errors = {}, [], []
    for tag, counters, _fallback in bench.PMC_PASSES:
        for sub, cs in ([(tag, counters)] if tag != "fetch_tcp" else [("fetch", "FETCH_SIZE"), ("tcp", "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum")]):
            out_dir = os.path.join(tmp, name + "_" + sub)
            cmd = [exe, "--kernel-trace", "--pmc"] + cs.split() + ["--output-format", "csv", "-d", out_dir, "-o", "p", "--", sys.executable,
                                                                     os.path.join(ROOT, "bench_rows.py"), "--pmc-row", name]
            try:
                r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, TMPDIR="/tmp"), timeout=300, cwd="/tmp")
            except Exception as e:  # pragma: no cover
                errors.append("%s: %r" % (sub, e))
                continue
            if r.returncode != 0:
                errors.append("%s: rc %d: %s" % (sub, r.returncode, r.stdout[-200:]))
                continue
            for path in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
                for x in csv.DictReader(open(path)):
                    if key in x.get("Kernel_Name", ""):
                        acc.setdefault(x["Counter_Name"], {}).setdefault(int(x["Dispatch_Id"]), 0.0)
                        acc[x["Counter_Name"]][int(x["Dispatch_Id"])] += float(x["Counter_Value"])
            if sub == "sq":
                for path in glob.glob(os.path.join(out_dir, "**", "*kernel_trace.csv"), recursive=True):
                    for x in csv.DictReader(open(path)):
                        if key in x.get("Kernel_Name", ""):
                            durs.append((int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) * 1e-9)
    if not keep_dir:
        shutil.rmtree(tmp, ignore_errors=True)
    if not acc or not durs:
        return {"error": "; ".join(errors) or "no counter rows for " + key}
    c = {k: float(np.mean(list(v.values()))) for k, v in acc.items()}
    secs = float(np.mean(durs))  # (the launch under the counter profiler)
    out = {"kernel_contains": key, "profiled_launch_ms": round(secs * 1e3, 4)}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        b = c["FETCH_SIZE"] * 1024.0 * 2.0 + c["WRITE_SIZE"] * 1024.0
        out["hbm"] = {"bytes_per_launch": int(b), "frac": round(b / secs / 1e9 / bench.HBM_PEAK_GBS, 4)}
    if "TCC_HIT_sum" in c:
        out["l2_hit_rate"] = round(c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
        out["l1"] = {"lookups_per_launch": int(c["TCP_TOTAL_CACHE_ACCESSES_sum"]), "frac": round(c["TCP_TOTAL_CACHE_ACCESSES_sum"] / secs / 1e9 / bench.L1_PEAK_GACC_S, 4)}
    if "SQ_INSTS_VALU" in c:
        out["valu"] = {"wave_insts": int(c["SQ_INSTS_VALU"]), "lane_util": round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_INSTS_VALU"]), 4)}
        if c.get("SQ_WAVE_CYCLES"):
            out["wait_frac_of_wave_cycles"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4)
    if errors:
        out["errors"] = errors
    return out


def next_rows(counters=True, pmc_dir=None):
    out = {}
    for name, fn in (("scenes", scene_rows), ("spheres_1m", spheres_row), ("cylinders", cylinders_row), ("embree_stream", embree_row),
                     ("wavefront_frame", wavefront_row)):
        t0 = time.perf_counter()
        try:
            r = fn()
        except Exception as e:  # pragma: no cover
            r = {"error": repr(e)}
        if name == "scenes" and "error" not in r:
            out.update(r)
        else:
            out[name] = r
        out.setdefault("_seconds", {})[name] = round(time.perf_counter() - t0, 1)
    if counters:  # hardware counters of the rows' dominant kernels (three rocprofv3 passes each, outside every timed region)
        for name in ("scene_10k", "spheres_1m", "cylinders"):
            if name in out and "error" not in out[name]:
                t0 = time.perf_counter()
                try:
                    out[name]["counters"] = row_counters(name, keep_dir=os.path.join(pmc_dir, "rows") if pmc_dir else None)
                except Exception as e:  # pragma: no cover
                    out[name]["counters"] = {"error": repr(e)}
                out["_seconds"]["counters_" + name] = round(time.perf_counter() - t0, 1)
    return out


if __name__ == "__main__":
    import json

    if len(sys.argv) > 2 and sys.argv[1] == "--pmc-row":
        pmc_row_child(sys.argv[2])
    else:
        print(json.dumps(next_rows(), indent=1))
