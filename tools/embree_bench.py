"""Throughput of the Embree-2 API entry points of libnanort_embree.so on the GPU box: rtcIntersect1M (one batched
two-level traversal per call, host RTCRay records in and out) against rtcIntersect (one GPU round trip per ray), on the
scene of tests/embree_fixture.py at a finer tessellation.  Usage: python tools/embree_bench.py [width height]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import embree_fixture as ef  # noqa: E402
import scene_fixture  # noqa: E402

from nanort_amd import scenes  # noqa: E402


def main():
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
    inc, libdir = os.path.join(ROOT, "include"), os.path.join(ROOT, "nanort_amd", "lib")
    d = tempfile.mkdtemp()
    exe = os.path.join(d, "embree_check")
    subprocess.run(["g++", "-std=c++11", "-O2", "-I", inc, os.path.join(ROOT, "tests", "cpp", "embree_check.cc"), "-o", exe,
                    "-L", libdir, "-lnanort_embree", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    orig = scene_fixture.instances
    ef.instances = lambda: orig(sphere_res=(512, 256), plane_res=(1000, 500))
    ms = ef.meshes()
    print("scene: %d meshes, %d triangles" % (len(ms), sum(f.shape[0] for _, f in ms)))
    open(os.path.join(d, "scene.bin"), "wb").write(ef.scene_bytes(ms))
    cam = scenes.camera_rays(w, h)
    r = np.zeros((cam.shape[0], 8), dtype=np.float32)
    r[:, 0:3], r[:, 3:6], r[:, 7] = cam["org"], cam["dir"], 1.0e30
    env = dict(os.environ, EMBREE_CHECK_TIMING="4")
    for mode, rays in (("stream", r), ("single", np.ascontiguousarray(r[:: max(1, r.shape[0] // 20000)]))):
        open(os.path.join(d, "rays.bin"), "wb").write(ef.rays_bytes(rays))
        out = subprocess.run([exe, os.path.join(d, "scene.bin"), os.path.join(d, "rays.bin"), os.path.join(d, "out.bin"), mode],
                             env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        print(out.stdout.strip())


if __name__ == "__main__":
    main()
