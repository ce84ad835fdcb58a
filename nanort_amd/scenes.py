"""Synthetic meshes and ray waves of SURVEY.md §8(d) (host side, numpy).

Thin ctypes wrapper over nanort_amd/lib/libnrt_scenes.so (csrc/scenes.c, built
with gcc by __graft_entry__.build()).  Deterministic: same bytes everywhere.
"""
import ctypes
import os

import numpy as np

from .wire import HIT_F32, RAY_F32

_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libnrt_scenes.so")
        if not os.path.exists(path):
            raise RuntimeError(
                "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C nanort_amd/csrc`) first" % path
            )
        L = ctypes.CDLL(path)
        vp, u32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
        L.nrt_scene_plane.argtypes = [u32, u32, vp, vp]
        L.nrt_scene_plane.restype = None
        L.nrt_scene_sphere.argtypes = [u32, u32, vp, vp]
        L.nrt_scene_sphere.restype = None
        L.nrt_rays_camera.argtypes = [u32, u32, u32, u32, vp]
        L.nrt_rays_camera.restype = None
        L.nrt_rays_camera_rows.argtypes = [u32, u32, u32, u32, u32, vp]
        L.nrt_rays_camera_rows.restype = None
        L.nrt_rays_secondary.argtypes = [ctypes.c_int, vp, vp, vp, vp, vp, u64, u64, vp]
        L.nrt_rays_secondary.restype = u64
        L.nrt_scene_random_spheres.argtypes = [u64, vp, vp, vp, vp]
        L.nrt_scene_random_spheres.restype = None
        L.nrt_rays_particle_camera.argtypes = [u32, u32, vp]
        L.nrt_scene_random_cylinders.argtypes = [u64, vp, vp, vp, vp]
        L.nrt_scene_random_cylinders.restype = None
        L.nrt_rays_particle_camera.restype = None
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def plane(nx, ny):
    """Plane(nx, ny): 2*nx*ny triangles. C3/C5 = plane(1000, 500); C4 = plane(2500, 2000)."""
    verts = np.empty(((nx + 1) * (ny + 1), 3), dtype=np.float32)
    faces = np.empty((2 * nx * ny, 3), dtype=np.uint32)
    _lib().nrt_scene_plane(nx, ny, _p(verts), _p(faces))
    return verts, faces


def sphere(nu=264, nv=132):
    """Closed lumpy sphere, 2*nu*(nv-1) triangles (69 168 at the defaults): C2 stand-in."""
    verts = np.empty((nu * (nv - 1) + 2, 3), dtype=np.float32)
    faces = np.empty((2 * nu * (nv - 1), 3), dtype=np.uint32)
    _lib().nrt_scene_sphere(nu, nv, _p(verts), _p(faces))
    return verts, faces


def random_spheres(n, bmin=(-1.0, -1.0, -1.0), bmax=(1.0, 1.0, 1.0)):
    """The particle example's scene (reference examples/particle_primitive/main.cc:295-325): centres (n, 3), radii (n,)."""
    centers = np.empty((n, 3), dtype=np.float32)
    radii = np.empty((n,), dtype=np.float32)
    lo = np.asarray(bmin, dtype=np.float32)
    hi = np.asarray(bmax, dtype=np.float32)
    _lib().nrt_scene_random_spheres(n, _p(lo), _p(hi), _p(centers), _p(radii))
    return centers, radii


def random_cylinders(n, bmin=(-1.0, -1.0, -1.0), bmax=(1.0, 1.0, 1.0)):
    """The cylinder example's scene (reference examples/cylinder_primitive/main.cc:428-462): end points (n, 2, 3),
    radii (n, 2)."""
    verts = np.empty((n, 2, 3), dtype=np.float32)
    radii = np.empty((n, 2), dtype=np.float32)
    lo = np.asarray(bmin, dtype=np.float32)
    hi = np.asarray(bmax, dtype=np.float32)
    _lib().nrt_scene_random_cylinders(n, _p(lo), _p(hi), _p(verts), _p(radii))
    return verts, radii


def particle_camera_rays(width, height):
    """That example's camera (main.cc:367-389), row-major."""
    rays = np.empty((width * height,), dtype=RAY_F32)
    _lib().nrt_rays_particle_camera(width, height, _p(rays))
    return rays


def camera_rays(width, height, y0=0, y1=None):
    """Wave 1: objrender camera, rows [y0, y1) of a width x height image, row-major."""
    if y1 is None:
        y1 = height
    rays = np.empty(((y1 - y0) * width,), dtype=RAY_F32)
    _lib().nrt_rays_camera(width, height, y0, y1, _p(rays))
    return rays


def camera_rays_rows(width, height, y0, y_step, rows):
    """Rows y0, y0+y_step, ... of the width x height camera image (multi-GPU interleaved tiles)."""
    rays = np.empty((rows * width,), dtype=RAY_F32)
    _lib().nrt_rays_camera_rows(width, height, y0, y_step, rows, _p(rays))
    return rays


def secondary_rays(kind, verts, faces, rays, hits, mask, pixel_base=0):
    """Wave 2 from wave-1 hits. kind: 'shadow' | 'bounce'. Returns a compacted ray buffer."""
    assert rays.dtype == RAY_F32 and hits.dtype == HIT_F32
    verts = np.ascontiguousarray(verts, dtype=np.float32)
    faces = np.ascontiguousarray(faces, dtype=np.uint32)
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    out = np.empty((int(np.count_nonzero(mask)),), dtype=RAY_F32)
    k = {"shadow": 0, "bounce": 1}[kind]
    m = _lib().nrt_rays_secondary(
        k, _p(verts), _p(faces), _p(rays), _p(hits), _p(mask), rays.shape[0], pixel_base, _p(out)
    )
    assert m == out.shape[0]
    return out


def load_c1_mesh():
    """Cornell box + Suzanne (980 triangles): the C1 fixture under tests/golden/."""
    path = os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "c1_mesh.npz"
    )
    d = np.load(path)
    return np.ascontiguousarray(d["vertices"]), np.ascontiguousarray(d["faces"])
