"""ctypes bindings for the two CPU checkers — TEST INFRASTRUCTURE ONLY.

  Oracle     oracle/liboracle.so        plain-C restatement (nanort_oracle.c)
  Reference  oracle/_ref/libnanort_ref.so   the unmodified reference header
             behind a C shim (ref_shim.cc); prebuilt in the build container,
             absent when it was never built (callers must then skip).
"""
import ctypes
import os

import numpy as np

from nanort_amd.wire import (
    TRACE_OPTIONS,
    default_trace_options,
    hit_dtype,
    node_dtype,
    ray_dtype,
    suffix,
)

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_PATH = os.path.join(_HERE, "liboracle.so")
REF_PATH = os.path.join(_HERE, "_ref", "libnanort_ref.so")


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _trace_opt_words(opts):
    if opts is None:
        opts = default_trace_options()
    return np.frombuffer(np.asarray(opts, dtype=TRACE_OPTIONS).tobytes(), dtype=np.uint32).copy()


class Oracle:
    """The plain-C restatement."""

    def __init__(self):
        if not os.path.exists(ORACLE_PATH):
            raise RuntimeError("%s missing: run `make -C oracle`" % ORACLE_PATH)
        L = ctypes.CDLL(ORACLE_PATH)
        vp, u32, u64, sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_size_t
        for s in ("f32", "f64"):
            f = getattr(L, "orc_build_" + s)
            f.argtypes = [vp, sz, vp, u32, u32, u32, u32, ctypes.POINTER(vp), vp, vp]
            f.restype = u64
            g = getattr(L, "orc_traverse_" + s)
            g.argtypes = [vp, vp, vp, sz, vp, vp, u64, vp, vp, vp, vp]
            g.restype = None
            w4 = getattr(L, "orc_traverse_wide4_model_" + s)
            w4.argtypes = [vp, vp, vp, sz, vp, vp, u64, vp, vp, vp, vp, vp, vp, u64, vp]
            w4.restype = None
        L.orc_free.argtypes = [vp]
        L.orc_sizeof.argtypes = [ctypes.c_int]
        L.orc_sizeof.restype = ctypes.c_int
        self.L = L

    def build(self, verts, faces, min_leaf=4, max_depth=256, bin_size=64, stride=None):
        """Serial reference build. Returns (nodes, indices, stats dict)."""
        real = verts.dtype
        s = suffix(real)
        faces = np.ascontiguousarray(faces, dtype=np.uint32)
        n = faces.shape[0]
        if stride is None:
            stride = 3 * verts.dtype.itemsize
        indices = np.empty((n,), dtype=np.uint32)
        stats = np.zeros((3,), dtype=np.uint32)
        nodes_ptr = ctypes.c_void_p()
        nn = getattr(self.L, "orc_build_" + s)(
            _p(verts), stride, _p(faces), n, min_leaf, max_depth, bin_size,
            ctypes.byref(nodes_ptr), _p(indices), _p(stats),
        )
        nd = node_dtype(real)
        if nn == 0:
            return np.empty((0,), dtype=nd), indices[:0], None
        buf = (ctypes.c_char * (nn * nd.itemsize)).from_address(nodes_ptr.value)
        nodes = np.frombuffer(buf, dtype=nd).copy()
        self.L.orc_free(nodes_ptr)
        return nodes, indices, {
            "max_tree_depth": int(stats[0]),
            "num_leaf_nodes": int(stats[1]),
            "num_branch_nodes": int(stats[2]),
        }

    def traverse(self, nodes, indices, verts, faces, rays, opts=None, stride=None, count=False):
        """Returns (hits, mask[, counters(nodes, leaves, tris, max_stack)])."""
        real = verts.dtype
        s = suffix(real)
        assert nodes.dtype == node_dtype(real) and rays.dtype == ray_dtype(real)
        faces = np.ascontiguousarray(faces, dtype=np.uint32)
        indices = np.ascontiguousarray(indices, dtype=np.uint32)
        nodes = np.ascontiguousarray(nodes)
        rays = np.ascontiguousarray(rays)
        if stride is None:
            stride = 3 * verts.dtype.itemsize
        n = rays.shape[0]
        hits = np.zeros((n,), dtype=hit_dtype(real))
        mask = np.zeros((n,), dtype=np.uint8)
        counters = np.zeros((4,), dtype=np.uint64) if count else None
        w = _trace_opt_words(opts)
        getattr(self.L, "orc_traverse_" + s)(
            _p(nodes), _p(indices), _p(verts), stride, _p(faces), _p(rays), n, _p(w),
            _p(hits), _p(mask), _p(counters),
        )
        if count:
            return hits, mask, counters
        return hits, mask

    def traverse_wide4_model(self, nodes, indices, verts, faces, rays, opts=None, trail_cap=0):
        """The sequential MODEL of the kernel's two-levels-per-step walk (oracle/wide4_model_body.inc).
        Returns (hits, mask, counters(steps, leaves, tris, max_stack), trail_reference, trail_model): the two trails are the
        leaf sequences (node indices, all rays concatenated) of the reference loop and of the model, when trail_cap > 0."""
        real = verts.dtype
        s = suffix(real)
        assert nodes.dtype == node_dtype(real) and rays.dtype == ray_dtype(real)
        faces = np.ascontiguousarray(faces, dtype=np.uint32)
        indices = np.ascontiguousarray(indices, dtype=np.uint32)
        nodes = np.ascontiguousarray(nodes)
        rays = np.ascontiguousarray(rays)
        n = rays.shape[0]
        hits = np.zeros((n,), dtype=hit_dtype(real))
        mask = np.zeros((n,), dtype=np.uint8)
        counters = np.zeros((4,), dtype=np.uint64)
        t_ref = np.zeros((max(1, trail_cap),), dtype=np.uint32)
        t_w4 = np.zeros((max(1, trail_cap),), dtype=np.uint32)
        lens = np.zeros((2,), dtype=np.uint64)
        w = _trace_opt_words(opts)
        getattr(self.L, "orc_traverse_wide4_model_" + s)(
            _p(nodes), _p(indices), _p(verts), 3 * verts.dtype.itemsize, _p(faces), _p(rays), n, _p(w), _p(hits), _p(mask),
            _p(counters), _p(t_ref) if trail_cap else None, _p(t_w4) if trail_cap else None, int(trail_cap), _p(lens))
        if trail_cap:
            assert lens[0] <= trail_cap and lens[1] <= trail_cap, "trail buffer too small: %s" % lens
            return hits, mask, counters, t_ref[: int(lens[0])], t_w4[: int(lens[1])]
        return hits, mask, counters, None, None


class _ShimBuildOptions(ctypes.Structure):
    _fields_ = [
        ("min_leaf_primitives", ctypes.c_uint32),
        ("max_tree_depth", ctypes.c_uint32),
        ("bin_size", ctypes.c_uint32),
        ("shallow_depth", ctypes.c_uint32),
        ("min_primitives_for_parallel_build", ctypes.c_uint32),
        ("cache_bbox", ctypes.c_uint32),
    ]


class _ShimStats(ctypes.Structure):
    _fields_ = [
        ("max_tree_depth", ctypes.c_uint32),
        ("num_leaf_nodes", ctypes.c_uint32),
        ("num_branch_nodes", ctypes.c_uint32),
        ("build_secs", ctypes.c_float),
    ]


def reference_available():
    return os.path.exists(REF_PATH)


class Reference:
    """One reference BVHAccel<T> over a caller-owned mesh (kept alive here)."""

    _lib = None
    PATH = REF_PATH

    @classmethod
    def lib(cls):
        if cls._lib is None:
            if not os.path.exists(cls.PATH):
                raise RuntimeError("%s missing: run `make -C oracle ref` where /root/reference exists" % cls.PATH)
            L = ctypes.CDLL(cls.PATH)
            vp, u32, u64, sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_size_t
            L.ref_sizeof.argtypes = [ctypes.c_int]
            L.ref_sizeof.restype = ctypes.c_int
            L.ref_max_threads.restype = ctypes.c_int
            for s in ("f32", "f64"):
                getattr(L, "ref_create_" + s).argtypes = [vp, sz, vp, u32]
                getattr(L, "ref_create_" + s).restype = vp
                getattr(L, "ref_destroy_" + s).argtypes = [vp]
                getattr(L, "ref_build_" + s).argtypes = [vp, vp, vp, ctypes.c_int]
                getattr(L, "ref_build_" + s).restype = ctypes.c_int
                getattr(L, "ref_num_nodes_" + s).argtypes = [vp]
                getattr(L, "ref_num_nodes_" + s).restype = u64
                getattr(L, "ref_num_indices_" + s).argtypes = [vp]
                getattr(L, "ref_num_indices_" + s).restype = u64
                getattr(L, "ref_get_tree_" + s).argtypes = [vp, vp, vp]
                getattr(L, "ref_load_tree_" + s).argtypes = [vp, vp, u64, vp, u64]
                getattr(L, "ref_load_tree_" + s).restype = ctypes.c_int
                getattr(L, "ref_bounding_box_" + s).argtypes = [vp, vp, vp]
                getattr(L, "ref_traverse_" + s).argtypes = [vp, vp, u64, vp, vp, vp, ctypes.c_int, ctypes.c_int]
                getattr(L, "ref_traverse_" + s).restype = ctypes.c_double
            cls._lib = L
        return cls._lib

    def __init__(self, verts, faces, stride=None):
        self.L = self.lib()
        self.real = verts.dtype
        self.s = suffix(self.real)
        self.verts = np.ascontiguousarray(verts)
        self.faces = np.ascontiguousarray(faces, dtype=np.uint32)
        if stride is None:
            stride = 3 * self.verts.dtype.itemsize
        self.stride = stride
        self.h = getattr(self.L, "ref_create_" + self.s)(_p(self.verts), stride, _p(self.faces), self.faces.shape[0])

    def __del__(self):
        try:
            getattr(self.L, "ref_destroy_" + self.s)(self.h)
        except Exception:
            pass

    def build(self, parallel=False, min_leaf=4, max_depth=256, bin_size=64, cache_bbox=False, threads=0):
        """Reference Build(). parallel=False forces the serial arm (nanort.h:2129)."""
        o = _ShimBuildOptions(
            min_leaf, max_depth, bin_size, 4, (1024 * 8) if parallel else 0xFFFFFFFF, 1 if cache_bbox else 0
        )
        st = _ShimStats()
        ok = getattr(self.L, "ref_build_" + self.s)(self.h, ctypes.byref(o), ctypes.byref(st), int(threads))
        return bool(ok), {
            "max_tree_depth": st.max_tree_depth,
            "num_leaf_nodes": st.num_leaf_nodes,
            "num_branch_nodes": st.num_branch_nodes,
            "build_secs": st.build_secs,
        }

    def tree(self):
        nn = getattr(self.L, "ref_num_nodes_" + self.s)(self.h)
        ni = getattr(self.L, "ref_num_indices_" + self.s)(self.h)
        nodes = np.zeros((nn,), dtype=node_dtype(self.real))
        indices = np.zeros((ni,), dtype=np.uint32)
        getattr(self.L, "ref_get_tree_" + self.s)(self.h, _p(nodes), _p(indices))
        return nodes, indices

    def load_tree(self, nodes, indices):
        nodes = np.ascontiguousarray(nodes, dtype=node_dtype(self.real))
        indices = np.ascontiguousarray(indices, dtype=np.uint32)
        return bool(
            getattr(self.L, "ref_load_tree_" + self.s)(self.h, _p(nodes), nodes.shape[0], _p(indices), indices.shape[0])
        )

    def bounding_box(self):
        bmin = np.zeros(3, dtype=self.real)
        bmax = np.zeros(3, dtype=self.real)
        getattr(self.L, "ref_bounding_box_" + self.s)(self.h, _p(bmin), _p(bmax))
        return bmin, bmax

    def traverse(self, rays, opts=None, threads=0, chunk=1920):
        """Reference Traverse() per ray under an OpenMP dynamic row loop.
        Returns (hits, mask, seconds)."""
        rays = np.ascontiguousarray(rays, dtype=ray_dtype(self.real))
        n = rays.shape[0]
        hits = np.zeros((n,), dtype=hit_dtype(self.real))
        mask = np.zeros((n,), dtype=np.uint8)
        o = np.asarray(default_trace_options() if opts is None else opts, dtype=TRACE_OPTIONS)
        secs = getattr(self.L, "ref_traverse_" + self.s)(
            self.h, _p(rays), n, _p(o.reshape(1)), _p(hits), _p(mask), threads, chunk
        )
        return hits, mask, secs

    def max_threads(self):
        return self.L.ref_max_threads()


def fnv1a64(data):
    """FNV-1a 64 over a bytes-like (vectorised per byte via Python int loop on chunks)."""
    h = 0xCBF29CE484222325
    for b in memoryview(data).tobytes():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


# ---------------------------------------------------------------------------
# Two-level scenes: reference nanosg behind a shim, and its C restatement
# ---------------------------------------------------------------------------
SCENE_HIT = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("prim_id", "<u4"), ("node_id", "<u4")])
REF_SCENE_PATH = os.path.join(_HERE, "_ref", "libnanosg_ref.so")


def scene_reference_available():
    return os.path.exists(REF_SCENE_PATH)


class SceneReference:
    """The unmodified examples/nanosg Scene<float, Mesh> (oracle/ref_scene_shim.cc)."""

    def __init__(self):
        L = ctypes.CDLL(REF_SCENE_PATH)
        vp, u32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
        L.refsg_create.restype = vp
        L.refsg_destroy.argtypes = [vp]
        L.refsg_add_node.argtypes = [vp, vp, u32, vp, u32, vp]
        L.refsg_add_node.restype = ctypes.c_int
        L.refsg_commit.argtypes = [vp]
        L.refsg_commit.restype = ctypes.c_int
        L.refsg_node_state.argtypes = [vp, u32, vp]
        L.refsg_traverse.argtypes = [vp, vp, u64, ctypes.c_int, vp, vp]
        L.refsg_bounds.argtypes = [vp, vp, vp]
        self.L = L
        self.h = L.refsg_create()
        self.n = 0

    def __del__(self):
        try:
            self.L.refsg_destroy(self.h)
        except Exception:
            pass

    def add_node(self, verts, faces, xform):
        verts = np.ascontiguousarray(verts, dtype=np.float32)
        faces = np.ascontiguousarray(faces, dtype=np.uint32)
        xform = np.ascontiguousarray(xform, dtype=np.float32).reshape(4, 4)
        self.n += 1
        return self.L.refsg_add_node(self.h, _p(verts), verts.shape[0], _p(faces), faces.shape[0], _p(xform))

    def commit(self):
        return bool(self.L.refsg_commit(self.h))

    def node_state(self, i):
        out = np.zeros(54, dtype=np.float32)
        self.L.refsg_node_state(self.h, i, _p(out))
        return {"xbmin": out[0:3], "xbmax": out[3:6], "inv_xform": out[6:22].reshape(4, 4),
                "inv_xform33": out[22:38].reshape(4, 4), "xform": out[38:54].reshape(4, 4)}

    def bounds(self):
        bmin = np.zeros(3, dtype=np.float32)
        bmax = np.zeros(3, dtype=np.float32)
        self.L.refsg_bounds(self.h, _p(bmin), _p(bmax))
        return bmin, bmax

    def traverse(self, rays, cull_back_face=False):
        rays = np.ascontiguousarray(rays, dtype=ray_dtype(np.float32))
        hits = np.zeros((rays.shape[0],), dtype=SCENE_HIT)
        mask = np.zeros((rays.shape[0],), dtype=np.uint8)
        self.L.refsg_traverse(self.h, _p(rays), rays.shape[0], 1 if cull_back_face else 0, _p(hits), _p(mask))
        return hits, mask


class _SgNode(ctypes.Structure):
    _fields_ = [
        ("nodes", ctypes.c_void_p), ("indices", ctypes.c_void_p), ("verts", ctypes.c_void_p), ("faces", ctypes.c_void_p),
        ("local_xform", ctypes.c_float * 16), ("lbmin", ctypes.c_float * 3), ("lbmax", ctypes.c_float * 3),
        ("xform", ctypes.c_float * 16), ("inv_xform", ctypes.c_float * 16), ("inv_xform33", ctypes.c_float * 16),
        ("xbmin", ctypes.c_float * 3), ("xbmax", ctypes.c_float * 3),
    ]


class SceneOracle:
    """C restatement of the two-level traversal (oracle/nanosg_oracle.c); per-node trees from `Oracle.build`."""

    def __init__(self, oracle=None):
        self.o = oracle or Oracle()
        L = self.o.L
        L.sgo_node_update.argtypes = [ctypes.c_void_p]
        L.sgo_traverse.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
        L.sgo_traverse_unordered_model.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32,
                                                   ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.sgo_sizeof_node.restype = ctypes.c_int
        assert L.sgo_sizeof_node() == ctypes.sizeof(_SgNode)
        self.L = L
        self.keep = []
        self.nodes = []

    def add_node(self, verts, faces, xform, tree=None):
        verts = np.ascontiguousarray(verts, dtype=np.float32)
        faces = np.ascontiguousarray(faces, dtype=np.uint32)
        if tree is None:
            nodes, idx, _ = self.o.build(verts, faces)
        else:
            nodes, idx = tree
        nodes = np.ascontiguousarray(nodes)
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        self.keep.append((verts, faces, nodes, idx))
        n = _SgNode()
        n.nodes, n.indices, n.verts, n.faces = nodes.ctypes.data, idx.ctypes.data, verts.ctypes.data, faces.ctypes.data
        x = np.ascontiguousarray(xform, dtype=np.float32).reshape(16)
        for i in range(16):
            n.local_xform[i] = float(x[i])
        for k in range(3):
            n.lbmin[k] = float(nodes[0]["bmin"][k])
            n.lbmax[k] = float(nodes[0]["bmax"][k])
        self.nodes.append(n)
        return len(self.nodes) - 1

    def commit(self):
        self.arr = (_SgNode * len(self.nodes))(*self.nodes)
        for i in range(len(self.nodes)):
            self.L.sgo_node_update(ctypes.byref(self.arr[i]))
        return True

    def node_state(self, i):
        n = self.arr[i]
        f = lambda a, shape: np.array(list(a), dtype=np.float32).reshape(shape)
        return {"xbmin": f(n.xbmin, 3), "xbmax": f(n.xbmax, 3), "inv_xform": f(n.inv_xform, (4, 4)),
                "inv_xform33": f(n.inv_xform33, (4, 4)), "xform": f(n.xform, (4, 4))}

    def traverse(self, rays):
        rays = np.ascontiguousarray(rays, dtype=ray_dtype(np.float32))
        hits = np.zeros((rays.shape[0],), dtype=SCENE_HIT)
        mask = np.zeros((rays.shape[0],), dtype=np.uint8)
        self.L.sgo_traverse(ctypes.cast(self.arr, ctypes.c_void_p), len(self.nodes), _p(rays), rays.shape[0], _p(hits), _p(mask))
        return hits, mask


    def traverse_unordered_model(self, rays, seed=1, roughly_front_to_back=False):
        """The single-pass walk's model (any visiting order, no list): (hits, mask, certified) — see nanosg_oracle.c."""
        rays = np.ascontiguousarray(rays, dtype=ray_dtype(np.float32))
        hits = np.zeros((rays.shape[0],), dtype=SCENE_HIT)
        mask = np.zeros((rays.shape[0],), dtype=np.uint8)
        cert = np.zeros((rays.shape[0],), dtype=np.uint8)
        self.L.sgo_traverse_unordered_model(ctypes.cast(self.arr, ctypes.c_void_p), len(self.nodes), _p(rays), rays.shape[0], seed,
                                            1 if roughly_front_to_back else 0, _p(hits), _p(mask), _p(cert))
        return hits, mask, cert


REF_V3_PATH = os.path.join(_HERE, "_ref", "libnanort_ref_v3.so")


class ReferenceV3(Reference):
    """The same shim compiled with -march=x86-64-v3 (AVX2 + FMA, contraction allowed): a stronger CPU TIMING baseline
    (SURVEY 8d asks for one); its results are not used for parity."""

    _lib = None
    PATH = REF_V3_PATH


def reference_v3_available():
    return os.path.exists(REF_V3_PATH)


# ---- sphere ("particle") custom primitive ----------------------------------------------------------------
REF_SPHERE_PATH = os.path.join(_HERE, "_ref", "libsphere_ref.so")
_HIT_F32 = hit_dtype(np.float32)
_NODE_F32 = node_dtype(np.float32)


def have_sphere_reference():
    return os.path.exists(REF_SPHERE_PATH)


class SphereOracle:
    """oracle/sphere_oracle.c: C restatement of the sphere intersector traced through Traverse."""

    def __init__(self):
        L = ctypes.CDLL(ORACLE_PATH)
        vp, u32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
        L.spo_traverse.argtypes = [vp, vp, vp, vp, vp, u64, u32, u32, vp, vp]
        L.spo_traverse.restype = None
        self.L = L

    def traverse(self, nodes, indices, centers, radii, rays, prim_ids_range=(0, 0x7FFFFFFF)):
        nodes = np.ascontiguousarray(nodes)
        indices = np.ascontiguousarray(indices, dtype=np.uint32)
        centers = np.ascontiguousarray(centers, dtype=np.float32)
        radii = np.ascontiguousarray(radii, dtype=np.float32)
        rays = np.ascontiguousarray(rays)
        n = rays.shape[0]
        hits = np.zeros(n, dtype=_HIT_F32)
        mask = np.zeros(n, dtype=np.uint8)
        self.L.spo_traverse(_p(nodes), _p(indices), _p(centers), _p(radii), _p(rays), n, prim_ids_range[0],
                            prim_ids_range[1], _p(hits), _p(mask))
        return hits, mask


class SphereReference:
    """The unmodified examples/particle_primitive classes over the unmodified nanort.h (oracle/ref_sphere_shim.cc)."""

    def __init__(self):
        L = ctypes.CDLL(REF_SPHERE_PATH)
        vp, u32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
        L.refsp_generate.argtypes = [vp, vp, u64, vp, vp]
        L.refsp_build.argtypes = [vp, vp, u32, vp, vp]
        L.refsp_build.restype = vp
        L.refsp_get_tree.argtypes = [vp, vp, vp]
        L.refsp_destroy.argtypes = [vp]
        L.refsp_traverse.argtypes = [vp, vp, u64, u32, u32, vp, vp]
        self.L = L
        self.h = None

    def __del__(self):
        try:
            if self.h:
                self.L.refsp_destroy(self.h)
        except Exception:
            pass

    def generate(self, n, bmin=(-1, -1, -1), bmax=(1, 1, 1)):
        centers = np.empty((n, 3), dtype=np.float32)
        radii = np.empty((n,), dtype=np.float32)
        lo, hi = np.asarray(bmin, dtype=np.float32), np.asarray(bmax, dtype=np.float32)
        self.L.refsp_generate(_p(centers), _p(radii), n, _p(lo), _p(hi))
        return centers, radii

    def build(self, centers, radii):
        centers = np.ascontiguousarray(centers, dtype=np.float32)
        radii = np.ascontiguousarray(radii, dtype=np.float32)
        nn = ctypes.c_uint32(0)
        stats = np.zeros(3, dtype=np.uint32)
        if self.h:
            self.L.refsp_destroy(self.h)
        self.h = self.L.refsp_build(_p(centers), _p(radii), radii.shape[0], ctypes.byref(nn), _p(stats))
        assert self.h
        nodes = np.zeros(nn.value, dtype=_NODE_F32)
        indices = np.zeros(radii.shape[0], dtype=np.uint32)
        self.L.refsp_get_tree(self.h, _p(nodes), _p(indices))
        return nodes, indices, {"max_tree_depth": int(stats[0]), "num_leaf_nodes": int(stats[1]), "num_branch_nodes": int(stats[2])}

    def traverse(self, rays, prim_ids_range=(0, 0x7FFFFFFF)):
        rays = np.ascontiguousarray(rays)
        n = rays.shape[0]
        hits = np.zeros(n, dtype=_HIT_F32)
        mask = np.zeros(n, dtype=np.uint8)
        self.L.refsp_traverse(self.h, _p(rays), n, prim_ids_range[0], prim_ids_range[1], _p(hits), _p(mask))
        return hits, mask


# ---- cylinder custom primitive ------------------------------------------------------------------------------
REF_CYLINDER_PATH = os.path.join(_HERE, "_ref", "libcylinder_ref.so")
CYL_HIT_F32 = np.dtype([("u", "<f4"), ("v", "<f4"), ("normal", "<f4", 3), ("t", "<f4"), ("prim_id", "<u4")])
assert CYL_HIT_F32.itemsize == 28


def have_cylinder_reference():
    return os.path.exists(REF_CYLINDER_PATH)


class CylinderOracle:
    """oracle/cylinder_oracle.c: C restatement of the cylinder intersector traced through Traverse."""

    def __init__(self):
        L = ctypes.CDLL(ORACLE_PATH)
        vp, u32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
        L.cyo_traverse.argtypes = [vp, vp, vp, vp, ctypes.c_int, vp, u64, u32, u32, vp, vp]
        L.cyo_traverse.restype = None
        self.L = L

    def traverse(self, nodes, indices, verts, radii, rays, prim_ids_range=(0, 0x7FFFFFFF), test_cap=True):
        nodes = np.ascontiguousarray(nodes)
        indices = np.ascontiguousarray(indices, dtype=np.uint32)
        verts = np.ascontiguousarray(verts, dtype=np.float32)
        radii = np.ascontiguousarray(radii, dtype=np.float32)
        rays = np.ascontiguousarray(rays)
        n = rays.shape[0]
        hits = np.zeros(n, dtype=CYL_HIT_F32)
        mask = np.zeros(n, dtype=np.uint8)
        self.L.cyo_traverse(_p(nodes), _p(indices), _p(verts), _p(radii), int(bool(test_cap)), _p(rays), n,
                            prim_ids_range[0], prim_ids_range[1], _p(hits), _p(mask))
        return hits, mask


class CylinderReference:
    """The unmodified examples/cylinder_primitive classes over the unmodified nanort.h (oracle/ref_cylinder_shim.cc)."""

    def __init__(self):
        L = ctypes.CDLL(REF_CYLINDER_PATH)
        vp, u32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
        L.refcy_generate.argtypes = [vp, vp, u64, vp, vp]
        L.refcy_build.argtypes = [vp, vp, u32, vp, vp]
        L.refcy_build.restype = vp
        L.refcy_get_tree.argtypes = [vp, vp, vp]
        L.refcy_destroy.argtypes = [vp]
        L.refcy_traverse.argtypes = [vp, vp, u64, u32, u32, ctypes.c_int, vp, vp]
        self.L = L
        self.h = None

    def __del__(self):
        try:
            if self.h:
                self.L.refcy_destroy(self.h)
        except Exception:
            pass

    def generate(self, n, bmin=(-1, -1, -1), bmax=(1, 1, 1)):
        verts = np.empty((n, 2, 3), dtype=np.float32)
        radii = np.empty((n, 2), dtype=np.float32)
        lo, hi = np.asarray(bmin, dtype=np.float32), np.asarray(bmax, dtype=np.float32)
        self.L.refcy_generate(_p(verts), _p(radii), n, _p(lo), _p(hi))
        return verts, radii

    def build(self, verts, radii):
        verts = np.ascontiguousarray(verts, dtype=np.float32)
        radii = np.ascontiguousarray(radii, dtype=np.float32)
        n = radii.size // 2
        nn = ctypes.c_uint32(0)
        stats = np.zeros(3, dtype=np.uint32)
        if self.h:
            self.L.refcy_destroy(self.h)
        self.h = self.L.refcy_build(_p(verts), _p(radii), n, ctypes.byref(nn), _p(stats))
        assert self.h
        nodes = np.zeros(nn.value, dtype=_NODE_F32)
        indices = np.zeros(n, dtype=np.uint32)
        self.L.refcy_get_tree(self.h, _p(nodes), _p(indices))
        return nodes, indices, {"max_tree_depth": int(stats[0]), "num_leaf_nodes": int(stats[1]), "num_branch_nodes": int(stats[2])}

    def traverse(self, rays, prim_ids_range=(0, 0x7FFFFFFF), test_cap=True):
        rays = np.ascontiguousarray(rays)
        n = rays.shape[0]
        hits = np.zeros(n, dtype=CYL_HIT_F32)
        mask = np.zeros(n, dtype=np.uint8)
        self.L.refcy_traverse(self.h, _p(rays), n, prim_ids_range[0], prim_ids_range[1], int(bool(test_cap)), _p(hits), _p(mask))
        return hits, mask
