"""Wire formats of the hot path, as numpy structured dtypes.

Layout-identical to the reference's PODs (all `file:line` relative to the
reference tree):

  RAY_*    nanort::Ray<T>                   nanort.h:474-496   36 B / 72 B
  NODE_*   nanort::BVHNode<T>               nanort.h:498-550   40 B / 64 B
  HIT_*    nanort::TriangleIntersection<T>  nanort.h:996-1005  16 B / 32 B
  TRACE_OPTIONS  nanort::BVHTraceOptions    nanort.h:604-624   16 B
  BUILD_OPTIONS_* nanort::BVHBuildOptions<T> nanort.h:559-583  28 B / 32 B
  BUILD_STATS    nanort::BVHBuildStatistics nanort.h:586-599   16 B

and to the C structs in include/nanort_hip.h (static_asserts there pin the
sizes; tests/test_capi.py pins these dtypes against both).
"""
import numpy as np

RAY_F32 = np.dtype(
    [("org", "<f4", 3), ("dir", "<f4", 3), ("min_t", "<f4"), ("max_t", "<f4"), ("type", "<u4")]
)
RAY_F64 = np.dtype(
    {
        "names": ["org", "dir", "min_t", "max_t", "type"],
        "formats": [("<f8", 3), ("<f8", 3), "<f8", "<f8", "<u4"],
        "offsets": [0, 24, 48, 56, 64],
        "itemsize": 72,
    }
)
NODE_F32 = np.dtype(
    [("bmin", "<f4", 3), ("bmax", "<f4", 3), ("flag", "<i4"), ("axis", "<i4"), ("data", "<u4", 2)]
)
NODE_F64 = np.dtype(
    [("bmin", "<f8", 3), ("bmax", "<f8", 3), ("flag", "<i4"), ("axis", "<i4"), ("data", "<u4", 2)]
)
HIT_F32 = np.dtype([("u", "<f4"), ("v", "<f4"), ("t", "<f4"), ("prim_id", "<u4")])
HIT_F64 = np.dtype(
    {
        "names": ["u", "v", "t", "prim_id"],
        "formats": ["<f8", "<f8", "<f8", "<u4"],
        "offsets": [0, 8, 16, 24],
        "itemsize": 32,
    }
)
TRACE_OPTIONS = np.dtype(
    [("prim_ids_range", "<u4", 2), ("skip_prim_id", "<u4"), ("cull_back_face", "u1"), ("pad", "u1", 3)]
)
BUILD_OPTIONS_F32 = np.dtype(
    [
        ("cost_t_aabb", "<f4"),
        ("min_leaf_primitives", "<u4"),
        ("max_tree_depth", "<u4"),
        ("bin_size", "<u4"),
        ("shallow_depth", "<u4"),
        ("min_primitives_for_parallel_build", "<u4"),
        ("cache_bbox", "u1"),
        ("pad", "u1", 3),
    ]
)
BUILD_OPTIONS_F64 = np.dtype(
    [
        ("cost_t_aabb", "<f8"),
        ("min_leaf_primitives", "<u4"),
        ("max_tree_depth", "<u4"),
        ("bin_size", "<u4"),
        ("shallow_depth", "<u4"),
        ("min_primitives_for_parallel_build", "<u4"),
        ("cache_bbox", "u1"),
        ("pad", "u1", 3),
    ]
)
BUILD_STATS = np.dtype(
    [("max_tree_depth", "<u4"), ("num_leaf_nodes", "<u4"), ("num_branch_nodes", "<u4"), ("build_secs", "<f4")]
)

assert RAY_F32.itemsize == 36 and RAY_F64.itemsize == 72
assert NODE_F32.itemsize == 40 and NODE_F64.itemsize == 64
assert HIT_F32.itemsize == 16 and HIT_F64.itemsize == 32
assert TRACE_OPTIONS.itemsize == 16 and BUILD_STATS.itemsize == 16
assert BUILD_OPTIONS_F32.itemsize == 28 and BUILD_OPTIONS_F64.itemsize == 32

# nanosg::Intersection<float> fields the two-level traversal fills (reference examples/nanosg/nanosg.h:307-318)
SCENE_HIT_F32 = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("prim_id", "<u4"), ("node_id", "<u4")])
assert SCENE_HIT_F32.itemsize == 20

# CylinderIntersection of the reference's cylinder example (examples/cylinder_primitive/main.cc:213-224)
CYL_HIT_F32 = np.dtype([("u", "<f4"), ("v", "<f4"), ("normal", "<f4", 3), ("t", "<f4"), ("prim_id", "<u4")])
assert CYL_HIT_F32.itemsize == 28

MISS_PRIM_ID = 0xFFFFFFFF


def ray_dtype(real):
    return RAY_F32 if np.dtype(real) == np.float32 else RAY_F64


def node_dtype(real):
    return NODE_F32 if np.dtype(real) == np.float32 else NODE_F64


def hit_dtype(real):
    return HIT_F32 if np.dtype(real) == np.float32 else HIT_F64


def suffix(real):
    return "f32" if np.dtype(real) == np.float32 else "f64"


def default_trace_options():
    """BVHTraceOptions() defaults — reference nanort.h:617-623."""
    o = np.zeros((), dtype=TRACE_OPTIONS)
    o["prim_ids_range"] = (0, 0x7FFFFFFF)
    o["skip_prim_id"] = 0xFFFFFFFF
    o["cull_back_face"] = 0
    return o


def default_build_options(real=np.float32):
    """BVHBuildOptions<T>() defaults — reference nanort.h:574-582."""
    o = np.zeros((), dtype=BUILD_OPTIONS_F32 if np.dtype(real) == np.float32 else BUILD_OPTIONS_F64)
    o["cost_t_aabb"] = 0.2
    o["min_leaf_primitives"] = 4
    o["max_tree_depth"] = 256
    o["bin_size"] = 64
    o["shallow_depth"] = 4
    o["min_primitives_for_parallel_build"] = 1024 * 8
    o["cache_bbox"] = 0
    return o


def widen_rays(rays_f32):
    """fp32 ray buffer -> fp64 ray buffer with the same values (config C5)."""
    out = np.zeros(rays_f32.shape, dtype=RAY_F64)
    for k in ("org", "dir", "min_t", "max_t"):
        out[k] = rays_f32[k].astype(np.float64)
    out["type"] = rays_f32["type"]
    return out
