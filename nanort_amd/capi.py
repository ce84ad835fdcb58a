"""ctypes view of the C ABI in include/nanort_hip.h (libnanort_hip.so).

Plumbing only: argument marshalling and a loud failure when the HIP library
is missing.  There is no CPU fallback of any kind behind these calls.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (NRT_USE_PROF_LIB=1: the profiling build of the same sources — include/nanort_hip_prof.h — for tools/ that read loop counters
# or per-wave time stamps; never set by the package, the tests or bench.py)
LIB_PATH = os.path.join(_HERE, "lib", "libnanort_hip_prof.so" if os.environ.get("NRT_USE_PROF_LIB") == "1" else "libnanort_hip.so")

NRT_OK, NRT_ERR_INVALID, NRT_ERR_EMPTY, NRT_ERR_DEVICE, NRT_ERR_PRECISION = 0, 1, 2, 3, 4

# Every symbol include/nanort_hip.h declares (tests/test_capi.py checks the
# header and this table against the built library).
SYMBOLS = [
    "nrtCreate", "nrtDestroy", "nrtLastError", "nrtVersion",
    "nrtSetMesh_f32", "nrtSetMesh_f64", "nrtSetSpheres_f32",
    "nrtSetCylinders_f32", "nrtTraverseBatchCylinders_f32", "nrtTraverseBatchCylindersDevice_f32",
    "nrtBuild_f32", "nrtBuild_f64",
    "nrtGetTree_f32", "nrtGetTree_f64", "nrtTreeSize", "nrtGetTreeBounds_f32", "nrtGetTreeBounds_f64",
    "nrtSetTree_f32", "nrtSetTree_f64",
    "nrtTraverseBatch_f32", "nrtTraverseBatch_f64",
    "nrtTraverseBatchDevice_f32", "nrtTraverseBatchDevice_f64", "nrtTraverseBatchesDevice_f32", "nrtTraverseBatchesDevice_f64", "nrtTraverseBatches_f32", "nrtTraverseBatches_f64",
    "nrtTraverseBatchMulti_f32", "nrtTraverseBatchMulti_f64", "nrtDeviceCount",
    "nrtTraverseCountDevice_f32", "nrtTraverseCountDevice_f64",
    "nrtOccludedBatch_f32", "nrtOccludedBatch_f64", "nrtOccludedBatchDevice_f32", "nrtOccludedBatchDevice_f64",
    "nrtLastTraverseMs", "nrtSetLaunchTiming", "nrtSetTunable", "nrtGetTunable", "nrtLastBuildMs", "nrtLastKernelName", "nrtHostAlloc", "nrtHostFree",
    "nrtGroupUniqueId", "nrtGroupCreate", "nrtGroupCreateRanked", "nrtGroupDestroy", "nrtGroupLastError", "nrtGroupSetTunable", "nrtGroupInfo",
    "nrtGroupTileRays", "nrtGroupTraverseGather_f32", "nrtGroupTraverseGather_f64", "nrtGroupTraverseGatherTiles_f32", "nrtGroupTraverseGatherTiles_f64", "nrtGroupSynchronize", "nrtGroupLastTraffic",
    "nrtSceneCreate", "nrtSceneDestroy", "nrtSceneLastError", "nrtSceneAddNode_f32", "nrtSceneCommit", "nrtSceneNodeState_f32",
    "nrtSceneBounds_f32", "nrtSceneTraverseBatch_f32", "nrtSceneTraverseBatchDevice_f32", "nrtSceneSetTunable", "nrtSceneLastRedone", "nrtSceneLastPath",
]


class NrtError(RuntimeError):
    def __init__(self, status, message):
        RuntimeError.__init__(self, "nanort_hip status %d: %s" % (status, message))
        self.status = status


class TraceCounters(ctypes.Structure):
    _fields_ = [
        ("nodes_visited", ctypes.c_uint64),
        ("leaves_tested", ctypes.c_uint64),
        ("tris_tested", ctypes.c_uint64),
        ("max_stack", ctypes.c_uint64),
    ]


_LIB = None


def lib():
    """Load libnanort_hip.so (once). Raises if it was not built — never falls back."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "%s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C nanort_amd/csrc`. nanort_amd has no CPU fallback." % LIB_PATH
        )
    try:
        # When torch is in the process its bundled HIP runtime (same SONAME,
        # libamdhip64.so.7) must be the one we bind to, so device pointers and
        # streams are shared.  Importing it first makes the loader reuse it.
        import torch  # noqa: F401
    except Exception:
        pass
    L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    vp, u32, u64, sz, i32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_int
    L.nrtCreate.argtypes = [i32, ctypes.POINTER(vp)]
    L.nrtCreate.restype = i32
    L.nrtDestroy.argtypes = [vp]
    L.nrtDestroy.restype = None
    L.nrtLastError.argtypes = [vp]
    L.nrtLastError.restype = ctypes.c_char_p
    L.nrtVersion.argtypes = []
    L.nrtVersion.restype = ctypes.c_char_p
    L.nrtTreeSize.argtypes = [vp, ctypes.POINTER(u64), ctypes.POINTER(u64)]
    L.nrtTreeSize.restype = i32
    for s in ("f32", "f64"):
        f = getattr(L, "nrtSetMesh_" + s)
        f.argtypes = [vp, vp, sz, vp, u32]
        f.restype = i32
        f = getattr(L, "nrtBuild_" + s)
        f.argtypes = [vp, vp, vp, ctypes.POINTER(u64)]
        f.restype = i32
        f = getattr(L, "nrtGetTree_" + s)
        f.argtypes = [vp, vp, vp]
        f.restype = i32
        f = getattr(L, "nrtSetTree_" + s)
        f.argtypes = [vp, vp, u64, vp, u64]
        f.restype = i32
        f = getattr(L, "nrtTraverseBatch_" + s)
        f.argtypes = [vp, vp, u64, vp, vp, vp]
        f.restype = i32
        f = getattr(L, "nrtTraverseBatchDevice_" + s)
        f.argtypes = [vp, vp, u64, vp, vp, vp, vp]
        f.restype = i32
        f = getattr(L, "nrtTraverseBatchMulti_" + s)
        f.argtypes = [vp, u32, vp, u64, u64, vp, vp, vp]
        f.restype = i32
        f = getattr(L, "nrtTraverseBatchesDevice_" + s)
        f.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp, vp]
        f.restype = i32
        f = getattr(L, "nrtTraverseBatches_" + s)
        f.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp]
        f.restype = i32
        f = getattr(L, "nrtTraverseCountDevice_" + s)
        f.argtypes = [vp, vp, u64, vp, ctypes.POINTER(TraceCounters)]
        f.restype = i32
    for sfx in ("f32", "f64"):
        f = getattr(L, "nrtOccludedBatch_" + sfx)
        f.argtypes = [vp, vp, u64, vp, vp]
        f.restype = i32
        f = getattr(L, "nrtOccludedBatchDevice_" + sfx)
        f.argtypes = [vp, vp, u64, vp, vp, vp]
        f.restype = i32
    L.nrtSetSpheres_f32.argtypes = [vp, vp, vp, u32]
    L.nrtSetSpheres_f32.restype = i32
    L.nrtSetCylinders_f32.argtypes = [vp, vp, vp, u32, i32]
    L.nrtSetCylinders_f32.restype = i32
    L.nrtTraverseBatchCylinders_f32.argtypes = [vp, vp, u64, vp, vp, vp]
    L.nrtTraverseBatchCylinders_f32.restype = i32
    L.nrtTraverseBatchCylindersDevice_f32.argtypes = [vp, vp, u64, vp, vp, vp, vp]
    L.nrtTraverseBatchCylindersDevice_f32.restype = i32
    L.nrtSceneCreate.argtypes = [i32, ctypes.POINTER(vp)]
    L.nrtSceneCreate.restype = i32
    L.nrtSceneDestroy.argtypes = [vp]
    L.nrtSceneDestroy.restype = None
    L.nrtSceneLastError.argtypes = [vp]
    L.nrtSceneLastError.restype = ctypes.c_char_p
    L.nrtSceneAddNode_f32.argtypes = [vp, vp, vp, ctypes.POINTER(u32)]
    L.nrtSceneAddNode_f32.restype = i32
    L.nrtSceneCommit.argtypes = [vp]
    L.nrtSceneNodeState_f32.argtypes = [vp, u32, vp]
    L.nrtSceneNodeState_f32.restype = i32
    L.nrtSceneCommit.restype = i32
    L.nrtSceneBounds_f32.argtypes = [vp, vp, vp]
    L.nrtSceneBounds_f32.restype = i32
    L.nrtSceneTraverseBatch_f32.argtypes = [vp, vp, u64, vp, vp]
    L.nrtSceneTraverseBatch_f32.restype = i32
    L.nrtSceneTraverseBatchDevice_f32.argtypes = [vp, vp, u64, vp, vp]
    L.nrtSceneTraverseBatchDevice_f32.restype = i32
    L.nrtSceneSetTunable.argtypes = [vp, ctypes.c_char_p, i32]
    L.nrtSceneSetTunable.restype = i32
    L.nrtSceneLastRedone.argtypes = [vp]
    L.nrtSceneLastRedone.restype = u64
    L.nrtSceneLastPath.argtypes = [vp]
    L.nrtSceneLastPath.restype = i32
    L.nrtHostAlloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(vp)]
    L.nrtHostAlloc.restype = i32
    L.nrtHostFree.argtypes = [vp]
    L.nrtHostFree.restype = None
    L.nrtSetLaunchTiming.argtypes = [vp, ctypes.c_int]
    L.nrtSetLaunchTiming.restype = ctypes.c_int
    L.nrtSetTunable.argtypes = [vp, ctypes.c_char_p, ctypes.c_longlong]
    L.nrtSetTunable.restype = i32
    L.nrtGetTunable.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_longlong)]
    L.nrtGetTunable.restype = i32
    L.nrtLastTraverseMs.argtypes = [vp]
    L.nrtLastTraverseMs.restype = ctypes.c_float
    L.nrtLastBuildMs.argtypes = [vp]
    L.nrtLastBuildMs.restype = ctypes.c_float
    L.nrtDeviceCount.argtypes = []
    L.nrtDeviceCount.restype = i32
    L.nrtLastKernelName.argtypes = [vp]
    L.nrtLastKernelName.restype = ctypes.c_char_p
    _LIB = L
    return L
