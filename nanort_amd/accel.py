"""Host-side mirror of the reference's operator interface for the hot path.

`BVHAccel` here plays the role of nanort::BVHAccel<T> (reference
nanort.h:698-860) for the built-in triangle plugin, with the same method names
and argument meaning; every call goes through the C ABI (include/nanort_hip.h)
into the HIP library.  `TraverseBatch` is the one addition: N rays per call
instead of the reference's one (`Traverse`, nanort.h:757-759).

    mesh  = TriangleMesh(vertices, faces, stride_bytes)   # nanort.h:925-930
    accel = BVHAccel(np.float32)
    accel.Build(mesh.num_faces, mesh, options)            # nanort.h:716-718
    hits, mask = accel.TraverseBatch(rays, trace_options) # N x Traverse

numpy arrays carry the reference's PODs (nanort_amd.wire).  torch is used only
by the *Device* variants (HBM-resident tensors, current stream).
"""
import ctypes

import numpy as np

from . import capi
from .wire import (
    BUILD_OPTIONS_F32,
    BUILD_OPTIONS_F64,
    BUILD_STATS,
    TRACE_OPTIONS,
    hit_dtype,
    node_dtype,
    ray_dtype,
    suffix,
)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class TriangleMesh:
    """nanort::TriangleMesh<T> (reference nanort.h:922-991): caller-owned flat arrays."""

    def __init__(self, vertices, faces, vertex_stride_bytes=None):
        self.vertices = np.ascontiguousarray(vertices)
        if self.vertices.dtype not in (np.float32, np.float64):
            raise TypeError("vertices must be float32 or float64")
        self.faces = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1, 3)
        self.vertex_stride_bytes = (
            3 * self.vertices.dtype.itemsize if vertex_stride_bytes is None else int(vertex_stride_bytes)
        )
        self.num_faces = int(self.faces.shape[0])

    def GetVertices(self):
        return self.vertices

    def GetFaces(self):
        return self.faces

    def GetVertexStrideBytes(self):
        return self.vertex_stride_bytes


class SphereGeometry:
    """The sphere ("particle") custom primitive of the reference (examples/particle_primitive/main.cc:82-291:
    SpherePred + SphereGeometry + SphereIntersector rolled into one description): xyz centres and one radius each."""

    def __init__(self, centers, radii):
        self.centers = np.ascontiguousarray(centers, dtype=np.float32).reshape(-1, 3)
        self.radii = np.ascontiguousarray(radii, dtype=np.float32).reshape(-1)
        if self.centers.shape[0] != self.radii.shape[0]:
            raise ValueError("one radius per centre")
        self.num_spheres = int(self.radii.shape[0])
        self.num_faces = self.num_spheres  # "number of primitives", under the name Build() checks


class CylinderGeometry:
    """The cylinder custom primitive of the reference (examples/cylinder_primitive/main.cc:94-424: CylinderPred +
    CylinderGeometry + CylinderIntersector): two end points (n, 2, 3) and two radii (n, 2) per cylinder, and the
    intersector's test_cap flag."""

    def __init__(self, endpoints, radii, test_cap=True):
        self.endpoints = np.ascontiguousarray(endpoints, dtype=np.float32).reshape(-1, 2, 3)
        self.radii = np.ascontiguousarray(radii, dtype=np.float32).reshape(-1, 2)
        if self.endpoints.shape[0] != self.radii.shape[0]:
            raise ValueError("two radii per cylinder")
        self.test_cap = bool(test_cap)
        self.num_faces = int(self.radii.shape[0])


class BVHAccel:
    """nanort::BVHAccel<T> on one MI355X (built-in triangle geometry, or the sphere / cylinder primitives in fp32)."""

    def __init__(self, real=np.float32, device=0):
        self.real = np.dtype(real)
        if self.real not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise TypeError("real must be float32 or float64")
        self._s = suffix(self.real)
        self._L = capi.lib()
        h = ctypes.c_void_p()
        st = self._L.nrtCreate(int(device), ctypes.byref(h))
        if st != capi.NRT_OK:
            raise capi.NrtError(st, self._L.nrtLastError(None).decode())
        self._h = h
        self.device = int(device)
        self._stats = np.zeros((), dtype=BUILD_STATS)
        self._mesh = None

    def close(self):
        if getattr(self, "_h", None):
            self._L.nrtDestroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != capi.NRT_OK:
            raise capi.NrtError(st, self._L.nrtLastError(self._h).decode())

    # -- mesh / build -------------------------------------------------------
    def SetMesh(self, mesh):
        if isinstance(mesh, CylinderGeometry):
            if self.real != np.float32:
                raise TypeError("cylinder primitives are fp32 (as the reference example)")
            self._check(self._L.nrtSetCylinders_f32(self._h, _p(mesh.endpoints), _p(mesh.radii), mesh.num_faces, int(mesh.test_cap)))
            self._mesh = mesh
            return
        if isinstance(mesh, SphereGeometry):
            if self.real != np.float32:
                raise TypeError("sphere primitives are fp32 (as the reference example)")
            self._check(self._L.nrtSetSpheres_f32(self._h, _p(mesh.centers), _p(mesh.radii), mesh.num_spheres))
            self._mesh = mesh
            return
        if mesh.vertices.dtype != self.real:
            raise TypeError("mesh precision %s != accel precision %s" % (mesh.vertices.dtype, self.real))
        self._check(
            getattr(self._L, "nrtSetMesh_" + self._s)(
                self._h, _p(mesh.vertices), mesh.vertex_stride_bytes, _p(mesh.faces), mesh.num_faces
            )
        )
        self._mesh = mesh

    def Build(self, num_primitives, mesh, options=None):
        """BVHAccel::Build (reference nanort.h:1892-2149). Returns False iff n == 0."""
        if num_primitives != mesh.num_faces:
            if isinstance(mesh, CylinderGeometry):
                mesh = CylinderGeometry(mesh.endpoints[:num_primitives], mesh.radii[:num_primitives], mesh.test_cap)
            elif isinstance(mesh, SphereGeometry):
                mesh = SphereGeometry(mesh.centers[:num_primitives], mesh.radii[:num_primitives])
            else:
                mesh = TriangleMesh(mesh.vertices, mesh.faces[:num_primitives], mesh.vertex_stride_bytes)
        self.SetMesh(mesh)
        if options is not None:
            want = BUILD_OPTIONS_F32 if self.real == np.float32 else BUILD_OPTIONS_F64
            options = np.asarray(options, dtype=want).reshape(1)
        nn = ctypes.c_uint64(0)
        st = getattr(self._L, "nrtBuild_" + self._s)(self._h, _p(options), _p(self._stats.reshape(1)), ctypes.byref(nn))
        if st == capi.NRT_ERR_EMPTY:
            return False
        self._check(st)
        return True

    def GetStatistics(self):
        return self._stats.copy()

    def LastBuildMs(self):
        return float(self._L.nrtLastBuildMs(self._h))

    def LastTraverseMs(self):
        return float(self._L.nrtLastTraverseMs(self._h))

    def SetLaunchTiming(self, on):
        """nrtSetLaunchTiming: on = bracket every launch with timing events (+ a completion event); off (the default) = no event
        in the stream, the kernel publishes a completion record whose stamps LastTraverseMs() reads."""
        self._check(self._L.nrtSetLaunchTiming(self._h, 1 if on else 0))

    def SetTunable(self, name, value):
        """nrtSetTunable: scheduling / layout tunables by name (include/nanort_hip.h lists them)."""
        self._check(self._L.nrtSetTunable(self._h, name.encode(), int(value)))

    def GetTunable(self, name):
        v = ctypes.c_longlong(0)
        self._check(self._L.nrtGetTunable(self._h, name.encode(), ctypes.byref(v)))
        return int(v.value)

    def LastKernelName(self):
        """The traversal kernel variant the most recent launch used (as rocprofv3 names it)."""
        return self._L.nrtLastKernelName(self._h).decode()

    def _tree_size(self):
        nn, ni = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(self._L.nrtTreeSize(self._h, ctypes.byref(nn), ctypes.byref(ni)))
        return int(nn.value), int(ni.value)

    def IsValid(self):
        return self._tree_size()[0] > 0

    def GetTree(self):
        nn, ni = self._tree_size()
        nodes = np.zeros((nn,), dtype=node_dtype(self.real))
        indices = np.zeros((ni,), dtype=np.uint32)
        if nn:
            self._check(getattr(self._L, "nrtGetTree_" + self._s)(self._h, _p(nodes), _p(indices)))
        return nodes, indices

    def GetNodes(self):
        return self.GetTree()[0]

    def GetIndices(self):
        return self.GetTree()[1]

    def BoundingBox(self):
        """BVHAccel::BoundingBox (reference nanort.h:792-804)."""
        nodes = self.GetNodes()
        if nodes.shape[0] == 0:
            m = np.finfo(self.real).max
            return np.full(3, m, self.real), np.full(3, -m, self.real)
        return nodes[0]["bmin"].copy(), nodes[0]["bmax"].copy()

    def SetTree(self, nodes, indices):
        """Adopt a tree built elsewhere (BVHAccel::Load, reference nanort.h:2219-2275)."""
        nodes = np.ascontiguousarray(nodes, dtype=node_dtype(self.real))
        indices = np.ascontiguousarray(indices, dtype=np.uint32)
        self._check(
            getattr(self._L, "nrtSetTree_" + self._s)(self._h, _p(nodes), nodes.shape[0], _p(indices), indices.shape[0])
        )

    # -- traverse -----------------------------------------------------------
    def TraverseBatch(self, rays, options=None):
        """N x BVHAccel::Traverse (reference nanort.h:2487-2556). Returns (hits, mask)."""
        rays = np.ascontiguousarray(rays, dtype=ray_dtype(self.real))
        n = rays.shape[0]
        mask = np.zeros((n,), dtype=np.uint8)
        if options is not None:
            options = np.asarray(options, dtype=TRACE_OPTIONS).reshape(1)
        if isinstance(self._mesh, CylinderGeometry):  # the example's 28-byte CylinderIntersection records
            from .wire import CYL_HIT_F32

            hits = np.zeros((n,), dtype=CYL_HIT_F32)
            self._check(self._L.nrtTraverseBatchCylinders_f32(self._h, _p(rays), n, _p(options), _p(hits), _p(mask)))
            return hits, mask
        hits = np.zeros((n,), dtype=hit_dtype(self.real))
        self._check(
            getattr(self._L, "nrtTraverseBatch_" + self._s)(self._h, _p(rays), n, _p(options), _p(hits), _p(mask))
        )
        return hits, mask

    def TraverseBatchDevice(self, d_rays, d_hits, d_mask=None, options=None, stream=None):
        """Same on HBM-resident torch uint8 tensors (raw PODs), async on `stream`
        (default: torch's current stream)."""
        import torch

        cyl = isinstance(self._mesh, CylinderGeometry)
        rsz, hsz = ray_dtype(self.real).itemsize, (28 if cyl else hit_dtype(self.real).itemsize)
        n = d_rays.numel() * d_rays.element_size() // rsz
        assert d_hits.numel() * d_hits.element_size() >= n * hsz
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        if options is not None:
            options = np.asarray(options, dtype=TRACE_OPTIONS).reshape(1)
        if cyl:
            self._check(self._L.nrtTraverseBatchCylindersDevice_f32(
                self._h, d_rays.data_ptr(), n, _p(options), d_hits.data_ptr(), None if d_mask is None else d_mask.data_ptr(), stream))
            return n
        self._check(
            getattr(self._L, "nrtTraverseBatchDevice_" + self._s)(
                self._h, d_rays.data_ptr(), n, _p(options), d_hits.data_ptr(),
                None if d_mask is None else d_mask.data_ptr(), stream,
            )
        )
        return n

    def TraverseBatchesDevice(self, batches, options=None, stream=None):
        """nrtTraverseBatchesDevice: several independent batches — a list of (d_rays, d_hits, d_mask or None, n or None[, "occlusion"])
        torch uint8 tensors — walked by ONE persistent launch (asynchronous on `stream`).  A batch marked "occlusion" is an
        occlusion query: only its d_mask is written (d_hits may be None).  Returns the ray counts."""
        import torch

        rsz, hsz = ray_dtype(self.real).itemsize, hit_dtype(self.real).itemsize
        nb = len(batches)
        rays = (ctypes.c_void_p * nb)()
        hits = (ctypes.c_void_p * nb)()
        masks = (ctypes.c_void_p * nb)()
        counts = (ctypes.c_uint64 * nb)()
        flags = (ctypes.c_uint32 * nb)()
        for k, b in enumerate(batches):
            d_rays, d_hits, d_mask = b[0], b[1], b[2]
            n = b[3] if len(b) > 3 and b[3] is not None else d_rays.numel() * d_rays.element_size() // rsz
            occ = len(b) > 4 and b[4] == "occlusion"
            assert occ or d_hits.numel() * d_hits.element_size() >= n * hsz
            flags[k] = 1 if occ else 0
            rays[k], hits[k], counts[k] = d_rays.data_ptr(), (None if d_hits is None else d_hits.data_ptr()), n
            masks[k] = None if d_mask is None else d_mask.data_ptr()
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        if options is not None:
            options = np.asarray(options, dtype=TRACE_OPTIONS).reshape(1)
        self._check(getattr(self._L, "nrtTraverseBatchesDevice_" + self._s)(self._h, nb, rays, counts, _p(options), hits, masks, flags, stream))
        return [int(c) for c in counts]

    def TraverseBatches(self, batches, options=None):
        """nrtTraverseBatches: several independent HOST batches — a list of ray arrays or (rays, "occlusion") pairs — uploaded
        together, walked by ONE persistent launch, downloaded together.  Returns a list of (hits, mask) per closest-hit batch
        and (None, mask) per occlusion batch: exactly what TraverseBatch / OccludedBatch return for each of them."""
        RAY, HIT = ray_dtype(self.real), hit_dtype(self.real)
        nb = len(batches)
        rays = (ctypes.c_void_p * nb)()
        hits = (ctypes.c_void_p * nb)()
        masks = (ctypes.c_void_p * nb)()
        counts = (ctypes.c_uint64 * nb)()
        flags = (ctypes.c_uint32 * nb)()
        keep, out = [], []
        for k, b in enumerate(batches):
            occ = isinstance(b, tuple) and len(b) > 1 and b[1] == "occlusion"
            r = np.ascontiguousarray(b[0] if isinstance(b, tuple) else b, dtype=RAY)
            h = None if occ else np.zeros((r.shape[0],), dtype=HIT)
            m = np.zeros((r.shape[0],), dtype=np.uint8)
            keep.append(r)
            out.append((h, m))
            rays[k], counts[k], flags[k] = (r.ctypes.data if r.shape[0] else None), r.shape[0], (1 if occ else 0)
            hits[k] = None if (occ or not r.shape[0]) else h.ctypes.data
            masks[k] = m.ctypes.data if r.shape[0] else None
        if options is not None:
            options = np.asarray(options, dtype=TRACE_OPTIONS).reshape(1)
        self._check(getattr(self._L, "nrtTraverseBatches_" + self._s)(self._h, nb, rays, counts, _p(options), hits, masks, flags))
        return out

    def OccludedBatch(self, rays, options=None):
        """Opt-in extension: only the hit flags of TraverseBatch(), each ray stopping at the first primitive it accepts."""
        rays = np.ascontiguousarray(rays, dtype=ray_dtype(self.real))
        mask = np.zeros((rays.shape[0],), dtype=np.uint8)
        if options is not None:
            options = np.asarray(options, dtype=TRACE_OPTIONS).reshape(1)
        self._check(getattr(self._L, "nrtOccludedBatch_" + self._s)(self._h, _p(rays), rays.shape[0], _p(options), _p(mask)))
        return mask

    def OccludedBatchDevice(self, d_rays, d_mask, options=None, stream=None):
        import torch

        n = d_rays.numel() * d_rays.element_size() // ray_dtype(self.real).itemsize
        assert d_mask.numel() >= n
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        if options is not None:
            options = np.asarray(options, dtype=TRACE_OPTIONS).reshape(1)
        self._check(getattr(self._L, "nrtOccludedBatchDevice_" + self._s)(self._h, d_rays.data_ptr(), n, _p(options), d_mask.data_ptr(), stream))
        return n

    def TraverseCountDevice(self, d_rays, options=None):
        """Work counters (nodes visited, leaves, triangle tests, max stack) of one batch."""
        rsz = ray_dtype(self.real).itemsize
        n = d_rays.numel() * d_rays.element_size() // rsz
        c = capi.TraceCounters()
        if options is not None:
            options = np.asarray(options, dtype=TRACE_OPTIONS).reshape(1)
        self._check(
            getattr(self._L, "nrtTraverseCountDevice_" + self._s)(self._h, d_rays.data_ptr(), n, _p(options), ctypes.byref(c))
        )
        return {
            "nodes_visited": int(c.nodes_visited),
            "leaves_tested": int(c.leaves_tested),
            "tris_tested": int(c.tris_tested),
            "max_stack": int(c.max_stack),
            "num_rays": int(n),
        }


class Scene:
    """nanosg::Scene<float, M> on one MI355X (reference examples/nanosg/nanosg.h:668-905): instanced two-level
    traversal.  Nodes are built `BVHAccel(np.float32)` objects plus nanosg's 4x4 local transform (row 3 =
    translation); `Traverse` of the reference becomes `TraverseBatch`."""

    def __init__(self, device=0):
        self._L = capi.lib()
        h = ctypes.c_void_p()
        st = self._L.nrtSceneCreate(int(device), ctypes.byref(h))
        if st != capi.NRT_OK:
            raise capi.NrtError(st, self._L.nrtSceneLastError(None).decode())
        self._h = h
        self._keep = []

    def close(self):
        if getattr(self, "_h", None):
            self._L.nrtSceneDestroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != capi.NRT_OK:
            raise capi.NrtError(st, self._L.nrtSceneLastError(self._h).decode())

    def AddNode(self, accel, local_xform):
        x = np.ascontiguousarray(local_xform, dtype=np.float32).reshape(16)
        nid = ctypes.c_uint32(0)
        self._check(self._L.nrtSceneAddNode_f32(self._h, accel._h, _p(x), ctypes.byref(nid)))
        self._keep.append(accel)
        return int(nid.value)

    def Commit(self):
        st = self._L.nrtSceneCommit(self._h)
        if st == capi.NRT_ERR_EMPTY:
            return False
        self._check(st)
        return True

    def GetBoundingBox(self):
        """Scene::GetBoundingBox (reference nanosg.h:761-769): (bmin, bmax) of a committed scene."""
        bmin = np.zeros(3, dtype=np.float32)
        bmax = np.zeros(3, dtype=np.float32)
        self._check(self._L.nrtSceneBounds_f32(self._h, _p(bmin), _p(bmax)))
        return bmin, bmax

    def NodeState(self, node_id):
        """xform, inv_xform, inv_xform33, inv_transpose_xform33 of a committed node (reference nanosg.h:397-437)."""
        out = np.zeros(64, dtype=np.float32)
        self._check(self._L.nrtSceneNodeState_f32(self._h, int(node_id), _p(out)))
        m = out.reshape(4, 4, 4)
        return {"xform": m[0], "inv_xform": m[1], "inv_xform33": m[2], "inv_transpose_xform33": m[3]}

    def SetTunable(self, name, value):
        """Scheduling knobs of the scene kernels (nrtSceneSetTunable): "single_pass", "trav_min", "refill_min", ..."""
        self._check(self._L.nrtSceneSetTunable(self._h, name.encode(), int(value)))

    def LastRedone(self):
        """Rays of the last call that the single-pass walk handed to the listing path (nrtSceneLastRedone)."""
        return int(self._L.nrtSceneLastRedone(self._h))

    def LastPath(self):
        """1: the last call went through the single-pass walk, 0: the listing path alone (nrtSceneLastPath)."""
        return int(self._L.nrtSceneLastPath(self._h))

    def TraverseBatch(self, rays):
        from .wire import RAY_F32, SCENE_HIT_F32

        rays = np.ascontiguousarray(rays, dtype=RAY_F32)
        hits = np.zeros((rays.shape[0],), dtype=SCENE_HIT_F32)
        mask = np.zeros((rays.shape[0],), dtype=np.uint8)
        self._check(self._L.nrtSceneTraverseBatch_f32(self._h, _p(rays), rays.shape[0], _p(hits), _p(mask)))
        return hits, mask

    def TraverseBatchDevice(self, d_rays, d_hits, d_mask=None):
        """Rays and results in HBM: torch uint8 tensors holding RAY_F32 records in, SCENE_HIT_F32 records (20 B) and the
        optional hit flags out.  Synchronous (see nrtSceneTraverseBatchDevice_f32); waits for torch's current stream first."""
        import torch

        from .wire import RAY_F32, SCENE_HIT_F32

        n = d_rays.numel() // RAY_F32.itemsize
        assert d_rays.is_cuda and d_hits.is_cuda and d_hits.numel() >= n * SCENE_HIT_F32.itemsize
        torch.cuda.current_stream().synchronize()
        self._check(self._L.nrtSceneTraverseBatchDevice_f32(self._h, d_rays.data_ptr(), n, d_hits.data_ptr(),
                                                            d_mask.data_ptr() if d_mask is not None else None))
