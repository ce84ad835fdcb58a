"""Multi-GPU partition of the hot path: rays are independent and the BVH is read-only, so
each rank holds a replica of the tree (deterministic GPU build from the same mesh — no broadcast needed) and
traces a tile of the image; the only exchange is one gather of the 16-byte hit records (RCCL over xGMI when
the backend is "nccl").

Tiles are interleaved rows: rank r of `world` owns rows r, r+world, r+2*world, ... — every rank sees the same
mix of the scene, so the slowest rank is not the one looking at the dense part.
"""
import numpy as np


def rank_rows(height, rank, world):
    """Rows of a `height`-row image owned by `rank` (interleaved)."""
    return np.arange(rank, height, world)


def rows_per_rank(height, rank, world):
    return (height - rank + world - 1) // world


def gather_buffer(local_hits, world, rank, dst=0):
    """The root's receive buffer for gather_hit_records (None elsewhere): world equal slices, rank-major."""
    import torch

    if rank != dst:
        return None
    return torch.empty(world * local_hits.numel(), dtype=local_hits.dtype, device=local_hits.device)


def gather_hit_records(local_hits, world, rank, dist, out=None, dst=0, async_op=False):
    """Gather equal-sized per-rank hit buffers (torch uint8 tensors) to rank `dst`: the root receives every rank's
    records over its direct links at once, the others only send (an all-gather would move (world-1)x the data into
    every GPU for nothing).  `out`: the root's buffer from gather_buffer() (allocated here when None).
    Returns (out or None, work)."""
    if out is None:
        out = gather_buffer(local_hits, world, rank, dst)
    work = dist.gather(local_hits, gather_list=list(out.chunk(world)) if rank == dst else None, dst=dst, async_op=async_op)
    return out, work


def assemble_image(gathered, width, height, world, record_dtype):
    """Undo the interleaving: gathered = [rank0 rows | rank1 rows | ...] -> row-major image records.
    Requires height % world == 0 (equal tiles)."""
    rec = np.frombuffer(gathered, dtype=record_dtype) if not isinstance(gathered, np.ndarray) else gathered.view(record_dtype)
    per = height // world
    img = np.empty((height, width), dtype=record_dtype)
    for r in range(world):
        img[r::world] = rec[r * per * width:(r + 1) * per * width].reshape(per, width)
    return img.reshape(-1)


class RankedGroup:
    """One process per GPU over the C ABI (include/nanort_hip.h: nrtGroupCreateRanked / nrtGroupTraverseGather_*): this
    process owns tile `rank` of `world` (its interleaved rows), traces it where its rays live and the records of every tile
    reach `root`'s GPU by RCCL send / recv, in FRAME order (the de-interleaving runs on the root GPU).  `dist` (a
    torch.distributed module with an initialised process group) only hands the 128-byte RCCL id round.  Plumbing: ctypes."""

    def __init__(self, accel, rank, world, dist=None, device="cuda"):
        import ctypes

        import torch

        from . import capi

        self._L = L = capi.lib()
        vp, u32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
        L.nrtGroupUniqueId.argtypes = [vp, ctypes.c_size_t]
        L.nrtGroupCreateRanked.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)]
        L.nrtGroupDestroy.argtypes = [vp]
        L.nrtGroupDestroy.restype = None
        L.nrtGroupLastError.argtypes = [vp]
        L.nrtGroupLastError.restype = ctypes.c_char_p
        L.nrtGroupSetTunable.argtypes = [vp, ctypes.c_char_p, ctypes.c_longlong]
        L.nrtGroupTileRays.argtypes = [u64, u64, u32, u32]
        L.nrtGroupTileRays.restype = u64
        L.nrtGroupSynchronize.argtypes = [vp]
        L.nrtGroupLastTraffic.argtypes = [vp, ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(u64)]
        for sfx in ("f32", "f64"):
            getattr(L, "nrtGroupTraverseGather_" + sfx).argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(u64), u64, u64, vp, u32, vp, vp]
            getattr(L, "nrtGroupTraverseGatherTiles_" + sfx).argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(u64), u64, vp, u32, vp, vp]
        self.rank, self.world, self._sfx = rank, world, accel._s
        ident = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (ctypes.c_char * 128)()
            st = L.nrtGroupUniqueId(buf, 128)
            if st != capi.NRT_OK:
                raise capi.NrtError(st, L.nrtGroupLastError(None).decode())
            ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        if world > 1:
            if dist is None:
                raise ValueError("a world of more than one rank needs a process group to hand the RCCL id round")
            t = ident.to(device)
            dist.broadcast(t, src=0)
            ident = t.cpu()
        raw = ident.numpy().tobytes()
        g = vp()
        st = L.nrtGroupCreateRanked(accel._h, raw, rank, world, ctypes.byref(g))
        if st != capi.NRT_OK:
            raise capi.NrtError(st, L.nrtGroupLastError(None).decode())
        self._g, self._ct = g, ctypes

    def tile_rays(self, total_rays, row_len):
        return int(self._L.nrtGroupTileRays(total_rays, row_len, self.rank, self.world))

    def set_tunable(self, name, value):
        self._check(self._L.nrtGroupSetTunable(self._g, name.encode(), int(value)))

    def _check(self, st):
        from . import capi

        if st != capi.NRT_OK:
            raise capi.NrtError(st, self._L.nrtGroupLastError(self._g).decode())

    def traverse_gather(self, d_rays, count, total_rays, row_len, root=0, frame_hits=None, frame_mask=None):
        """d_rays: this rank's tile (torch uint8 tensor on its GPU, complete); frame_*: torch uint8 tensors on the root's GPU
        (root only).  Asynchronous: synchronize() before reading the frame."""
        ct = self._ct
        ptrs = (ct.c_void_p * 1)(d_rays.data_ptr() if count else None)
        counts = (ct.c_uint64 * 1)(count)
        f = getattr(self._L, "nrtGroupTraverseGather_" + self._sfx)
        self._check(f(self._g, ptrs, counts, total_rays, row_len, None, root,
                      frame_hits.data_ptr() if frame_hits is not None else None, frame_mask.data_ptr() if frame_mask is not None else None))

    def traverse_gather_tiles(self, d_rays, count, slot_rays, root=0, tiles_hits=None, tiles_mask=None):
        """Ragged waves: this rank's `count` <= slot_rays rays; the root receives every tile's whole slot, tile-major."""
        ct = self._ct
        ptrs = (ct.c_void_p * 1)(d_rays.data_ptr() if count else None)
        counts = (ct.c_uint64 * 1)(count)
        f = getattr(self._L, "nrtGroupTraverseGatherTiles_" + self._sfx)
        self._check(f(self._g, ptrs, counts, slot_rays, None, root,
                      tiles_hits.data_ptr() if tiles_hits is not None else None, tiles_mask.data_ptr() if tiles_mask is not None else None))

    def synchronize(self):
        self._check(self._L.nrtGroupSynchronize(self._g))

    def last_traffic(self):
        ct = self._ct
        a, b, c = ct.c_uint64(), ct.c_uint64(), ct.c_uint64()
        self._L.nrtGroupLastTraffic(self._g, ct.byref(a), ct.byref(b), ct.byref(c))
        return {"rccl": int(a.value), "peer": int(b.value), "in_place": int(c.value)}

    def close(self):
        if getattr(self, "_g", None):
            self._L.nrtGroupDestroy(self._g)
            self._g = None

    __del__ = close
