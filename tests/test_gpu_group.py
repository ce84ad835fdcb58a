"""nrtGroup*: the C-ABI multi-GPU path for DEVICE-RESIDENT rays — row-interleaved tiles traced
by one context each, the records gathered to the root GPU (RCCL send / recv, or peer copies) and put into frame order by a kernel
on the root.  On the one-GPU box the tiles' contexts share device 0: the split, the exchange (`self_send` = 1 sends the root
tile's records through ncclSend / ncclRecv on a one-rank communicator), the frame-order kernel, buffer reuse over several frames
and the error paths are the code an 8-GPU node runs.  Driven from C++ (tests/cpp/group_check.cc) and, once, from a process that
has torch's own RCCL loaded (the library binds to the copy already in the process)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from nanort_amd import BVHAccel, TriangleMesh, capi, scenes
from nanort_amd.wire import hit_dtype, widen_rays

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "nanort_amd", "lib")


@pytest.fixture(scope="module")
def group_check(tmp_path_factory):
    d = tmp_path_factory.mktemp("group")
    exe = d / "group_check"
    r = subprocess.run(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-D__HIP_PLATFORM_AMD__", "-I", INC, "-isystem", "/opt/rocm/include",
                        os.path.join(ROOT, "tests", "cpp", "group_check.cc"), "-o", str(exe), "-L", LIBDIR, "-lnanort_hip",
                        "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return str(exe), d


def write_inputs(d, v, f, rays, tag):
    mesh, rp = os.path.join(str(d), "mesh_%s.bin" % tag), os.path.join(str(d), "rays_%s.bin" % tag)
    with open(mesh, "wb") as fp:
        fp.write(np.array([v.shape[0], f.shape[0]], dtype=np.uint32).tobytes())
        fp.write(np.ascontiguousarray(v).tobytes())
        fp.write(np.ascontiguousarray(f).tobytes())
    with open(rp, "wb") as fp:
        fp.write(np.array([rays.shape[0]], dtype=np.uint64).tobytes())
        fp.write(np.ascontiguousarray(rays).tobytes())
    return mesh, rp


def run(exe, args):
    r = subprocess.run([exe] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    return r.returncode, r.stdout


@pytest.mark.parametrize("f64", [False, True])
def test_cpp_host_gathers_the_frame_of_row_interleaved_tiles(group_check, f64):
    exe, d = group_check
    v, f = scenes.sphere(96, 48)
    W, H = 331, 67  # ragged: 67 rows over 2 / 3 / 8 tiles, and a short last row when row_len does not divide the frame
    rays = scenes.camera_rays(W, H)
    if f64:
        v, rays = v.astype(np.float64), widen_rays(rays)
    mesh, rp = write_inputs(d, v, f, rays, "f64" if f64 else "f32")
    real = "f64" if f64 else "f32"
    hit_b = 32 if f64 else 16
    n = rays.shape[0]
    for tiles, row_len, extra in ((2, W, []), (3, 1000, []), (8, W, ["root=5"]), (1, W, []), (2, W, ["transport=1"]),
                                  (2, W, ["self_send=1"]), (3, W, ["self_send=1", "root=2"]), (2, W, ["transport=1", "self_send=1"])):
        rc, out = run(exe, [real, mesh, rp, tiles, row_len] + extra)
        assert rc == 0 and "frame_mismatches 0 " in out and "tile_slot_mismatches 0 " in out and "slot_overflow_status 1" in out, (tiles, row_len, extra, out)
        assert "rays %d " % n in out and "wrong_count_status 1 " in out and "interleaved rows hold" in out, out
        line = [ln for ln in out.splitlines() if ln.startswith("rays ")][0].split()
        moved = {line[i]: int(line[i + 1]) for i in range(0, len(line), 2)}
        root = int([e for e in extra if e.startswith("root=")][0][5:]) if any(e.startswith("root=") for e in extra) else 0
        rows = (n + row_len - 1) // row_len
        root_rays = sum(min(row_len, n - r * row_len) for r in range(root, rows, tiles))
        if "self_send=1" in extra:  # the root tile's records and flags took the exchange, everything else was read in place
            key = "bytes_peer" if "transport=1" in extra else "bytes_rccl"
            assert moved[key] == root_rays * (hit_b + 1) and moved["bytes_in_place"] == (n - root_rays) * hit_b, (extra, moved)
            if key == "bytes_rccl":
                assert "rccl_bound 1" in out, out
        else:  # one device: nothing travels
            assert moved["bytes_rccl"] == 0 and moved["bytes_peer"] == 0 and moved["bytes_in_place"] == n * hit_b, moved


def test_more_tiles_than_rows_and_a_single_ray(group_check):
    """Edge cases of the split: tiles that own no row at all (8 tiles, 5 rows), a frame of one ray, a frame of one short row."""
    exe, d = group_check
    v, f = scenes.sphere(32, 16)
    for tag, (w, h), tiles, row_len in (("few_rows", (40, 5), 8, 40), ("one_ray", (1, 1), 3, 7), ("short_row", (13, 1), 2, 64)):
        rays = scenes.camera_rays(w, h)
        mesh, rp = write_inputs(d, v, f, rays, tag)
        for extra in ([], ["self_send=1", "root=%d" % (tiles - 1)] if tag == "few_rows" else ["self_send=1"]):
            rc, out = run(exe, ["f32", mesh, rp, tiles, row_len] + extra)
            if tag == "few_rows" and extra:  # (the root tile owns no rays: nothing to send to itself, everything read in place)
                assert rc == 0 and "frame_mismatches 0 " in out and "bytes_rccl 0 " in out, out
            else:
                assert rc == 0 and "frame_mismatches 0 " in out and "tile_slot_mismatches 0 " in out, (tag, extra, out)


def test_ranked_group_of_one_sends_to_itself(group_check):
    """nrtGroupCreateRanked (one process per GPU — bench.py's launch model) with a world of one: ncclGetUniqueId, ncclCommInitRank and,
    with self_send, a grouped ncclSend / ncclRecv of the records and flags on that communicator."""
    exe, d = group_check
    v, f = scenes.sphere(64, 32)
    rays = scenes.camera_rays(200, 50)
    mesh, rp = write_inputs(d, v, f, rays, "ranked")
    for extra in (["ranked=1"], ["ranked=1", "self_send=1"]):
        rc, out = run(exe, ["f32", mesh, rp, 1, 200] + extra)
        assert rc == 0 and "frame_mismatches 0 " in out and "tile_slot_mismatches 0 " in out and "tiles 1 local 1 ranks 1 rccl_bound 1" in out, out
    assert "bytes_rccl %d " % (rays.shape[0] * 17) in out, out


def test_group_in_a_process_that_already_holds_torchs_rccl():
    """From Python with torch imported (its bundled librccl is in the process): the library binds to that copy; device tensors in,
    the frame as a device tensor out, equal to one context's TraverseBatchDevice over the whole ray array."""
    import torch

    L = capi.lib()
    vp, u32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
    L.nrtGroupCreate.argtypes = [ctypes.POINTER(vp), u32, ctypes.POINTER(vp)]
    L.nrtGroupDestroy.argtypes = [vp]
    L.nrtGroupDestroy.restype = None
    L.nrtGroupLastError.argtypes = [vp]
    L.nrtGroupLastError.restype = ctypes.c_char_p
    L.nrtGroupSetTunable.argtypes = [vp, ctypes.c_char_p, ctypes.c_longlong]
    L.nrtGroupTileRays.argtypes = [u64, u64, u32, u32]
    L.nrtGroupTileRays.restype = u64
    L.nrtGroupTraverseGather_f32.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(u64), u64, u64, vp, u32, vp, vp]
    L.nrtGroupSynchronize.argtypes = [vp]
    v, f = scenes.sphere(96, 48)
    W, H, N = 256, 96, 4
    rays = scenes.camera_rays(W, H)
    n = rays.shape[0]
    accs = []
    for _ in range(N):
        a = BVHAccel(np.float32)
        assert a.Build(f.shape[0], TriangleMesh(v, f))
        accs.append(a)
    HIT = hit_dtype(np.float32)
    d_all = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
    d_ref = torch.empty(n * HIT.itemsize, dtype=torch.uint8, device="cuda")
    d_refm = torch.empty(n, dtype=torch.uint8, device="cuda")
    accs[0].TraverseBatchDevice(d_all, d_ref, d_refm)
    torch.cuda.synchronize()
    tiles = [torch.from_numpy(np.ascontiguousarray(rays.reshape(H, W)[t::N]).view(np.uint8).reshape(-1)).cuda() for t in range(N)]
    counts = (u64 * N)(*[int(L.nrtGroupTileRays(n, W, t, N)) for t in range(N)])
    assert [int(c) for c in counts] == [W * len(range(t, H, N)) for t in range(N)]
    ctxs = (vp * N)(*[a._h for a in accs])
    g = vp()
    assert L.nrtGroupCreate(ctxs, N, ctypes.byref(g)) == capi.NRT_OK, L.nrtGroupLastError(None)
    try:
        for self_send in (0, 1):
            assert L.nrtGroupSetTunable(g, b"self_send", self_send) == capi.NRT_OK, L.nrtGroupLastError(g)
            frame = torch.full((n * HIT.itemsize,), 0xCD, dtype=torch.uint8, device="cuda")
            fmask = torch.full((n,), 0xCD, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            ptrs = (vp * N)(*[t.data_ptr() for t in tiles])
            st = L.nrtGroupTraverseGather_f32(g, ptrs, counts, n, W, None, 1, frame.data_ptr(), fmask.data_ptr())
            assert st == capi.NRT_OK, L.nrtGroupLastError(g)
            assert L.nrtGroupSynchronize(g) == capi.NRT_OK
            assert torch.equal(frame, d_ref) and torch.equal(fmask, d_refm), "self_send=%d" % self_send
    finally:
        L.nrtGroupDestroy(g)
