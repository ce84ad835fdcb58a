"""SURVEY §8f row 3, second half: the Embree-2 C API (include/embree2/, nanort_amd/lib/libnanort_embree.so) that
replaces the reference's examples/embree-api/nanort-embree.cc.  CPU tests: exports, ABI against the Embree 2.17
headers the reference vendors, source compatibility with the reference demo, error paths, and the fixture pinned on
the C restatement.  GPU tests (-m gpu): tests/cpp/embree_check.cc through the library vs tests/golden/embree_ref.npz
(oracle/gen_golden_embree.py: the unmodified nanosg + the shim's field mapping)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import embree_fixture as ef

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "nanort_amd", "lib")
LIB = os.path.join(LIBDIR, "libnanort_embree.so")
REF = os.environ.get("REFERENCE", "/root/reference")
REF_EMBREE_INC = os.path.join(REF, "examples", "embree-api", "include")
needs_reference = pytest.mark.skipif(not os.path.exists(os.path.join(REF_EMBREE_INC, "embree2", "rtcore.h")),
                                     reason="reference tree not present (GPU box)")


def declared_symbols():
    txt = open(os.path.join(INC, "embree2", "rtcore.h")).read()
    return sorted(set(re.findall(r"RTCORE_API[^;(]*?\b(rtc[A-Za-z0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) == 19 and "rtcIntersect1M" in names and "rtcCommit" in names
    L = ctypes.CDLL(LIB)
    for n in names:
        assert hasattr(L, n), n
    # and nothing undeclared leaks out of the library
    out = subprocess.run(["nm", "-D", "--defined-only", LIB], stdout=subprocess.PIPE, text=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l)
    assert exported == names


def test_library_covers_the_reference_shims_entry_points():
    """Every function the reference shim defines (nanort-embree.cc:454-693, listed here so the check also runs where the
    reference tree is absent) is exported."""
    ref_shim = ["rtcNewDevice", "rtcDeleteScene", "rtcDeleteDevice", "rtcDeviceSetErrorFunction2", "rtcDeviceNewScene",
                "rtcGetBounds", "rtcIntersect", "rtcNewTriangleMesh", "rtcMapBuffer", "rtcUnmapBuffer", "rtcNewInstance2",
                "rtcSetTransform2", "rtcUpdate", "rtcCommit"]
    assert set(ref_shim) <= set(declared_symbols())
    src = os.path.join(REF, "examples", "embree-api", "nanort-embree.cc")
    if os.path.exists(src):
        found = re.findall(r"^RTCORE_API[^(]*?\b(rtc[A-Za-z0-9]+)\s*\(", open(src).read(), flags=re.M)
        assert sorted(found) == sorted(ref_shim)


PROBE = r"""
#include <embree2/rtcore.h>
#include <embree2/rtcore_ray.h>
#include <stddef.h>
#include <stdio.h>
int main() {
  printf("RTCRay %zu %zu", sizeof(RTCRay), alignof(RTCRay));
#define O(f) printf(" %s=%zu", #f, offsetof(RTCRay, f));
  O(org) O(align0) O(dir) O(align1) O(tnear) O(tfar) O(time) O(mask) O(Ng) O(align2) O(u) O(v) O(geomID) O(primID) O(instID)
  printf("\nRTCBounds %zu %zu %zu %zu\n", sizeof(RTCBounds), alignof(RTCBounds), offsetof(RTCBounds, lower_x), offsetof(RTCBounds, upper_x));
  printf("ctx %zu %zu %zu\n", sizeof(RTCIntersectContext), offsetof(RTCIntersectContext, flags), offsetof(RTCIntersectContext, userRayExt));
#define E(x) printf("%s=%d\n", #x, (int)(x));
  E(RTC_NO_ERROR) E(RTC_UNKNOWN_ERROR) E(RTC_INVALID_ARGUMENT) E(RTC_INVALID_OPERATION) E(RTC_OUT_OF_MEMORY) E(RTC_UNSUPPORTED_CPU) E(RTC_CANCELLED)
  E(RTC_SCENE_STATIC) E(RTC_SCENE_DYNAMIC) E(RTC_SCENE_COMPACT) E(RTC_SCENE_COHERENT) E(RTC_SCENE_INCOHERENT) E(RTC_SCENE_HIGH_QUALITY) E(RTC_SCENE_ROBUST)
  E(RTC_INTERSECT1) E(RTC_INTERSECT4) E(RTC_INTERSECT8) E(RTC_INTERSECT16) E(RTC_INTERPOLATE) E(RTC_INTERSECT_STREAM)
  E(RTC_INTERSECT_COHERENT) E(RTC_INTERSECT_INCOHERENT) E(RTC_INDEX_BUFFER) E(RTC_VERTEX_BUFFER)
  E(RTC_GEOMETRY_STATIC) E(RTC_GEOMETRY_DEFORMABLE) E(RTC_GEOMETRY_DYNAMIC)
  E(RTC_MATRIX_ROW_MAJOR) E(RTC_MATRIX_COLUMN_MAJOR) E(RTC_MATRIX_COLUMN_MAJOR_ALIGNED16)
  printf("invalid=%u\n", RTC_INVALID_GEOMETRY_ID);
  /* the signatures: taking the address with the expected type fails to compile if a header disagrees */
  void (*f1)(RTCScene, RTCRay &) = &rtcIntersect;
  void (*f2)(RTCScene, const RTCIntersectContext *, RTCRay *, const size_t, const size_t) = &rtcIntersect1M;
  unsigned (*f3)(RTCScene, RTCGeometryFlags, size_t, size_t, size_t) = &rtcNewTriangleMesh;
  void *(*f4)(RTCScene, unsigned, RTCBufferType) = &rtcMapBuffer;
  void (*f5)(RTCScene, RTCBounds &) = &rtcGetBounds;
  void (*f6)(RTCScene, const RTCIntersectContext *, RTCRay **, const size_t) = &rtcIntersect1Mp;
  void (*f7)(RTCScene, const RTCIntersectContext *, RTCRay *, const size_t, const size_t) = &rtcOccluded1M;
  return (f1 && f2 && f3 && f4 && f5 && f6 && f7) ? 0 : 1;
}
"""


@needs_reference
def test_abi_equals_the_vendored_embree_headers(tmp_path):
    """Layouts, enumerator values and function types of include/embree2 == those of the Embree 2.17 headers vendored by the
    reference: a program compiled against either links against libnanort_embree.so."""
    src = tmp_path / "probe.cc"
    src.write_text(PROBE)
    outs = []
    for inc in (INC, REF_EMBREE_INC):
        exe = tmp_path / ("probe_" + str(len(outs)))
        r = subprocess.run(["g++", "-std=c++11", "-w", "-I", inc, str(src), "-o", str(exe), "-L", LIBDIR, "-lnanort_embree",
                            "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
        outs.append(subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True, check=True).stdout)
    assert outs[0] == outs[1]
    assert outs[0].startswith("RTCRay 96 16 org=0")


@needs_reference
def test_reference_demo_and_check_program_compile_against_either_header_set():
    """examples/embree-api/main.cc (the reference's demo, unchanged) type-checks against include/embree2; the test driver
    type-checks against the vendored Embree headers."""
    demo = os.path.join(REF, "examples", "embree-api", "main.cc")
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-w", "-I", INC, "-I", os.path.join(REF, "examples", "nanosg"),
                        "-I", os.path.join(REF, "examples", "common"), "-I", REF, demo], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    chk = os.path.join(ROOT, "tests", "cpp", "embree_check.cc")
    for inc in (INC, REF_EMBREE_INC):
        r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-I", inc, chk], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout


@needs_reference
def test_reference_shim_is_unbuildable_as_documented(tmp_path):
    """DESIGN.md says the reference's own shim cannot serve as a compiled oracle; keep that claim honest."""
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-w", "-I", REF, "-I", os.path.join(REF, "examples", "nanosg"),
                        "-I", REF_EMBREE_INC, os.path.join(REF, "examples", "embree-api", "nanort-embree.cc")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode != 0 and "Traverse" in r.stdout


def test_fixture_pinned_on_the_restatement(oracle, golden_dir):
    """The C restatement of nanosg's traversal (oracle/nanosg_oracle.c) + the shim's field mapping reproduce the fixture
    the unmodified nanosg generated, in every field."""
    from nanort_amd.wire import RAY_F32
    from oracle import bindings as ob

    g = np.load(os.path.join(golden_dir, "embree_ref.npz"))
    O = ob.SceneOracle(oracle)
    for v, f in ef.meshes():
        O.add_node(v, f, np.eye(4, dtype=np.float32))
    assert O.commit()
    r = ef.rays()
    nr = np.zeros((r.shape[0],), dtype=RAY_F32)
    nr["org"], nr["dir"], nr["min_t"], nr["max_t"] = r[:, 0:3], r[:, 3:6], r[:, 6], r[:, 7]
    h, m = O.traverse(nr)
    hit = m != 0
    assert np.array_equal(m, g["hit"])
    assert np.where(hit, h["t"], r[:, 7]).astype(np.float32).tobytes() == g["tfar"].tobytes()
    assert h["u"][hit].tobytes() == g["u"].tobytes() and h["v"][hit].tobytes() == g["v"].tobytes()
    assert np.array_equal(h["node_id"][hit], g["geomID"][hit]) and np.array_equal(h["prim_id"][hit], g["primID"][hit])
    assert (g["geomID"][~hit] == ef.INVALID).all() and (g["primID"][~hit] == ef.INVALID).all()


def _api():
    L = ctypes.CDLL(LIB)
    vp, u32, sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_size_t
    L.rtcNewDevice.restype, L.rtcNewDevice.argtypes = vp, [ctypes.c_char_p]
    L.rtcDeleteDevice.argtypes = [vp]
    L.rtcDeviceGetError.restype, L.rtcDeviceGetError.argtypes = ctypes.c_int, [vp]
    L.rtcDeviceNewScene.restype, L.rtcDeviceNewScene.argtypes = vp, [vp, ctypes.c_int, ctypes.c_int]
    L.rtcDeleteScene.argtypes = [vp]
    L.rtcCommit.argtypes = [vp]
    L.rtcGetBounds.argtypes = [vp, vp]
    L.rtcNewTriangleMesh.restype, L.rtcNewTriangleMesh.argtypes = u32, [vp, ctypes.c_int, sz, sz, sz]
    L.rtcMapBuffer.restype, L.rtcMapBuffer.argtypes = vp, [vp, u32, ctypes.c_int]
    L.rtcUnmapBuffer.argtypes = [vp, u32, ctypes.c_int]
    L.rtcNewInstance2.restype, L.rtcNewInstance2.argtypes = u32, [vp, vp, sz]
    L.rtcIntersect.argtypes = [vp, vp]
    L.rtcIntersect1M.argtypes = [vp, vp, vp, sz, sz]
    L.rtcDeviceSetErrorFunction2.argtypes = [vp, vp, vp]
    return L


RTC_INVALID_ARGUMENT, RTC_INVALID_OPERATION, RTC_UNKNOWN_ERROR = 2, 3, 1
RTC_INDEX_BUFFER, RTC_VERTEX_BUFFER = 0x01000000, 0x02000000


def test_argument_errors_ids_and_uncommitted_scene():
    """The reference shim's argument checks (nanort-embree.cc:567-588, 601-631), its 1-based ids, nanosg's invalid box for
    an uncommitted scene, and a query before rtcCommit: a recorded error and a miss — no GPU involved."""
    L = _api()
    messages = []
    CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p)
    cb = CB(lambda up, code, s: messages.append((code, s.decode())))
    dev = L.rtcNewDevice(None)
    L.rtcDeviceSetErrorFunction2(dev, ctypes.cast(cb, ctypes.c_void_p), None)
    sc = L.rtcDeviceNewScene(dev, 0, 1)
    assert L.rtcDeviceGetError(dev) == 0
    assert L.rtcNewTriangleMesh(sc, 0, 4, 8, 2) == 0 and L.rtcDeviceGetError(dev) == RTC_INVALID_ARGUMENT
    assert "Motion blur" in messages[-1][1]
    assert L.rtcNewTriangleMesh(sc, 0, 0, 8, 1) == 0 and L.rtcDeviceGetError(dev) == RTC_INVALID_ARGUMENT
    assert L.rtcNewTriangleMesh(sc, 0, 4, 0, 1) == 0 and L.rtcDeviceGetError(dev) == RTC_INVALID_ARGUMENT
    assert L.rtcDeviceGetError(dev) == 0  # reading clears
    assert L.rtcNewTriangleMesh(sc, 0, 1, 3, 1) == 1 and L.rtcNewTriangleMesh(sc, 0, 2, 4, 1) == 2
    v = L.rtcMapBuffer(sc, 2, RTC_VERTEX_BUFFER)
    f = L.rtcMapBuffer(sc, 2, RTC_INDEX_BUFFER)
    assert v and f and v != f
    # the vertex buffer really has Embree's 16-byte stride: 4 vertices x 4 floats are writable and zero-initialised
    assert np.ctypeslib.as_array(ctypes.cast(v, ctypes.POINTER(ctypes.c_float)), (16,)).tolist() == [0.0] * 16
    assert not L.rtcMapBuffer(sc, 7, RTC_VERTEX_BUFFER) and L.rtcDeviceGetError(dev) == RTC_INVALID_ARGUMENT
    assert not L.rtcMapBuffer(sc, 1, 0x03000000) and L.rtcDeviceGetError(dev) == RTC_INVALID_ARGUMENT
    assert L.rtcNewInstance2(sc, sc, 1) == 0 and L.rtcDeviceGetError(dev) == RTC_INVALID_OPERATION
    b = np.zeros(8, dtype=np.float32)
    L.rtcGetBounds(sc, b.ctypes.data)
    fmax = np.finfo(np.float32).max
    assert b[0] == fmax and b[4] == -fmax  # nanosg.h:745-753
    ray = np.zeros(1, dtype=ef.RTC_RAY)
    ray["dir"] = (0, 0, 1)
    ray["tfar"] = 5.0
    ray["geomID"] = 123
    L.rtcIntersect(sc, ray.ctypes.data)
    assert ray["geomID"][0] == ef.INVALID and ray["primID"][0] == ef.INVALID and ray["tfar"][0] == 5.0
    assert L.rtcDeviceGetError(dev) == RTC_INVALID_OPERATION
    L.rtcDeleteScene(sc)
    L.rtcDeleteDevice(dev)


def test_commit_fails_loudly_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = _api()
    dev = L.rtcNewDevice(None)
    sc = L.rtcDeviceNewScene(dev, 0, 1)
    assert L.rtcNewTriangleMesh(sc, 0, 1, 3, 1) == 1
    L.rtcCommit(sc)
    assert L.rtcDeviceGetError(dev) == RTC_UNKNOWN_ERROR
    ray = np.zeros(4, dtype=ef.RTC_RAY)
    L.rtcIntersect1M(sc, None, ray.ctypes.data, 4, 96)
    assert (ray["geomID"] == ef.INVALID).all()  # no CPU stand-in: nothing is traced
    L.rtcDeleteDevice(dev)  # deletes its scenes too


# ---------------------------------------------------------------- GPU ----------------------------------------------------


@pytest.fixture(scope="module")
def check_exe(tmp_path_factory):
    d = tmp_path_factory.mktemp("embree")
    exe = d / "embree_check"
    r = subprocess.run(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-I", INC, os.path.join(ROOT, "tests", "cpp", "embree_check.cc"),
                        "-o", str(exe), "-L", LIBDIR, "-lnanort_embree", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    (d / "scene.bin").write_bytes(ef.scene_bytes())
    return d, exe


def run_check(check_exe, rays, mode):
    d, exe = check_exe
    (d / "rays.bin").write_bytes(ef.rays_bytes(rays))
    r = subprocess.run([str(exe), str(d / "scene.bin"), str(d / "rays.bin"), str(d / "out.bin"), mode],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and "RTC error" not in r.stdout, r.stdout
    return ef.parse_out((d / "out.bin").read_bytes(), 5, rays.shape[0])


def assert_matches_golden(out, rays, g, sel, oracle):
    """`out`: RTCRay records for rays[sel] of the fixture.  Hit flag, tfar and geomID are exact.  The mesh trees are built
    by the GPU builder, so primID/u/v may name the other triangle of an EXACT tie (a shared edge; the reference keeps
    whichever it tests last, nanort.h:1133) — each such ray is verified with the restatement restricted to either
    triangle: same local t, and the reported u, v are that triangle's."""
    from helpers import trace_options
    from nanort_amd.wire import RAY_F32

    hit_all = g["hit"] != 0
    hit = hit_all[sel]
    pos = np.cumsum(hit_all) - 1  # index into the compacted u / v arrays
    assert np.array_equal(out["geomID"] != ef.INVALID, hit)
    assert out["tfar"].tobytes() == g["tfar"][sel].tobytes()
    assert np.array_equal(out["geomID"], g["geomID"][sel])
    assert (out["instID"] == ef.INVALID).all()
    same = out["primID"] == g["primID"][sel]
    assert same[~hit].all() and same.mean() > 0.99
    k = same & hit
    assert np.array_equal(out["u"][k], g["u"][pos[sel][k]]) and np.array_equal(out["v"][k], g["v"][pos[sel][k]])
    ms, trees = ef.meshes(), {}
    for i in np.nonzero(~same)[0]:
        gid = int(out["geomID"][i])
        v, f = ms[gid]
        if gid not in trees:
            trees[gid] = oracle.build(v, f)[:2]
        lr = np.zeros(1, dtype=RAY_F32)  # the local ray of an identity node: nanosg.h:806-817
        lr["org"], lr["dir"], lr["min_t"], lr["max_t"] = out["org"][i], out["dir"][i], 0.0, np.finfo(np.float32).max
        rec = []
        for p_ in (int(out["primID"][i]), int(g["primID"][sel][i])):
            h1, m1 = oracle.traverse(trees[gid][0], trees[gid][1], v, f, lr, trace_options(range_=(p_, p_ + 1)))
            assert m1[0] == 1
            rec.append(h1[0])
        assert rec[0]["t"] == rec[1]["t"], "ray %d: prims %d / %d are not an exact tie" % (i, out["primID"][i], g["primID"][sel][i])
        assert rec[0]["u"] == out["u"][i] and rec[0]["v"] == out["v"][i]
    # untouched: inputs, time, mask, Ng, the padding — and u, v on a miss (nanort-embree.cc:549-553)
    a5 = np.frombuffer(b"\xa5" * 4, dtype="<f4")[0].tobytes()
    for name in ("align0", "align1", "time", "align2"):
        assert out[name].tobytes() == a5 * out.shape[0], name
    assert out["Ng"].tobytes() == a5 * 3 * out.shape[0]
    assert (out["mask"] == 0xA5A5A5A5).all()
    assert out["u"][~hit].tobytes() == a5 * int((~hit).sum())
    assert np.array_equal(out["org"], rays[sel][:, 0:3]) and np.array_equal(out["dir"], rays[sel][:, 3:6])
    assert np.array_equal(out["tnear"], rays[sel][:, 6])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["stream", "streamp", "recommit"])
def test_ray_streams_match_the_reference_fixture(check_exe, golden_dir, oracle, mode):
    g = np.load(os.path.join(golden_dir, "embree_ref.npz"))
    rays = ef.rays()
    bounds, ids, out = run_check(check_exe, rays, mode)
    assert bounds.tobytes() == g["bounds"].tobytes()
    assert np.array_equal(ids, g["ids"])
    assert_matches_golden(out, rays, g, np.arange(rays.shape[0]), oracle)


@pytest.mark.gpu
def test_single_ray_calls_match_the_reference_fixture(check_exe, golden_dir, oracle):
    """rtcIntersect, one GPU round trip per ray (the reference demo's loop): a spread of 400 fixture rays."""
    g = np.load(os.path.join(golden_dir, "embree_ref.npz"))
    rays = ef.rays()
    sel = np.arange(0, rays.shape[0], 51)
    _, _, out = run_check(check_exe, np.ascontiguousarray(rays[sel]), "single")
    assert_matches_golden(out, rays, g, sel, oracle)


@pytest.mark.gpu
def test_occlusion_queries(check_exe, golden_dir):
    g = np.load(os.path.join(golden_dir, "embree_ref.npz"))
    rays = ef.rays()
    _, _, out = run_check(check_exe, rays, "occluded")
    hit = g["hit"] != 0
    assert (out["geomID"][hit] == 0).all() and (out["geomID"][~hit] == 0xA5A5A5A5).all()
    assert np.array_equal(out["tfar"], rays[:, 7]) and (out["primID"] == 0xA5A5A5A5).all()
