"""The C-ABI library loads and exports every symbol include/nanort_hip.h declares (no compute: CPU box)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from nanort_amd import capi, wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nanort_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    return sorted(set(re.findall(r"NRT_API\s+[\w\s\*]+?\b(nrt\w+)\s*\(", src)))


def test_header_symbols_match_binding_table():
    assert declared_symbols() == sorted(capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    for name in declared_symbols():
        assert hasattr(L, name), "libnanort_hip.so does not export " + name
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    exported = set(re.findall(r" T (nrt\w+)", out))
    assert exported == set(declared_symbols()), "exported C symbols differ from the header"


def test_profiling_entry_points_live_in_the_profiling_library_only():
    """libnanort_hip_prof.so = the same sources with -DNRT_PROF: the product ABI plus the three profiling calls of
    include/nanort_hip_prof.h; the product library exports none of them."""
    prof = os.path.join(os.path.dirname(capi.LIB_PATH), "libnanort_hip_prof.so")
    assert os.path.exists(prof), "make -C nanort_amd/csrc builds it"

    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], stdout=subprocess.PIPE, text=True).stdout
        return set(re.findall(r" T (nrt\w+)", out))

    base = exported(os.path.join(os.path.dirname(capi.LIB_PATH), "libnanort_hip.so"))
    assert not any("Debug" in n for n in base)
    hdr = open(os.path.join(ROOT, "include", "nanort_hip_prof.h")).read()
    extra = set(re.findall(r"NRT_API\s+[\w\s\*]+?\b(nrt\w+)\s*\(", hdr))
    assert extra == {"nrtDebugCounters", "nrtDebugWaveClocks", "nrtSceneDebugCounters"}
    assert exported(prof) == base | extra


def test_library_has_gfx950_code_object():
    data = open(capi.LIB_PATH, "rb").read()
    assert b"gfx950" in data and b"k_traverse" in data and b"k_subtree" in data


def test_header_compiles_as_c_and_cxx(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "nanort_hip.h"\nint main(void){return sizeof(nrt_ray_f32)==36 && sizeof(nrt_node_f64)==64 ? 0 : 1;}\n')
    for cc, std in (("gcc", "-std=c99"), ("g++", "-std=c++11")):
        exe = tmp_path / ("t_" + cc)
        subprocess.check_call([cc, std, "-x", "c" if cc == "gcc" else "c++", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                               str(src), "-o", str(exe)])
        assert subprocess.call([str(exe)]) == 0


def test_wire_dtypes_match_reference_layouts():
    assert wire.RAY_F32.itemsize == 36 and wire.RAY_F64.itemsize == 72
    assert wire.NODE_F32.itemsize == 40 and wire.NODE_F64.itemsize == 64
    assert wire.HIT_F32.itemsize == 16 and wire.HIT_F64.itemsize == 32
    assert wire.NODE_F32.fields["flag"][1] == 24 and wire.NODE_F32.fields["data"][1] == 32
    assert wire.NODE_F64.fields["flag"][1] == 48 and wire.NODE_F64.fields["data"][1] == 56
    o = wire.default_trace_options()
    assert tuple(o["prim_ids_range"]) == (0, 0x7FFFFFFF) and o["skip_prim_id"] == 0xFFFFFFFF
    b = wire.default_build_options()
    assert (b["min_leaf_primitives"], b["max_tree_depth"], b["bin_size"]) == (4, 256, 64)


def test_version_and_null_error_string():
    L = capi.lib()
    assert b"gfx950" in L.nrtVersion()
    assert L.nrtLastError(None) is not None


def test_create_fails_loudly_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = capi.lib()
    h = ctypes.c_void_p()
    st = L.nrtCreate(0, ctypes.byref(h))
    assert st == capi.NRT_ERR_DEVICE and not h.value
    assert b"no HIP device" in L.nrtLastError(None)
    from nanort_amd import BVHAccel, NrtError

    with pytest.raises(NrtError):
        BVHAccel(np.float32)


def test_product_never_imports_the_oracle():
    """nanort_amd/, include/, examples/ and tools/ must not reference oracle/ (no CPU path behind the product; scripts that
    use the oracle as a checker live under tests/)."""
    bad = []
    for base in ("nanort_amd", "include", "examples", "tools"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".h", ".hip", ".c", ".cc", ".cpp", ".sh")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"(from|import)\s+oracle|oracle/|liboracle|libnanort_ref", txt):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad
