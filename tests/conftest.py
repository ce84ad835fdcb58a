import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The library's default two-level walk enters a record's four slots by entry distance (tunable order4 = 1): hit flags and t
# are the reference's bits on the same node array, but among primitives at EXACTLY the same t it may name another one than the
# reference's leaf order would.  The bulk of this suite states the stronger property — every field bit-identical to the
# restated reference on the same node array — which is the contract of the reference-order walk (order4 = 0): contexts made by
# these tests therefore start with order4 = 0 (NRT_<TUNABLE> overrides a default at nrtCreate) unless a test chooses otherwise.
# The default walk is covered by tests/test_gpu_order4.py (tie-aware, every differing ray re-verified), by the multi-batch and
# multi-context tests' order4 arms, by bench.py's in-run parity and by the fuzz soak's --default-walk mode.
os.environ.setdefault("NRT_ORDER4", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    from nanort_amd import capi

    need = [
        capi.LIB_PATH,
        os.path.join(ROOT, "nanort_amd", "lib", "libnrt_scenes.so"),
        os.path.join(ROOT, "nanort_amd", "lib", "libnanort_embree.so"),
        os.path.join(ROOT, "oracle", "liboracle.so"),
    ]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__

        __graft_entry__.build()


_ensure_built()


@pytest.fixture(scope="session")
def oracle():
    from oracle.bindings import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def c1_mesh():
    from nanort_amd import scenes

    return scenes.load_c1_mesh()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
