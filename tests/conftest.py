import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The library's default walk is the reference-order walk (tunable order4 = 0): every field of every record bit-identical to
# the restated reference on the same node array — that is what the GPU suite asserts wherever it compares with the oracle on
# the tree the GPU built, at every BASELINE config size.  The opt-in distance-ordered walk (order4 = 1) is covered by
# tests/test_gpu_order4.py and by the `order4` arms of the config-size tests (tie-aware: every differing ray re-verified).
# NRT_<TUNABLE> environment overrides are honoured by the product library only under NRT_ALLOW_ENV=1; no test relies on one
# being set for the whole session.
for _k in [k for k in os.environ if k.startswith("NRT_") and k not in ("NRT_USE_PROF_LIB", "NRT_BENCH_TEST_SHARED_GPU")]:
    del os.environ[_k]  # a stray variable in the caller's shell must not change what the suite tests


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
