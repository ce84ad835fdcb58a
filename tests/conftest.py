import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


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
