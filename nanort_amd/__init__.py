"""nanort_amd — MI355X (gfx950) backend for the hot path of lighttransport/nanort:
BVHAccel::Build + BVHAccel::Traverse for triangle meshes.

  include/nanort_hip.h   the C ABI (the drop-in boundary)
  include/nanort.h       API-compatible header-only C++ host side
  nanort_amd/csrc/       hand-written HIP kernels + the C ABI implementation
  nanort_amd/accel.py    Python mirror of the reference interface (ctypes)
  nanort_amd/scenes.py   the synthetic workloads of SURVEY.md §8(d)
"""
from . import wire  # noqa: F401
from .accel import BVHAccel, CylinderGeometry, Scene, SphereGeometry, TriangleMesh  # noqa: F401
from .capi import NrtError  # noqa: F401
