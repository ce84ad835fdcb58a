#!/usr/bin/env python3
"""Writes tests/golden/cylinders_ref.npz from the UNMODIFIED reference (oracle/_ref/libcylinder_ref.so:
examples/cylinder_primitive/main.cc over nanort.h): the reference-built tree of the test scene and the reference's
hit records for the test rays (tests/sphere_fixture.py rays; scene = the example's generator at N_CYLINDERS), with
caps, without caps, and with a restricted prim_ids_range.
Run in the build container (needs /root/reference): python oracle/gen_golden_cylinders.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle.bindings import CylinderReference  # noqa: E402
import sphere_fixture  # noqa: E402
from nanort_amd import scenes  # noqa: E402


def main():
    R = CylinderReference()
    v, r = scenes.random_cylinders(sphere_fixture.N_CYLINDERS)
    rv, rr = R.generate(sphere_fixture.N_CYLINDERS)
    assert v.tobytes() == rv.tobytes() and r.tobytes() == rr.tobytes(), "scene generator differs from the example's"
    nodes, idx, st = R.build(v, r)
    rays = sphere_fixture.rays()
    h, m = R.traverse(rays)
    h0, m0 = R.traverse(rays, test_cap=False)
    h2, m2 = R.traverse(rays, (500, 2500))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cylinders_ref.npz"), nodes=nodes, indices=idx, hits=h, mask=m,
                        hits_nocap=h0, mask_nocap=m0, hits_range=h2, mask_range=m2)
    print("cylinders_ref.npz:", st, "rays", rays.shape[0], "hits", int(m.sum()), "no caps", int(m0.sum()), "in range", int(m2.sum()),
          "cap hits", int(((h["v"] == 0) | (h["v"] == 1))[m == 1].sum()))


if __name__ == "__main__":
    main()
