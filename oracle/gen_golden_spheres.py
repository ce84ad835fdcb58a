#!/usr/bin/env python3
"""Writes tests/golden/spheres_ref.npz from the UNMODIFIED reference (oracle/_ref/libsphere_ref.so:
examples/particle_primitive/main.cc over nanort.h): the reference-built tree of the test scene and the
reference's hit records for the test rays (tests/sphere_fixture.py), full range and a restricted prim_ids_range.
Run in the build container (needs /root/reference): python oracle/gen_golden_spheres.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle.bindings import SphereReference  # noqa: E402
import sphere_fixture  # noqa: E402


def main():
    R = SphereReference()
    c, r = sphere_fixture.scene()
    rc, rr = R.generate(sphere_fixture.N_SPHERES)
    assert c.tobytes() == rc.tobytes() and r.tobytes() == rr.tobytes(), "scene generator differs from the example's"
    nodes, idx, st = R.build(c, r)
    rays = sphere_fixture.rays()
    h, m = R.traverse(rays)
    h2, m2 = R.traverse(rays, (1000, 3000))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "spheres_ref.npz"), nodes=nodes, indices=idx, hits=h, mask=m,
                        hits_range=h2, mask_range=m2, stats=np.array([st["max_tree_depth"], st["num_leaf_nodes"], st["num_branch_nodes"]]))
    print("spheres_ref.npz:", st, "rays", rays.shape[0], "hits", int(m.sum()), "hits in range", int(m2.sum()))


if __name__ == "__main__":
    main()
