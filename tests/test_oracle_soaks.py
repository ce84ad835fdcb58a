"""Short runs of the CPU soaks that pin the checkers on the unmodified reference (tests/checks/fuzz_oracle_vs_reference.py,
fuzz_prim_oracles_vs_reference.py): random hostile inputs, every record bit for bit.  Only where the reference-built
libraries exist (the build container); the long runs of round 1 covered 109 M + 15.7 M rays."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
needs_ref = pytest.mark.skipif(
    not all(os.path.exists(os.path.join(REF, n)) for n in ("libnanort_ref.so", "libsphere_ref.so", "libcylinder_ref.so", "libnanosg_ref.so")),
    reason="oracle/_ref not built (needs the reference tree)")


@needs_ref
@pytest.mark.parametrize("script,token", [("fuzz_oracle_vs_reference.py", "oracle == reference"),
                                          ("fuzz_prim_oracles_vs_reference.py", "restatements == references")])
def test_restatements_equal_the_reference_on_random_hostile_inputs(script, token):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "checks", script), "8", "3"], cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and token in r.stdout, r.stdout[-2000:]


def test_header_host_path_equals_the_restatement_on_random_hostile_inputs():
    """include/nanort.h without the GPU backend (its own builder + per-ray Traverse) against the restatement walking the
    header's own tree: tests/checks/fuzz_header_host_path.py for a few seconds (round 1: 53 M rays, no difference)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "checks", "fuzz_header_host_path.py"), "8", "5"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "header host path == restatement" in r.stdout, r.stdout[-2000:]


def test_saved_cross_tree_cases_stay_documented():
    """The two residual cases of round 1's cross-tree fuzz (tests/golden/fuzz_case_r01_crosstree_*.npz):
    on these grid-aligned meshes the REFERENCE's own answer depends on the tree beyond exact ties (a triangle's t one ulp
    below its leaf box's entry distance; rays lying in a triangle's plane).  Same-tree parity is exact (GPU:
    tests/test_gpu_fuzz_cases.py replays them); across trees the restatement agrees with itself on >= 99.8 % of the rays and
    NOT on all of them — if that ever becomes 100 % the documented caveat can go."""
    import glob
    import os

    from oracle.bindings import Oracle

    orc = Oracle()
    here = os.path.dirname(os.path.abspath(__file__))
    differing = 0
    for case in sorted(glob.glob(os.path.join(here, "golden", "fuzz_case_r01_crosstree_*.npz"))):
        d = np.load(case)
        v, f, rays, opts, nodes, idx = d["v"], d["f"], d["rays"], d["opts"], d["nodes"], d["idx"]
        h, m = orc.traverse(nodes, idx, v, f, rays, opts)          # the tree the GPU builder produced in that round
        n2, i2, _ = orc.build(v, f)                                  # the reference builder's tree
        h2, m2 = orc.traverse(n2, i2, v, f, rays, opts)
        assert (m == m2).mean() > 0.998
        both = (m == 1) & (m2 == 1) & np.isfinite(h["t"]) & np.isfinite(h2["t"])
        close = np.abs(h["t"][both] - h2["t"][both]) <= 1e-5 * np.maximum(1.0, np.abs(h2["t"][both]))
        assert close.mean() > 0.998
        differing += int((m != m2).sum()) + int((~close).sum())
    assert differing > 0, "the cross-tree difference these fixtures document is gone"
