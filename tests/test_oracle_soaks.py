"""Short runs of the CPU soaks that pin the checkers on the unmodified reference (tests/checks/fuzz_oracle_vs_reference.py,
fuzz_prim_oracles_vs_reference.py): random hostile inputs, every record bit for bit.  Only where the reference-built
libraries exist (the build container); the long runs of round 1 covered 109 M + 15.7 M rays."""
import os
import subprocess
import sys

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
