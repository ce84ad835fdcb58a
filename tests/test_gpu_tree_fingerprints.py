"""The GPU builder's trees, bit for bit: fingerprints (md5 of node array + index permutation) of 26 builds over meshes,
precisions and build options, recorded at the end of round 1 (tests/golden/tree_fingerprints.txt == tools/tree_hash_r01.txt).
Every builder change since — fewer launches per level, host read-backs off the critical path, the subtree kernel's
records out of LDS, one-chunk nodes split inside k_bin, the hand-off size decoupled from the bin rule, layout from
parent links — had to reproduce every line; this test keeps it that way."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_builder_reproduces_the_recorded_trees():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tree_hash.py")], cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = [l for l in r.stdout.splitlines() if l.strip()]
    want = [l for l in open(os.path.join(ROOT, "tests", "golden", "tree_fingerprints.txt")).read().splitlines() if l.strip()]
    assert len(want) == 26
    assert got == want, "\n".join("%s\n  != %s" % (g, w) for g, w in zip(got, want) if g != w)
