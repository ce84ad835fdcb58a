"""The fold rule of the traversal kernel's drain-time work splitting, checked on the CPU against the restated reference
loop (oracle/split_model_body.inc; long runs: tests/checks/fuzz_split_model.py)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "checks"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_split_and_fold_equals_the_sequential_loop_on_hostile_inputs():
    from fuzz_split_model import one_round
    from oracle.bindings import Oracle

    rng = np.random.default_rng(7)
    orc = Oracle()
    stats = {"rays": 0, "split_rays": 0, "segments": 0, "flagged": 0, "would_differ_without_flag": 0}
    t_end = time.time() + 6.0
    rounds = 0
    while time.time() < t_end or rounds < 12:
        ok, _ = one_round(rng, orc, stats)
        assert ok, "the folded result differs from the sequential loop (round %d, %s)" % (rounds, stats)
        rounds += 1
    assert stats["split_rays"] > 1000 and stats["segments"] > stats["split_rays"]
    # the consistency flag is what makes the fold exact on this kind of geometry: it must actually fire here
    assert stats["flagged"] > 0


def test_c1_camera_wave_split_at_every_opportunity():
    from nanort_amd import scenes
    from oracle.bindings import Oracle

    orc = Oracle()
    v, f = scenes.load_c1_mesh()
    rays = scenes.camera_rays(128, 128)
    nodes, idx, _ = orc.build(v, f)
    oh, om = orc.traverse(nodes, idx, v, f, rays)
    sh, sm, fl, sp = orc.traverse_split_model(nodes, idx, v, f, rays, split_permille=1000, seed=3)
    assert np.array_equal(om, sm) and oh.tobytes() == sh.tobytes()
    assert int(sp.max()) >= 5 and int(fl.sum()) < rays.shape[0] // 50  # a real scene: (almost) no ray needs the re-run
