"""CPU soak of the two-levels-per-step walk (oracle/wide4_model_body.inc, the model of the traversal kernel's
NRT_STEP_NODE4) against the restated reference loop on the hostile generator of hostile.py: same hit records,
same SEQUENCE of visited leaves, same numbers of leaf and triangle tests.
Usage: python tests/checks/fuzz_wide4_model.py [seconds] [seed]     (needs no GPU)"""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/checks')
from hostile import hostile_case
from oracle.bindings import Oracle


def one_round(rng, orc, stats):
    v, f, rays, opts, nodes, idx = hostile_case(rng, orc)
    oh, om, oc = orc.traverse(nodes, idx, v, f, rays, opts, count=True)
    cap = int(oc[1]) + 16  # leaves the reference loop tests, all rays
    wh, wm, wc, t_ref, t_w4 = orc.traverse_wide4_model(nodes, idx, v, f, rays, opts, trail_cap=cap)
    same = (np.array_equal(om, wm) and all(oh[k].tobytes() == wh[k].tobytes() for k in ("t", "u", "v", "prim_id"))
            and np.array_equal(t_ref, t_w4) and int(wc[1]) == int(oc[1]) and int(wc[2]) == int(oc[2]))
    stats["rays"] += rays.shape[0]; stats["leaves"] += int(oc[1]); stats["steps"] += int(wc[0]); stats["nodes_ref"] += int(oc[0])
    stats["max_stack"] = max(stats["max_stack"], int(wc[3]))
    return same, (v, f, rays, opts, nodes, idx)


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    orc = Oracle()
    stats = {"rays": 0, "leaves": 0, "steps": 0, "nodes_ref": 0, "max_stack": 0}
    t_end = time.time() + budget
    rounds = 0
    while time.time() < t_end:
        ok, case = one_round(rng, orc, stats)
        if not ok:
            np.savez("/tmp/wide4_model_failure.npz", **dict(zip(("v", "f", "rays", "opts", "nodes", "idx"), case)))
            print("MISMATCH in round %d (case saved to /tmp/wide4_model_failure.npz)" % rounds, stats)
            sys.exit(1)
        rounds += 1
    print("wide4 model ok: %d rounds, %s" % (rounds, stats))
