"""Shared test helpers (CPU side): tie-aware hit comparison, option builders."""
import numpy as np

from nanort_amd.wire import default_trace_options


def trace_options(range_=None, skip=None, cull=None):
    o = default_trace_options()
    if range_ is not None:
        o["prim_ids_range"] = range_
    if skip is not None:
        o["skip_prim_id"] = skip
    if cull is not None:
        o["cull_back_face"] = 1 if cull else 0
    return o


def _fields_equal(a, b):
    """Bitwise equality of the four meaningful fields (fp64 records carry 4 padding bytes)."""
    return all(np.ascontiguousarray(a[k]).tobytes() == np.ascontiguousarray(b[k]).tobytes() for k in ("t", "u", "v", "prim_id"))


def assert_hits_identical(a_hits, a_mask, b_hits, b_mask):
    """Bit-for-bit equality of hit records (same tree => same tie breaks)."""
    assert np.array_equal(a_mask, b_mask), "hit masks differ at %s" % np.nonzero(a_mask != b_mask)[0][:8]
    if not _fields_equal(a_hits, b_hits):
        bad = np.nonzero(
            (a_hits["t"] != b_hits["t"]) | (a_hits["prim_id"] != b_hits["prim_id"])
            | (a_hits["u"] != b_hits["u"]) | (a_hits["v"] != b_hits["v"])
        )[0]
        raise AssertionError("hit records differ at %d rays, first %s:\n%s\n%s" % (
            bad.size, bad[:4], a_hits[bad[:4]], b_hits[bad[:4]]))


def assert_hits_match(ref_hits, ref_mask, hits, mask, oracle, onodes, oindices, verts, faces, rays,
                      base_opts=None, max_ties=None):
    """Parity across DIFFERENT trees: hit flags and t bit-equal
    everywhere; u, v, prim_id bit-equal except at true ties — two primitives at
    exactly the same t along the ray (shared edges/vertices), where the
    reference itself keeps whichever it tested last (nanort.h:1133 accepts
    equality).  Every such ray is verified: the oracle restricted to the
    reported primitive must reproduce the reported record bit-for-bit.
    Returns the number of ties."""
    assert np.array_equal(ref_mask, mask), "hit masks differ at %s" % np.nonzero(ref_mask != mask)[0][:8]
    rt, ht = np.ascontiguousarray(ref_hits["t"]), np.ascontiguousarray(hits["t"])
    ubits = np.uint32 if rt.dtype.itemsize == 4 else np.uint64
    same_bits = rt.view(ubits) == ht.view(ubits)
    # (the one pair of equal values with different bits is +0.0 / -0.0: two primitives met at t == 0, e.g. a ray that starts
    # on a shared vertex — a tie like any other, verified below)
    t_ok = same_bits | (rt == ht)
    assert t_ok.all(), "t differs at %s" % np.nonzero(~t_ok)[0][:8]
    diff_mask = (ref_hits["prim_id"] != hits["prim_id"]) | ~same_bits
    diff = np.nonzero(diff_mask)[0]
    same = ~diff_mask
    assert np.array_equal(ref_hits["u"][same], hits["u"][same]) and np.array_equal(ref_hits["v"][same], hits["v"][same]), \
        "u/v differ on rays that report the same primitive"
    if max_ties is not None:
        assert diff.size <= max_ties, "%d prim_id mismatches" % diff.size
    for i in diff:
        o = default_trace_options() if base_opts is None else base_opts.copy()
        p = int(hits["prim_id"][i])
        lo = max(p, int(o["prim_ids_range"][0]))
        hi = min(p + 1, int(o["prim_ids_range"][1]))
        o["prim_ids_range"] = (lo, hi)
        h1, m1 = oracle.traverse(onodes, oindices, verts, faces, rays[i:i + 1], o)
        assert m1[0] == 1 and _fields_equal(h1, hits[i:i + 1]), \
            "ray %d: reported prim %d is not an exact tie of the reference's prim %d" % (i, p, ref_hits["prim_id"][i])
    return int(diff.size)


def walk_order_bits(kernel_name):
    """Template arguments of k_traverse_wide: <T, STACK, STATS, KIND, PLAIN, CLOCK, WIDTH, ORDER> -> (WIDTH, ORDER).  ORDER bit 0:
    slots entered by entry distance (opt-in, tunable order4); bit 1: the leaf phase over items (tunable leaf_compact, default on —
    records bit-identical either way)."""
    args = kernel_name.split("<", 1)[1].rstrip(">").split(", ")
    return int(args[6]), int(args[7])


def is_reference_order_two_level_walk(kernel_name):
    w, o = walk_order_bits(kernel_name)
    return w == 4 and (o & 1) == 0


def is_distance_order_two_level_walk(kernel_name):
    w, o = walk_order_bits(kernel_name)
    return w == 4 and (o & 1) == 1
