"""The `roofline` object of the headline: which unit binds (computed from this run's counters, not assumed), and ONE chain of bytes
per launch — algorithmic -> requested (what the timed kernel asks of the L1) -> l1 (what the L1 looked up, counters)
-> fetched (HBM side, counters) -> compulsory (every byte once).  What the numbers mean is written up in DESIGN.md 3.1."""
from . import HBM_PEAK_GBS, L1_PEAK_GACC_S

L1_PIECE_BYTES = 16  # one tag look-up serves one lane's 16-byte piece (global_load_dwordx4) of a scattered record
L1_PEAK_GBS = L1_PEAK_GACC_S * L1_PIECE_BYTES  # 876 G look-ups/s x 16 B: the measured ceiling of tools/ubench/node_fetch.hip, in bytes


def io_bytes(wl, n):
    """ray record in + hit record and flag out, once"""
    return n * (wl.RAY.itemsize + wl.HIT.itemsize + 1)


def compulsory_bytes(wl, bytes1, bytes2):
    """Per average launch: every ray record read and every hit record + flag written once, plus the part of the private tree a
    launch can touch read ONCE (an upper bound: the whole Wide4Node / WideNode array and every leaf record, capped by what the
    walk fetched at all).  traffic / compulsory is the re-read factor."""
    branches = int(wl.stats["num_branch_nodes"])
    rec_b = 128 if wl.rb == 4 else 112
    tri_b = 40 if wl.rb == 4 else 80
    tree_once = branches * rec_b + wl.faces.shape[0] * tri_b
    per_ray = 52 if wl.rb == 4 else 104
    alg_tree = {"primary": bytes1 - per_ray * wl.n1, "bounce": bytes2 - per_ray * wl.n2}
    comp = sum(io_bytes(wl, n) + min(tree_once, max(0, alg_tree[w])) for w, n in (("primary", wl.n1), ("bounce", wl.n2))) / 2.0
    return comp, tree_once


def requested_bytes(wl, walk_counts):
    """What the TIMED kernel asks of the L1 per average launch, from the counting instantiation of the same kernel (per ray: records
    stepped through and leaf primitives tested): record bytes x steps + leaf-record bytes x primitives + ray in + record and flag out."""
    if not walk_counts:
        return None
    rec_b = walk_counts.get("record_bytes", 128 if wl.rb == 4 else 112)
    tri_b = 40 if wl.rb == 4 else 80
    tot = 0
    for w, n in (("primary", wl.n1), ("bounce", wl.n2)):
        c = walk_counts.get(w)
        if not c:
            return None
        tot += rec_b * c["steps"] + tri_b * c["prims"] + io_bytes(wl, n)
    return tot / 2.0


def headline(kernel_name, launch_ms, k_ms, alg_bytes_launch, alg_gbs, comp, counters, requested, build):
    """Assemble the object.  `counters` = roofline_from_counters(...) (or None: UNMEASURED); every fraction is of a stated peak."""
    roof = {"kernel": kernel_name, "launch_ms": round(launch_ms, 4),
            "per_wave_ms": {w: round(v, 4) for w, v in k_ms.items()}}
    hb = counters.get("hbm") if counters else None
    va = counters.get("valu") if counters else None
    l1 = counters.get("l1") if counters else None
    fracs = {}
    if hb:
        fracs["hbm"] = hb["frac"]
    if va:
        fracs["valu"] = va["frac"]
    if l1:
        fracs["l1"] = l1["frac"]
    # `bound`: the unit this run's counters show closest to its peak.  Without counters nothing is claimed.
    units = {
        "hbm": lambda: {"achieved": hb["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hb["frac"]},
        "l1": lambda: {"achieved": round(l1["achieved_Glookups_s"] * L1_PIECE_BYTES, 1), "peak": L1_PEAK_GBS, "unit": "GB/s", "frac": l1["frac"]},
        "valu": lambda: {"achieved": va["achieved_Tlaneops"], "peak": va["peak_Tlaneops"], "unit": "Tlane-op/s", "frac": va["frac"]},
    }
    if fracs:
        bound = max(fracs, key=fracs.get)
        roof["bound"] = bound
        roof.update(units[bound]())
    else:
        roof.update({"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None})
    roof["traffic"] = hb["bytes_per_launch"] if hb else None  # HBM-side bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE)
    roof["fracs"] = fracs
    # the contract's HBM figure, whichever unit binds
    roof["hbm"] = ({"achieved": hb["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hb["frac"], "l2_hit_rate": hb.get("l2_hit_rate")}
                   if hb else None)
    l1_bytes = l1["lookups_per_launch"] * L1_PIECE_BYTES if l1 else None
    chain = {"algorithmic": int(alg_bytes_launch), "requested": int(requested) if requested else None, "l1": int(l1_bytes) if l1_bytes else None,
             "fetched": roof["traffic"], "compulsory": int(comp)}
    roof["bytes_per_launch"] = chain
    ratios = {}
    if requested:
        ratios["algorithmic_over_requested"] = round(alg_bytes_launch / requested, 3)
        if roof["traffic"]:
            ratios["requested_over_fetched"] = round(requested / roof["traffic"], 3)
    if roof["traffic"]:
        ratios["fetched_over_compulsory"] = round(roof["traffic"] / comp, 3)
    roof["ratios"] = ratios
    roof["algorithmic_GBs"] = round(alg_gbs, 1)
    roof["algorithmic_x_hbm_peak"] = round(alg_gbs / HBM_PEAK_GBS, 4)
    if va:
        roof["lane_util"] = va["lane_util"]
        roof["issue_busy"] = va["issue_busy"]
        w = [p.get("wait_frac_of_wave_cycles") for p in va["per_wave"].values()]
        if all(x is not None for x in w):
            roof["wait"] = round(sum(w) / len(w), 4)
        b = [p.get("lds_bank_conflict_cycles") for p in va["per_wave"].values()]
        if all(x is not None for x in b):
            roof["lds_bank_conflict_cycles"] = int(sum(b))
    roof["build"] = build
    roof["source"] = "in-run rocprofv3 --pmc passes" if counters and fracs else "UNMEASURED (no counter pass in this run)"
    return roof
