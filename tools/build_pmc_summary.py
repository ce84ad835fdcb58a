"""Per-kernel summary of tools/build_pmc.sh's counter passes: for every build kernel the launches per build, the mean duration
and the counter SUMS PER BUILD (four builds traced; the first is the cold one and is left out), with the derived figures the
review asked for: LDS instructions per VALU instruction, bank-conflict cycles per LDS instruction, share of the wave cycles
spent waiting, HBM-side bytes (FETCH_SIZE KiB x 1024 x 2 on gfx950 + WRITE_SIZE KiB x 1024).
    python tools/build_pmc_summary.py gpurun_out/build_pmc_TAG"""
import csv, glob, os, sys

root = sys.argv[1]
BUILDS = 4


def short(n):
    return n.replace("void ", "").split("(")[0].replace("nrt::", "").split("<")[0]


per = {}    # kernel -> counter -> sum over the warm builds
durs = {}   # kernel -> [durations us]
launches = {}
for d in sorted(glob.glob(os.path.join(root, "p*"))):
    if not os.path.isdir(d):
        continue
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not cc:
        continue
    rows = list(csv.DictReader(open(cc[0])))
    # dispatches in order; a build starts at k_init_scene (or the first k_prim_records): skip the first (cold) build
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    name_of = {int(r["Dispatch_Id"]): short(r["Kernel_Name"]) for r in rows}
    starts = [i for i in ids if name_of[i].startswith("k_prim_records")]
    if len(starts) < 2:
        continue
    first_warm = starts[1]
    nb = len(starts) - 1
    for r in rows:
        i = int(r["Dispatch_Id"])
        if i < first_warm:
            continue
        k = name_of[i]
        per.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
        per[k][r["Counter_Name"]] += float(r["Counter_Value"]) / nb
    if kt and not durs:
        tr = list(csv.DictReader(open(kt[0])))
        tr.sort(key=lambda r: int(r["Start_Timestamp"]))
        names = [short(r["Kernel_Name"]) for r in tr]
        st = [j for j, n in enumerate(names) if n.startswith("k_prim_records")]
        if len(st) >= 2:
            for j in range(st[1], len(tr)):
                durs.setdefault(names[j], []).append((int(tr[j]["End_Timestamp"]) - int(tr[j]["Start_Timestamp"])) * 1e-3)
            for k, v in durs.items():
                launches[k] = len(v) / float(len(st) - 1)

tot_us = sum(sum(v) for v in durs.values()) / max(1, BUILDS - 1)
print("kernels of one warm build (under the profiler): %.1f us of kernel time" % tot_us)
print("%-22s %6s %9s %9s | per build: %11s %11s %9s %9s %9s %8s %8s %10s" % (
    "kernel", "launch", "us/launch", "us/build", "VALU insts", "LDS insts", "LDS/VALU", "bankcf/LDS", "wait", "fetch MB", "write MB", "HBM GB/s"))
for k in sorted(durs, key=lambda k: -sum(durs[k])):
    c = per.get(k, {})
    n = launches.get(k, 0)
    us_build = sum(durs[k]) / max(1, BUILDS - 1)
    valu, lds = c.get("SQ_INSTS_VALU"), c.get("SQ_INSTS_LDS")
    bank = c.get("SQ_LDS_BANK_CONFLICT")
    wait, wc = c.get("SQ_WAIT_ANY"), c.get("SQ_WAVE_CYCLES")
    f, w = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
    fmb = f * 1024 * 2 / 1e6 if f is not None else None
    wmb = w * 1024 / 1e6 if w is not None else None
    gbs = (fmb + wmb) / us_build * 1e-3 * 1e3 if (fmb is not None and wmb is not None and us_build > 0) else None  # MB/us = TB/s -> GB/s x1000

    def fm(x, p="%.3g"):
        return "-" if x is None else p % x

    print("%-22s %6.1f %9.1f %9.1f | %22s %11s %9s %9s %9s %8s %8s %10s" % (
        k, n, us_build / max(n, 1e-9), us_build, fm(valu, "%.4g"), fm(lds, "%.4g"),
        fm(lds / valu if valu and lds is not None else None), fm(bank / lds if lds and bank is not None else None),
        fm(wait / wc if wc and wait is not None else None), fm(fmb, "%.1f"), fm(wmb, "%.1f"), fm(gbs * 1e3 if gbs is not None else None, "%.0f")))
print()
print("raw per-build counter sums:")
for k in sorted(per):
    print(" ", k, {n: round(v, 1) for n, v in sorted(per[k].items())})
