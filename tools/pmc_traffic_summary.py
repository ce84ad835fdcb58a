#!/usr/bin/env python3
"""Per-launch HBM traffic of the traversal kernel from the PMC passes, calibrated on the stream-only launches."""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
prod = [k for k in acc if "k_traverse_wide<float, 10" in k]
cal = [k for k in acc if "k_traverse_wide<float, 16" in k]
out = {}
if prod and cal:
    P, C = acc[prod[0]], acc[cal[0]]
    mean = lambda v: sum(v) / len(v)
    n1 = 2073600
    known_read = n1 * 36.0          # every ray read once
    known_write = n1 * 17.0         # 16-B hit + 1-B mask per ray
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB
    cal_read = mean(C["FETCH_SIZE"]) * 1024.0
    cal_write = mean(C["WRITE_SIZE"]) * 1024.0
    fr, fw = known_read / cal_read, known_write / cal_write
    # production launches alternate primary, bounce
    f = P["FETCH_SIZE"]; w = P["WRITE_SIZE"]
    fetch = mean(f) * 1024.0; write = mean(w) * 1024.0
    out = {
        "kernel": prod[0],
        "raw_FETCH_SIZE_KiB_per_launch": mean(f), "raw_WRITE_SIZE_KiB_per_launch": mean(w),
        "calibration": {"known_read_bytes": known_read, "reported_read_bytes": cal_read, "read_factor": fr,
                        "known_write_bytes": known_write, "reported_write_bytes": cal_write, "write_factor": fw,
                        "note": "stream-only launches of the same kernel (NRT_DEBUG=6); MI355X guide: FETCH_SIZE under-reports wide reads by 2x on gfx950"},
        "hbm_bytes_per_launch": fetch * 2.0 + write,
        "hbm_bytes_per_launch_calibrated": fetch * fr + write * fw,
        "tcc_hit_rate": (mean(P["TCC_HIT_sum"]) / (mean(P["TCC_HIT_sum"]) + mean(P["TCC_MISS_sum"]))) if "TCC_HIT_sum" in P else None,
        "lds_bank_conflict_cycles": mean(P["SQ_LDS_BANK_CONFLICT"]) if "SQ_LDS_BANK_CONFLICT" in P else None,
        "lds_idx_active_cycles": mean(P["SQ_LDS_IDX_ACTIVE"]) if "SQ_LDS_IDX_ACTIVE" in P else None,
        "valu_lane_utilisation": (mean(P["SQ_THREAD_CYCLES_VALU"]) / (mean(P["SQ_INSTS_VALU"]) * 64.0)) if "SQ_INSTS_VALU" in P else None,
        "wait_fraction_of_wave_cycles": (mean(P["SQ_WAIT_ANY"]) / mean(P["SQ_WAVE_CYCLES"])) if "SQ_WAVE_CYCLES" in P else None,
    }
print(json.dumps(out, indent=1))
