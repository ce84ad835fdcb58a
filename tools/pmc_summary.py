#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes: per kernel name, mean counter value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row.get("Kernel_Name", "?")
                short = k.split("(")[0].replace("void ", "")
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc, key=lambda k: -len(acc[k])):
        if "k_traverse" not in k and "k_subtree" not in k and "k_bin" not in k and "k_partition" not in k:
            continue
        print("==", k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print("   %-32s n=%-4d mean=%.4g" % (c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main(sys.argv[1])
