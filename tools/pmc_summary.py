#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per-dispatch mean of every counter for the scan kernels."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
vals = defaultdict(list)
for path in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            if "pfmscan" not in name:
                continue
            kern = name.split("(")[0].replace("void pfmscan::", "")
            vals[(kern, row["Counter_Name"])].append(float(row["Counter_Value"]))
for (kern, ctr), v in sorted(vals.items()):
    print("%-60s %-24s n=%d mean=%.6g" % (kern, ctr, len(v), sum(v) / len(v)))
