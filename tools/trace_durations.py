#!/usr/bin/env python3
"""print per-dispatch durations (ms) of the pfmscan kernels from a rocprofv3 kernel trace dir"""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    d = [(r["Kernel_Name"].split("(")[0][-40:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
         for r in csv.DictReader(open(f)) if "pfmscan" in r["Kernel_Name"]]
    print(f)
    print([round(x[1], 3) for x in d])
