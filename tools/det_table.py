#!/usr/bin/env python3
"""Per-pass table of a detector kernel trace (rocprofv3 --kernel-trace --stats CSV of tools/prof_det.py B reps): python tools/det_table.py file.csv [reps=5]"""
import csv
import sys

reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if "rocclr" in n:
        continue
    c, a = int(r["Calls"]), float(r["AverageNs"]) / 1000
    per = c / reps * a
    tot += per
    print("%-88s %4.1f x %7.1f us = %7.1f /pass  (min %6.1f max %6.1f)" % (n[:88], c / reps, a, per, float(r["MinNs"]) / 1000, float(r["MaxNs"]) / 1000))
print("kernels per pass: %.1f us" % tot)
