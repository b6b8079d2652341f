#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv as ms/step:  python tools/kstats.py FILE [steps_incl_warmup=13] [rows=45]"""
import csv, sys
f = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 13.0; rows = int(sys.argv[3]) if len(sys.argv) > 3 else 45
tot = 0.0
for i, r in enumerate(csv.DictReader(open(f))):
    ms = float(r["TotalDurationNs"]) / 1e6 / steps; tot += ms
    if i < rows:
        print("%-96s %7.1f %9.3f %9.1f" % (r["Name"][:96], float(r["Calls"]) / steps, ms, float(r["AverageNs"]) / 1e3))
print("total kernel ms/step: %.2f" % tot)
