#!/usr/bin/env python3
"""Print calls / average us of kernels whose name contains PATTERN from a rocprofv3 --stats directory.
Usage: kstat.py DIR PATTERN [PATTERN ...]"""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(p in r["Name"] for p in sys.argv[2:]):
            n = r["Name"].replace("lshm::", "").replace("void ", "").split("(")[0]
            print(f"{n:60s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:8.1f} us")
