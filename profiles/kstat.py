#!/usr/bin/env python3
"""Print calls / average us of kernels whose name contains PATTERN from a rocprofv3 --stats directory.
Usage: kstat.py DIR PATTERN [PATTERN ...]"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from names import short
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(p in r["Name"] for p in sys.argv[2:]):
            n = short(r["Name"])
            print(f"{n:60s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:8.1f} us")
