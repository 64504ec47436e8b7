#!/usr/bin/env python3
"""Mean PMC counter values per kernel from rocprofv3 --pmc output directories.
Usage: pmc_by_kernel.py DIR [DIR ...]"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from names import short
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in sorted(acc.items()):
    if 'at::' in k or 'elementwise' in k: continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {sum(v)/len(v):16.0f}   (n={len(v)})")
