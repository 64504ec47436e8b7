#!/usr/bin/env python3
"""MFMA-pipe utilisation per kernel: SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs of the chip)
divided by 1024 x the dispatch's cycles.  GRBM_GUI_ACTIVE of the same rocprofv3 --pmc pass is summed over the
8 XCDs, so cycles = GRBM_GUI_ACTIVE / 8 (cross-check column: the kernel-trace duration x 2.1 GHz).
Usage: mfma_util.py PMC_DIR KERNEL_STATS_CSV"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from names import short
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
stats = {}
for r in csv.DictReader(open(sys.argv[2])):
    k = short(r["Name"])
    stats[k] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]))
print(f"{'kernel':62s} {'avg us':>8s} {'% time':>7s} {'MFMA insts':>11s} {'busy cyc':>11s} {'GUI/8 cyc':>10s} {'MFMA util':>10s} {'(by time)':>10s}")
for k, (calls, avg, pct) in sorted(stats.items(), key=lambda kv: -kv[1][2])[:30]:
    c = acc.get(k)
    if not c or "at::" in k:
        continue
    busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
    gui = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
    insts = sum(c["SQ_INSTS_MFMA"]) / len(c["SQ_INSTS_MFMA"])
    util = busy / (1024.0 * gui / 8.0) if gui else 0.0
    util_t = busy / (1024.0 * avg * 2100.0)
    print(f"{k:62s} {avg:8.1f} {pct:7.1f} {insts:11.0f} {busy:11.0f} {gui/8:10.0f} {100*util:9.1f}% {100*util_t:9.1f}%")
