#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel totals per step and the
timeline of the last step.  Usage: tools_prof.py DIR NSTEPS [min_us]"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from names import short
d, nsteps = sys.argv[1], int(sys.argv[2])
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
st = (glob.glob(d + "/*kernel_stats.csv") + glob.glob(d + "/*/*kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(st)))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print(f"total kernel time per step: {tot/nsteps/1e6:.3f} ms")
for r in rows[:40]:
    n = r['Name'].replace('lshm::', '').replace('void ', '')[:84]
    print(f"{n:86s} calls/step={int(r['Calls'])/nsteps:6.1f} ms/step={int(r['TotalDurationNs'])/nsteps/1e6:7.3f} avg={float(r['AverageNs'])/1e3:8.1f}us {float(r['Percentage']):5.1f}%")
if min_us >= 0:
    tr = (glob.glob(d + "/*kernel_trace.csv") + glob.glob(d + "/*/*kernel_trace.csv"))[0]
    rows = list(csv.DictReader(open(tr)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    per = len(rows) // nsteps
    last = rows[-per:]
    t0 = int(last[0]['Start_Timestamp'])
    print("--- last step, kernels >=", min_us, "us")
    for r in last:
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        if dur < min_us: continue
        nm = r['Kernel_Name'].replace('lshm::', '').replace('void ', '')[:60]
        gx = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']); gy = int(r['Grid_Size_Y']) // max(1, int(r['Workgroup_Size_Y']))
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {dur:8.1f}us grid={gx:5d}x{gy:3d}x{r['Grid_Size_Z']:>4s} vgpr={r['VGPR_Count']:>3s} {nm}")
