#!/usr/bin/env python3
"""Chronological listing of the kernels of the last training step in a rocprofv3 kernel trace,
with the queue each ran on, its start offset, duration and the idle gap since the previous kernel
on the same queue.  Usage: step_trace.py DIR > listing.txt"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from names import short
d = sys.argv[1]
files = glob.glob(d + "/*kernel_trace.csv") + glob.glob(d + "/*/*kernel_trace.csv")
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for r in rows:
    r['Kernel_Name'] = short(r['Kernel_Name']) if r['Kernel_Name'].strip('"').startswith('_ZN4lshm') else r['Kernel_Name']
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
# a step = one Adam update and everything up to the next one (round 3: the closure forward of iteration k+1 starts
# beside the no-grad forward of iteration k, so the listing runs Adam -> [two forwards] -> reconstruction pass ->
# backward; the wall time between two Adam launches is the iteration time)
start, end = idx[-3], idx[-2]
last = rows[start:end]
t0 = int(last[0]['Start_Timestamp'])
qkey = 'Queue_Id' if 'Queue_Id' in last[0] else 'Stream_Id'
prev_end = {}
busy_union = 0
cur_s, cur_e = None, None
for r in last:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy_union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy_union += cur_e - cur_s
wall = int(rows[end]['Start_Timestamp']) - t0
print(f"# step wall {wall/1e3:.1f} us, {len(last)} kernels, device busy (union) {busy_union/1e3:.1f} us, "
      f"sum {sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in last)/1e3:.1f} us")
for r in last:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    q = r[qkey]
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = e
    print(f"{(s-t0)/1e3:9.1f} q={q:>3s} dur={(e-s)/1e3:7.1f} gap={gap:7.1f} grid={r.get('Grid_Size_X','?'):>8s} {r['Kernel_Name'][:90]}")
# per-kernel totals inside that step
agg = {}
for r in last:
    n = r['Kernel_Name'].replace('lshm::', '').replace('void ', '')
    n = n.split('(')[0]
    a = agg.setdefault(n, [0, 0])
    a[0] += 1
    a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print("# ---- totals by kernel within the step")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"# {n:70s} n={c:3d} sum={t/1e3:8.1f} us avg={t/c/1e3:7.1f} us")
