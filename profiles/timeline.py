#!/usr/bin/env python3
"""Phase breakdown of one training step from a rocprofv3 kernel trace (multi-stream aware):
wall time between marker kernels, summed kernel time and stream overlap inside each phase.
Usage: timeline.py DIR NSTEPS"""
import csv, glob, sys
d, nsteps = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(glob.glob(d + "/*/*kernel_trace.csv")[0])))
rows = [r for r in rows if 'lshm' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last step = from the last recon_kernel's preceding uv_harmonics to the end
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
per = idx[-1] - idx[-2]
end = len(rows)
start = end - per
last = rows[start:end]
t0 = int(last[0]['Start_Timestamp'])
markers = ['uv_harmonics', 'residual_split', 'recon_kernel', 'khm', 'combine_dx1', 'adam_kernel', 'multiplier_update']
phase_start = t0
phase_name = 'start'
busy = 0
acc = []
cur_end = t0
ksum = 0
for r in last:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    nm = r['Kernel_Name']
    hit = [m for m in markers if m in nm]
    if hit and (hit[0] != phase_name.split('#')[0]):
        acc.append((phase_name, (s - phase_start) / 1e3, ksum / 1e3))
        phase_start, phase_name, ksum = s, hit[0], 0
    ksum += e - s
acc.append((phase_name, (int(last[-1]['End_Timestamp']) - phase_start) / 1e3, ksum / 1e3))
tot = (int(last[-1]['End_Timestamp']) - t0) / 1e3
print(f"step wall {tot:.1f} us, kernels in step {len(last)}")
for n, w, k in acc:
    print(f"  phase from {n:18s} wall {w:8.1f} us   sum of kernel time {k:8.1f} us")
