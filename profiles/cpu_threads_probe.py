import sys, time, torch
sys.path.insert(0, ".")
import bench
class A: batch=256; K=10; bpb=8; cpu_steps=1
import os
print("cpus", os.cpu_count())
for n in (8, 16, 32, 64):
    torch.set_num_threads(n)
    t0=time.time(); r=bench.cpu_baseline(A); print(n, r["value"], r["s_per_step"], "wall", round(time.time()-t0,1), flush=True)
