#!/usr/bin/env python3
"""How far ahead of the device does the host run?  Enqueue N ADMM iterations without synchronising and time (a) the
host loop alone, (b) loop + final synchronise.  (a) << (b): the device is the bottleneck; (a) ~ (b): the launch path is.
Usage: python profiles/host_ahead_probe.py [N=20]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lshm_amd import KHarmonicTrainer, TrainConfig  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
tr = KHarmonicTrainer(TrainConfig(Kc=10), batch=256, batch_per_bline=8, default_batch=32, device=dev)
tr.init_parameters(seed=0)
g = torch.Generator().manual_seed(0)
tr.new_minibatch(torch.randn(256, 4, 128, 128, generator=g).to(dev), (1000.0 * torch.randn(256, 2, generator=g)).to(dev))
for _ in range(5):
    tr.step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(N):
        tr.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host enqueue {1e3 * (t1 - t0) / N:.3f} ms/iteration, with device {1e3 * (t2 - t0) / N:.3f} ms/iteration")
# per phase: host time of the three calls of an iteration (closure backward | Adam | forwards + reconstruction pass)
acc = [0.0, 0.0, 0.0]
torch.cuda.synchronize()
for _ in range(N):
    t0 = time.perf_counter(); tr._closure_fwd_bwd()
    t1 = time.perf_counter(); tr._adam()
    t2 = time.perf_counter(); tr._multipliers()
    t3 = time.perf_counter()
    acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2
torch.cuda.synchronize()
print("host time per call: closure (backward of the saved forward) %.3f ms, Adam %.3f ms, multiplier update + next forwards %.3f ms"
      % tuple(1e3 * a / N for a in acc))
