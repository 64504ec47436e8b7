#!/usr/bin/env python3
"""Does running the host far ahead of the device cost device time?  The same N ADMM iterations (a) free-running,
(b) with a synchronise every iteration, (c) with the host held at most `lag` iterations ahead (an event per iteration).
Usage: python profiles/queue_depth_probe.py [N=40]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lshm_amd import KHarmonicTrainer, TrainConfig  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
tr = KHarmonicTrainer(TrainConfig(Kc=10), batch=256, batch_per_bline=8, default_batch=32, device=dev)
tr.init_parameters(seed=0)
g = torch.Generator().manual_seed(0)
tr.new_minibatch(torch.randn(256, 4, 128, 128, generator=g).to(dev), (1000.0 * torch.randn(256, 2, generator=g)).to(dev))
for _ in range(5):
    tr.step()
torch.cuda.synchronize()


def run(lag):
    evs = []
    t0 = time.perf_counter()
    for i in range(N):
        tr.step()
        if lag == 0:
            torch.cuda.synchronize()
        elif lag > 0:
            e = torch.cuda.Event()
            e.record()
            evs.append(e)
            if len(evs) > lag:
                evs.pop(0).synchronize()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / N


for rep in range(2):
    print(f"free-running {run(-1):.4f} ms | sync every iteration {run(0):.4f} | host <= 1 ahead {run(1):.4f} | <= 2 ahead {run(2):.4f}"
          f" | pool {os.environ.get('LSHM_EVENT_POOL', '128')}")
