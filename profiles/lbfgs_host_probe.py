#!/usr/bin/env python3
"""Where does an LBFGS iteration (LBFGSNew: history 7, max_iter 4, line search, batch mode) spend its 18.5 ms?
cProfile of the host side of three iterations at B=256, K=10 + the closure counts.
Usage: python profiles/lbfgs_host_probe.py"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lshm_amd import KHarmonicTrainer, TrainConfig  # noqa: E402

dev = torch.device("cuda:0")
tr = KHarmonicTrainer(TrainConfig(Kc=10), batch=256, batch_per_bline=8, default_batch=32, device=dev)
tr.init_parameters(seed=0)
g = torch.Generator().manual_seed(0)
tr.new_minibatch(torch.randn(256, 4, 128, 128, generator=g).to(dev), (1000.0 * torch.randn(256, 2, generator=g)).to(dev))
opt = tr.make_lbfgs()
for _ in range(2):
    tr.step_lbfgs(opt)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    tr.step_lbfgs(opt)
torch.cuda.synchronize()
pr.disable()
print(f"{1e3 * (time.perf_counter() - t0) / 3:.2f} ms per iteration (profiled)")
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
