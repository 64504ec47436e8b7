#!/usr/bin/env python3
"""Where one ADMM iteration spends its time WITHOUT a profiler: device timestamps at the phase boundaries
(LSHM_SCHED_PHASE_EVENTS in the engine's schedule word -> lshm_engine_phase_times, include/lshm.h), B = 256, K = 10, fp32, the default schedule.
Each mark is a hipEventRecord (a marker packet: ~6 us of idle queue on its stream), so the instrumented iteration is
a little slower than the bench line's.  Usage: [PROBE_SCHEDULE_OFF=name,..] python profiles/phase_times_probe.py"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lshm_amd import KHarmonicTrainer, TrainConfig, _lib as L

B, dev = 256, torch.device("cuda:0")
cfg = TrainConfig(Kc=10, schedule_off=("phase_events",) + tuple(n for n in os.environ.get("PROBE_SCHEDULE_OFF", "").split(",") if n),
                  tune=int(os.environ.get("PROBE_TUNE", 0)))
tr = KHarmonicTrainer(cfg, batch=B, batch_per_bline=8, default_batch=B // 8, device=dev)
tr.init_parameters(seed=0)
gen = torch.Generator(device="cpu").manual_seed(1234)
x = torch.randn(B, cfg.num_in_channels, 128, 128, generator=gen)
x = (x - x.mean()) / x.std()
uv = 1000.0 * torch.randn(B, 2, generator=gen)
tr.new_minibatch(x.to(dev), uv.to(dev))
names = ["closure entry", "1-D backward done (main stream)", "backward done (main stream)", "weight-gradient stream done",
         "closure end (streams joined)", "update entry (after Adam)", "closure forward done", "no-grad forward done",
         "reconstruction pass done", "update end"]
lib = L.load()
acc = [0.0] * 10
nrep = 0
for it in range(60):
    tr.step()
    if it >= 20 and it % 4 == 0:  # read (= synchronise) only now and then, so that the host is ahead again in between
        ms = (C.c_float * 10)()
        L.check(lib.lshm_engine_phase_times(tr._h, ms, 10), "phase_times")
        for i in range(10):
            acc[i] += ms[i]
        nrep += 1
print(f"phase boundaries of one ADMM iteration, ms after the closure's first launch (mean of {nrep} instrumented iterations, each read 3 iterations after the previous synchronisation):")
prev = 0.0
for i, n in enumerate(names):
    v = acc[i] / nrep
    print(f"  {v:8.3f}  {n}")
print("derived: backward (main) %.3f, side stream tail after the main stream %.3f, Adam + joins %.3f, paired forwards %.3f (closure) / %.3f (no-grad), reconstruction pass %.3f, iteration %.3f"
      % (acc[2] / nrep, (acc[3] - acc[2]) / nrep, (acc[5] - acc[4]) / nrep, (acc[6] - acc[5]) / nrep, (acc[7] - acc[5]) / nrep,
         (acc[8] - acc[7]) / nrep, acc[9] / nrep))
