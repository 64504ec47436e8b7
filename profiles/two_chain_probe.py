#!/usr/bin/env python3
"""Probe: does running the minibatch as two independent half-batch chains (two engines, two streams, two
host threads) overlap the latency-bound deep layers of one half with the bandwidth-bound outer layers
of the other?  Gradients are NOT combined here -- this only measures the achievable overlap."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshm_amd import KHarmonicTrainer, TrainConfig

dev = torch.device("cuda:0")
def make(B, seed):
    tr = KHarmonicTrainer(TrainConfig(Kc=10), batch=B, batch_per_bline=8, default_batch=B // 8, device=dev)
    tr.init_parameters(seed=0)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 4, 128, 128, generator=g)
    uv = 1000.0 * torch.randn(B, 2, generator=g)
    tr.new_minibatch(x.to(dev), uv.to(dev))
    return tr

def run(trs, steps, warm=20):
    streams = [torch.cuda.Stream() for _ in trs]
    def work(tr, s, n):
        with torch.cuda.stream(s):
            for _ in range(n):
                tr.step()
    for n in (warm, steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(tr, s, n)) for tr, s in zip(trs, streams)]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dt / steps * 1e3

steps = 200
one = make(256, 1)
print("one chain  B=256: %.3f ms/step" % run([one], steps))
del one
a, b = make(128, 1), make(128, 2)
print("one chain  B=128: %.3f ms/step" % run([a], steps))
print("two chains B=128: %.3f ms/step for 256 patches" % run([a, b], steps))
