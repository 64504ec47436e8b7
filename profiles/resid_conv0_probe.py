#!/usr/bin/env python3
"""lshm_resid_conv0 (conv0 of netT / netF straight from x and the 2-D reconstruction) alone on the stream against the two
launches it replaces (lshm_residual_split + lshm_conv_fwd_pair), B = 256, HIP-event timing.
Usage: python profiles/resid_conv0_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lshm_amd import _lib as L
lib = L.load()
B, dev = 256, "cuda"

def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

x, x1 = torch.randn(B, 4, 128, 128, device=dev), torch.randn(B, 4, 128, 128, device=dev)
w = [torch.randn(8, 4, 4, device=dev) * 0.3 for _ in range(2)]
b = [torch.randn(8, device=dev) * 0.1 for _ in range(2)]
y = [torch.empty(B, 8, 4096, device=dev) for _ in range(2)]
row, col = torch.empty(B, 4, 16384, device=dev), torch.empty(B, 4, 16384, device=dev)
st, P = L.stream(), L.ptr
def fused():
    L.check(lib.lshm_resid_conv0(P(x), P(x1), P(w[0]), P(b[0]), P(y[0]), P(w[1]), P(b[1]), P(y[1]), B, st))
def rs():
    L.check(lib.lshm_residual_split(P(x), P(x1), P(row), P(col), B * 4, 128, st))
def cv():
    L.check(lib.lshm_conv_fwd_pair(2, P(row), P(w[0]), P(b[0]), P(y[0]), P(col), P(w[1]), P(b[1]), P(y[1]), B, 4, 8, 1, 16384, 0, 0, 1, None, 0, st))
tf, t1, t2 = timeit(fused), timeit(rs), timeit(cv)
print(f"resid_conv0 {tf:6.1f} us ({335.5/tf:5.2f} TB/s of 336 MB)   residual_split {t1:6.1f} us + conv0 pair {t2:6.1f} us = {t1+t2:6.1f} us (470 MB)")
