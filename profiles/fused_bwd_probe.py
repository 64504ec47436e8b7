#!/usr/bin/env python3
"""Each one-pass backward kernel (lshm_conv_bwd_fused) alone on the stream against the two kernels it replaces
(lshm_conv_wgrad + lshm_conv_dgrad, back to back), B = 256, HIP-event timing; algorithmic bytes = read dz + saved
input once, write the data gradient.  Usage: python profiles/fused_bwd_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lshm_amd import _lib as L
lib = L.load()
B, dev = 256, "cuda"
CH = (4, 8, 12, 24, 48, 96, 192)

def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

cases = [("1d tconv5", 3, 5, 1), ("1d tconv4", 3, 4, 1), ("1d conv1", 2, 1, 1), ("1d conv0", 2, 0, 0), ("2d tconv5", 1, 5, 1)]
for name, kind, i, elu in cases:
    tr = kind in (1, 3)
    cin, cout = (CH[6 - i], CH[5 - i]) if tr else (CH[i], CH[i + 1])
    if kind < 2:
        hw = (2 << i) if tr else (128 >> i)
        ishape, Hin, Win = (B, cin, hw, hw), hw, hw
        oshape = (B, cout, 2 * hw, 2 * hw) if tr else (B, cout, hw // 2, hw // 2)
        wshape = ((cin, cout) if tr else (cout, cin)) + (4, 4)
    else:
        Ln = (4 << (2 * i)) if tr else (16384 >> (2 * i))
        ishape, Hin, Win = (B, cin, Ln), 1, Ln
        oshape = (B, cout, 4 * Ln) if tr else (B, cout, Ln // 4)
        wshape = ((cin, cout) if tr else (cout, cin)) + (4,)
    x = torch.nn.functional.elu(torch.randn(ishape, device=dev)); dz = torch.randn(oshape, device=dev)
    w = torch.randn(wshape, device=dev) * 0.1
    dw, db, dx = torch.empty_like(w), torch.empty(cout, device=dev), torch.empty_like(x)
    nws = lib.lshm_conv_workspace_floats(kind, B, cin, cout, Hin, Win)
    ws = torch.empty(nws, device=dev); st = L.stream(); P = L.ptr
    def fused():
        L.check(lib.lshm_conv_bwd_fused(kind, P(x), P(dz), P(w), P(dw), P(db), P(dx), elu, B, cin, cout, Hin, Win, P(ws), nws, st))
    def wg():
        L.check(lib.lshm_conv_wgrad(kind, P(x), P(dz), P(dw), P(db), B, cin, cout, Hin, Win, 0, 0, P(ws), nws, 0, st))
    def dg():
        L.check(lib.lshm_conv_dgrad(kind, P(dz), P(w), P(dx), P(x) if elu else None, B, cin, cout, Hin, Win, 0, 0, P(ws), nws, st))
    tf, tw, td = timeit(fused), timeit(wg), timeit(dg)
    mb = 4 * (2 * x.numel() + dz.numel()) / 1e6
    print(f"{name:10s} fused {tf:7.1f} us ({mb/tf*1e-3*1e3:6.0f} GB/s of {mb:.0f} MB)   wgrad {tw:7.1f} + dgrad {td:7.1f} = {tw+td:7.1f} us")
