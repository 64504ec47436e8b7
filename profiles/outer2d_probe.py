#!/usr/bin/env python3
"""The outer layers of the 2-D autoencoder (conv0, conv1, tconv4, tconv5; src/lofar_models.py:31-33,56-57), each kernel
alone on the stream at B = 256: forward, data gradient, weight gradient; HIP-event timing and the algorithmic bytes
each moves.  Usage: python profiles/outer2d_probe.py"""
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

for name, kind, i in [("conv0", 0, 0), ("conv1", 0, 1), ("tconv4", 1, 4), ("tconv5", 1, 5)]:
    tr = kind == 1
    cin, cout = (CH[6 - i], CH[5 - i]) if tr else (CH[i], CH[i + 1])
    hw = (2 << i) if tr else (128 >> i)
    ishape = (B, cin, hw, hw)
    oshape = (B, cout, 2 * hw, 2 * hw) if tr else (B, cout, hw // 2, hw // 2)
    wshape = ((cin, cout) if tr else (cout, cin)) + (4, 4)
    x = torch.nn.functional.elu(torch.randn(ishape, device=dev)); dz = torch.randn(oshape, device=dev)
    y = torch.empty(oshape, device=dev)
    w = torch.randn(wshape, device=dev) * 0.1; bias = torch.zeros(cout, device=dev)
    dw, db, dx = torch.empty_like(w), torch.empty(cout, device=dev), torch.empty_like(x)
    nws = lib.lshm_conv_workspace_floats(kind, B, cin, cout, hw, hw)
    ws = torch.empty(nws, device=dev); st = L.stream(); P = L.ptr
    def fw():
        L.check(lib.lshm_conv_fwd(kind, P(x), P(w), P(bias), P(y), B, cin, cout, hw, hw, 0, 0, 1, P(ws), nws, st))
    def wg():
        L.check(lib.lshm_conv_wgrad(kind, P(x), P(dz), P(dw), P(db), B, cin, cout, hw, hw, 0, 0, P(ws), nws, 0, st))
    def dg():
        L.check(lib.lshm_conv_dgrad(kind, P(dz), P(w), P(dx), P(x), B, cin, cout, hw, hw, 0, 0, P(ws), nws, st))
    tf, tw, td = timeit(fw), timeit(wg), timeit(dg)
    mx, mz = 4 * x.numel() / 1e6, 4 * dz.numel() / 1e6
    print(f"2d {name:7s} fwd {tf:6.1f} us ({(mx+mz)/tf:5.2f} TB/s of {mx+mz:4.0f} MB)  dgrad {td:6.1f} us ({(2*mx+mz)/td:5.2f} TB/s of {2*mx+mz:4.0f} MB)"
          f"  wgrad {tw:6.1f} us ({(mx+mz)/tw:5.2f} TB/s)")
