#!/usr/bin/env python3
"""The two LDS-resident chains of the 1-D mid layers (lshm_conv1d_chain3), each alone on the stream, against the three
launches they replace (lshm_conv_fwd x 3, paired problems as in the step: netT and netF share a launch), B = 256,
HIP-event timing.  Under `rocprofv3 --pmc ...` the same script gives the counters of the chain kernels.
Usage: python profiles/chain_probe.py"""
import ctypes as C, os, sys
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

for up in (0, 1):
    ch = (96, 48, 24, 12) if up else (12, 24, 48, 96)
    Ls = [16, 64, 256, 1024] if up else [1024, 256, 64, 16]
    kind = 3 if up else 2
    x = torch.randn(B, ch[0], Ls[0], device=dev)
    w = [torch.randn((ch[k], ch[k + 1], 4) if up else (ch[k + 1], ch[k], 4), device=dev) * 0.1 for k in range(3)]
    b = [torch.zeros(ch[k + 1], device=dev) for k in range(3)]
    y = [torch.empty(B, ch[k + 1], Ls[k + 1], device=dev) for k in range(3)]
    arr = lambda ts: (C.c_void_p * 3)(*[t.data_ptr() for t in ts])
    st = L.stream()
    def chain():
        L.check(lib.lshm_conv1d_chain3(up, L.ptr(x), arr(w), arr(b), arr(y), None, 1, 0 if up else 1, B, st))
    wsz = [lib.lshm_conv_workspace_floats(kind, B, ch[k], ch[k + 1], 1, Ls[k]) for k in range(3)]
    ws = torch.empty(max(max(wsz), 1), device=dev)
    def three():
        cur = x
        for k in range(3):
            L.check(lib.lshm_conv_fwd(kind, L.ptr(cur), L.ptr(w[k]), L.ptr(b[k]), L.ptr(y[k]), B, ch[k], ch[k + 1], 1, Ls[k], 0, 0, 1,
                                      L.ptr(ws), wsz[k], st))
            cur = y[k]
    tc, t3 = timeit(chain), timeit(three)
    flop = 2.0 * B * sum(ch[k] * ch[k + 1] * 4 * (Ls[k] if up else Ls[k + 1]) for k in range(3))
    print(f"{'up  ' if up else 'down'} chain {tc:6.1f} us ({flop/tc*1e-6:5.1f} TFLOP/s)   three launches {t3:6.1f} us   (one problem; the step runs netT and netF as a pair)")
