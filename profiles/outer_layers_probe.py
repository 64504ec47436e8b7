#!/usr/bin/env python3
"""Launch fwd / dgrad / wgrad of the four outer 2-D layers (the direct LDS-patch kernels) a few times:
target of `rocprofv3 --pmc ...` passes (dev aid).  Usage: outer_layers_probe.py [B]"""
import sys
import torch
sys.path.insert(0, ".")
from lshm_amd import _lib as L

lib = L.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda")
P = L.ptr
for kind, Cin, Cout, H in ((0, 4, 8, 128), (0, 8, 12, 64), (1, 12, 8, 32), (1, 8, 4, 64)):
    Ho = H // 2 if kind == 0 else H * 2
    x = torch.randn(B, Cin, H, H, device=dev)
    w = torch.randn(*((Cout, Cin) if kind == 0 else (Cin, Cout)), 4, 4, device=dev) * 0.1
    b = torch.zeros(Cout, device=dev)
    y = torch.empty(B, Cout, Ho, Ho, device=dev)
    dz = torch.randn(B, Cout, Ho, Ho, device=dev)
    dx = torch.empty_like(x)
    dw, db = torch.empty_like(w), torch.empty_like(b)
    nws = lib.lshm_conv_workspace_floats(kind, B, Cin, Cout, H, H)
    ws = torch.empty(nws, device=dev)
    st = L.stream()
    for _ in range(4):
        L.check(lib.lshm_conv_fwd(kind, P(x), P(w), P(b), P(y), B, Cin, Cout, H, H, 0, 0, 1, P(ws), nws, st))
        L.check(lib.lshm_conv_dgrad(kind, P(dz), P(w), P(dx), P(x), B, Cin, Cout, H, H, 0, 0, P(ws), nws, st))
        L.check(lib.lshm_conv_wgrad(kind, P(x), P(dz), P(dw), P(db), B, Cin, Cout, H, H, 0, 0, P(ws), nws, 0, st))
    torch.cuda.synchronize()
