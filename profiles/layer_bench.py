#!/usr/bin/env python3
"""Per-layer micro-benchmark through the C ABI (dev aid): us per launch of fwd / dgrad / wgrad for
every layer shape of the autoencoders at batch B."""
import sys
import torch
sys.path.insert(0, ".")
from lshm_amd import _lib as L

lib = L.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda")
CH = (4, 8, 12, 24, 48, 96, 192)


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def layer(kind, Cin, Cout, Hin, Win):
    nd2 = kind < 2
    ishape = (B, Cin, Hin, Win) if nd2 else (B, Cin, Win)
    if kind == 0: oshape = (B, Cout, Hin // 2, Win // 2)
    elif kind == 1: oshape = (B, Cout, Hin * 2, Win * 2)
    elif kind == 2: oshape = (B, Cout, (Win - 2) // 4 + 1)
    else: oshape = (B, Cout, Win * 4)
    k = (4, 4) if nd2 else (4,)
    wshape = ((Cout, Cin) if kind in (0, 2) else (Cin, Cout)) + k
    x = torch.randn(ishape, device=dev); w = torch.randn(wshape, device=dev) * 0.1
    b = torch.zeros(Cout, device=dev); y = torch.empty(oshape, device=dev)
    dz = torch.randn(oshape, device=dev); dx = torch.empty(ishape, device=dev)
    dw = torch.empty_like(w); db = torch.empty_like(b)
    nws = lib.lshm_conv_workspace_floats(kind, B, Cin, Cout, Hin, Win)
    ws = torch.empty(nws, device=dev)
    P = L.ptr
    st = L.stream()
    f = timeit(lambda: L.check(lib.lshm_conv_fwd(kind, P(x), P(w), P(b), P(y), B, Cin, Cout, Hin, Win, 0, 0, 1, P(ws), nws, st)))
    d = timeit(lambda: L.check(lib.lshm_conv_dgrad(kind, P(dz), P(w), P(dx), P(x), B, Cin, Cout, Hin, Win, 0, 0, P(ws), nws, st)))
    g = timeit(lambda: L.check(lib.lshm_conv_wgrad(kind, P(x), P(dz), P(dw), P(db), B, Cin, Cout, Hin, Win, 0, 0, P(ws), nws, 0, st)))
    nb = 4 * (x.numel() + y.numel())
    fl = 2.0 * y.numel() * Cin * (16 if nd2 else 4) if kind in (0, 2) else 2.0 * x.numel() * Cout * (16 if nd2 else 4)
    print(f"kind{kind} {Cin:3d}->{Cout:3d} in {Hin}x{Win:<6d} fwd {f:7.1f}us ({nb/f/1e6:6.2f} TB/s, {fl/f/1e6:6.1f} TF)  dgrad {d:7.1f}us  wgrad {g:7.1f}us")


# optional filter: layer_bench.py B [kind [index]]
only_kind = int(sys.argv[2]) if len(sys.argv) > 2 else None
only_idx = int(sys.argv[3]) if len(sys.argv) > 3 else None
for kind in range(4):
    for i in range(6):
        if (only_kind is not None and kind != only_kind) or (only_idx is not None and i != only_idx):
            continue
        if kind == 0: layer(0, CH[i], CH[i + 1], 128 >> i, 128 >> i)
        elif kind == 1: layer(1, CH[6 - i], CH[5 - i], 2 << i, 2 << i)
        elif kind == 2: layer(2, CH[i], CH[i + 1], 1, 16384 >> (2 * i))
        else: layer(3, CH[6 - i], CH[5 - i], 1, 4 << (2 * i))
