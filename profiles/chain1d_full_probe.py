#!/usr/bin/env python3
"""lshm_chain1d_full_fwd (conv2 .. tconv3 of a 1-D autoencoder as one launch) alone on the stream, B = 256 (one problem:
the step runs netT and netF as a pair), with the stage boundaries of workgroup 0.  Usage: python profiles/chain1d_full_probe.py"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lshm_amd import _lib as L
lib = L.load()
B, dev = 256, "cuda"
g = torch.Generator().manual_seed(0)
ch = [12, 24, 48, 96, 192]
w = [torch.randn(ch[i + 1], ch[i], 4, generator=g) * (3.0 / (4 * ch[i])) ** 0.5 for i in range(4)]
w += [torch.randn(16, 784, generator=g) * 0.05, torch.randn(16, 16, generator=g) * 0.3, torch.randn(16, 16, generator=g) * 0.3,
      torch.randn(768, 32, generator=g) * 0.2]
w += [torch.randn(ch[4 - i], ch[3 - i], 4, generator=g) * (3.0 / ch[4 - i]) ** 0.5 for i in range(4)]
bs = [torch.randn(n, generator=g) * 0.1 for n in (24, 48, 96, 192, 16, 16, 16, 768, 96, 48, 24, 12)]
shapes = [(B, 24, 256), (B, 48, 64), (B, 96, 16), (B, 784), (B, 16), (B, 256), (B, 32), (B, 768), (B, 96, 16), (B, 48, 64), (B, 24, 256), (B, 12, 1024)]
x1 = torch.nn.functional.elu(torch.randn(B, 12, 1024, generator=g)).to(dev)
wd, bd = [t.to(dev) for t in w], [t.to(dev) for t in bs]
outs = [torch.zeros(s, device=dev) for s in shapes]
arr = lambda ts: (C.c_void_p * 12)(*[t.data_ptr() for t in ts])
st = L.stream()
def run(stamps=None):
    L.check(lib.lshm_chain1d_full_fwd(L.ptr(x1), arr(wd), arr(bd), arr(outs), 256, B, L.ptr(stamps) if stamps is not None else None, st))
for _ in range(5): run()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); a.record()
for _ in range(30): run()
b.record(); torch.cuda.synchronize()
print(f"chain1d_full_fwd, one problem, B = {B}: {a.elapsed_time(b) / 30 * 1e3:.1f} us")
stamps = torch.zeros(64, dtype=torch.int64, device=dev)
run(stamps); torch.cuda.synchronize()
sv = stamps.cpu().tolist()
n = max(i for i, v in enumerate(sv) if v) + 1
print("stage boundaries (shader-clock ticks between consecutive barriers):", [sv[i + 1] - sv[i] for i in range(n - 1)], "total", sv[n - 1] - sv[0])
