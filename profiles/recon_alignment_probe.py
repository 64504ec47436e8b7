#!/usr/bin/env python3
"""Probe: does the relative placement of the ten image-sized streams of recon_kernel matter?
Ten arrays carved from one allocation at a pitch of exactly 64 MiB versus the same with a skew added
to each array's offset."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshm_amd import _lib as L
lib = L.load(); dev = torch.device("cuda"); B = 256
n = B * 4 * 128 * 128
def timeit(fn, iters=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
sums = torch.zeros(8, device=dev, dtype=torch.float64)
nws = lib.lshm_recon_workspace_floats(B * 4, 128); ws = torch.empty(nws, device=dev)
for skew in (0, 64, 1024, 4096 + 64, 65536 + 1024, 1 << 20):   # floats
    big = torch.randn(10 * (n + skew) + 16, device=dev)
    img = [big[i * (n + skew): i * (n + skew) + n] for i in range(10)]
    def recon():
        L.check(lib.lshm_recon_losses_fwd_bwd(*[L.ptr(t) for t in img[:7]], 1.0, B * 4, 128, L.ptr(sums), L.ptr(img[7]),
                                              L.ptr(img[8]), L.ptr(img[9]), L.ptr(ws), L.stream()))
    def mult():
        L.check(lib.lshm_multiplier_update(*[L.ptr(t) for t in img[:7]], 1.0, B * 4, 128, L.stream()))
    print(f"skew {skew*4:9d} B: recon {timeit(recon):7.1f} us   multiplier_update {timeit(mult):7.1f} us")
    del big, img

# the same kernel right after other work has swept ~1.3 GB through the caches (as inside a step)
big = torch.randn(10 * n + 16, device=dev)
img = [big[i * n: i * n + n] for i in range(10)]
junk_a = torch.randn(160 * (1 << 20), device=dev); junk_b = torch.empty_like(junk_a)
small = torch.zeros(4096, device=dev)
junk_c = torch.zeros(1 << 30, device=dev); junk_d = torch.empty_like(junk_c)   # 4 GiB each
def recon():
    L.check(lib.lshm_recon_losses_fwd_bwd(*[L.ptr(t) for t in img[:7]], 1.0, B * 4, 128, L.ptr(sums), L.ptr(img[7]),
                                          L.ptr(img[8]), L.ptr(img[9]), L.ptr(ws), L.stream()))
for mode in ("back-to-back", "after a 1.3 GB copy", "after an 8 GiB copy", "after 80 tiny kernels", "after 2 ms idle", "after rewriting x1 x2 x3c (as the forward does)"):
    tot = 0.0
    for it in range(12):
        if mode.startswith("after a 1.3"): junk_b.copy_(junk_a)
        if mode.startswith("after an 8"): junk_d.copy_(junk_c)
        if mode.startswith("after 80"):
            for _ in range(80): small.add_(1.0)
        if mode.startswith("after 2 ms"):
            torch.cuda.synchronize(); import time; time.sleep(0.002)
        if mode.startswith("after rew"):
            for k in (1, 2, 3): img[k].mul_(1.0001)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); recon(); b.record(); torch.cuda.synchronize()
        if it >= 2: tot += a.elapsed_time(b)
    print(f"recon {mode}: {tot / 10 * 1e3:7.1f} us")
