#!/usr/bin/env python3
"""lshm_deep2d_fwd (conv3 .. tconv2 of the 2-D autoencoder as one launch, round 4) alone on the stream against the
launches it replaces (lshm_conv_fwd x 7 + lshm_linear_fwd x 4, each with its split-K combine where the tile table
splits), B = 256, HIP-event timing; the three variants; two launches side by side on two streams (the paired forwards).
Under `rocprofv3 --kernel-trace --stats` the same script separates the packing launch from the chain.
Usage: python profiles/deep2d_probe.py"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lshm_amd import _lib as L
from tests.test_gpu_fullsize_ops import _deep2d_problem
lib = L.load()
B, dev = int(os.environ.get("PROBE_B", 256)), "cuda"
Ld, hd = 224, 16

def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

w, b, x2, uv1, uv3 = _deep2d_problem(B, 1)
wd, bd, x2d = [t.to(dev) for t in w], [t.to(dev) for t in b], x2.to(dev)
shapes = [(B, 48, 8, 8), (B, 96, 4, 4), (B, 768 + hd), (B, Ld), (B, 256), (B, Ld + hd), (B, 768), (B, 96, 4, 4), (B, 48, 8, 8), (B, 24, 16, 16),
          (B, 12, 32, 32)]
def fresh():
    outs = [torch.zeros(s, device=dev) for s in shapes]
    outs[2][:, 768:] = uv1.to(dev); outs[5][:, Ld:] = uv3.to(dev)
    return outs
arr = lambda ts: (C.c_void_p * 11)(*[t.data_ptr() for t in ts])
outs, packed = fresh(), torch.empty(lib.lshm_deep2d_packed_floats(), device=dev)
st = L.stream()
for variant, name in ((0, "1 patch x 1024 threads"), (1, "2 patches x 1024 threads"), (2, "1 patch x 512 threads")):
    def deep():
        L.check(lib.lshm_deep2d_fwd(L.ptr(x2d), arr(wd), arr(bd), arr(outs), 256, L.ptr(packed), B, variant, None, st))
    print(f"deep2d_fwd + pack, {name}: {timeit(deep):6.1f} us")
    # stage boundaries of workgroup 0 (shader clock, 100 MHz constant-rate counter on gfx950: 10 ns per tick)
    stamps = torch.zeros(32, dtype=torch.int64, device=dev)
    L.check(lib.lshm_deep2d_fwd(L.ptr(x2d), arr(wd), arr(bd), arr(outs), 256, L.ptr(packed), B, variant, L.ptr(stamps), st))
    torch.cuda.synchronize()
    sv = stamps.cpu().tolist()
    n = max(i for i, v in enumerate(sv) if v) + 1
    names = ["zero", "load", "conv3", "reduce3", "conv4", "reduce4", "conv5", "reduce5", "fc1", "fc2in", "fc2out", "fc3", "tconv0", "reduce-t0",
             "tconv1", "reduce-t1", "tconv2", "tconv3"]
    d = [sv[i + 1] - sv[i] for i in range(n - 1)]
    print("   stages (ticks of the s_memtime counter): " + ", ".join(f"{names[i] if i < len(names) else i} {d[i]}" for i in range(len(d))) + f"; total {sv[n - 1] - sv[0]}")

layers = [(0, 24, 48, 16), (0, 48, 96, 8), (0, 96, 192, 4)]
tl = [(192, 96, 2), (96, 48, 4), (48, 24, 8), (24, 12, 16)]
ys = [torch.empty(B, co, h // 2, h // 2, device=dev) for _, _, co, h in layers]
ty = [torch.empty(B, co, 2 * h, 2 * h, device=dev) for _, co, h in tl]
cat1 = torch.zeros(B, 784, device=dev); z1 = torch.empty(B, Ld, device=dev); mu = torch.empty(B, Ld, device=dev)
cat3 = torch.zeros(B, 240, device=dev); d0 = torch.empty(B, 768, device=dev)
wsn = max([lib.lshm_conv_workspace_floats(0, B, ci, co, h, h) for _, ci, co, h in layers] +
          [lib.lshm_conv_workspace_floats(1, B, ci, co, h, h) for ci, co, h in tl] +
          [lib.lshm_linear_workspace_floats(B, 784, Ld), lib.lshm_linear_workspace_floats(B, 240, 768)])
ws = torch.empty(max(wsn, 1), device=dev)
def separate():
    cur = x2d
    for li, (kind, ci, co, h) in enumerate(layers):
        out = ys[li] if li < 2 else cat1  # conv5 writes into cat1 (row pitch 784)
        L.check(lib.lshm_conv_fwd(kind, L.ptr(cur), L.ptr(wd[li]), L.ptr(bd[li]), L.ptr(out), B, ci, co, h, h, 0, 784 if li == 2 else 0, 1,
                                  L.ptr(ws), wsn, st))
        cur = out
    L.check(lib.lshm_linear_fwd(L.ptr(cat1), 784, L.ptr(wd[3]), L.ptr(bd[3]), L.ptr(z1), Ld, B, 784, Ld, 1, L.ptr(ws), wsn, st))
    L.check(lib.lshm_linear_fwd(L.ptr(z1), Ld, L.ptr(wd[4]), L.ptr(bd[4]), L.ptr(mu), Ld, B, Ld, Ld, 1, L.ptr(ws), wsn, st))
    L.check(lib.lshm_linear_fwd(L.ptr(mu), Ld, L.ptr(wd[5]), L.ptr(bd[5]), L.ptr(cat3), 240, B, Ld, Ld, 1, L.ptr(ws), wsn, st))
    L.check(lib.lshm_linear_fwd(L.ptr(cat3), 240, L.ptr(wd[6]), L.ptr(bd[6]), L.ptr(d0), 768, B, 240, 768, 0, L.ptr(ws), wsn, st))
    cur = d0
    for li, (ci, co, h) in zip((7, 8, 9, 10), tl):
        L.check(lib.lshm_conv_fwd(1, L.ptr(cur), L.ptr(wd[li]), L.ptr(bd[li]), L.ptr(ty[li - 7]), B, ci, co, h, h, 0, 0, 1, L.ptr(ws), wsn, st))
        cur = ty[li - 7]
print(f"the eleven separate launches (+ split-K combines): {timeit(separate):6.1f} us")

# two chains side by side (the paired forwards of the default schedule): second set of buffers, second stream
outs2, packed2 = fresh(), torch.empty_like(packed)
s2 = torch.cuda.Stream()
for variant, name in ((0, "1 patch x 1024 threads"), (1, "2 patches x 1024 threads"), (2, "1 patch x 512 threads")):
    def pair():
        s2.wait_stream(torch.cuda.current_stream())  # fork first: the two launches depend on nothing of each other
        L.check(lib.lshm_deep2d_fwd(L.ptr(x2d), arr(wd), arr(bd), arr(outs), 256, L.ptr(packed), B, variant, None, st))
        with torch.cuda.stream(s2):
            L.check(lib.lshm_deep2d_fwd(L.ptr(x2d), arr(wd), arr(bd), arr(outs2), 256, L.ptr(packed2), B, variant, None, L.stream()))
        torch.cuda.current_stream().wait_stream(s2)
    print(f"two deep2d_fwd side by side, {name}: {timeit(pair):6.1f} us")
