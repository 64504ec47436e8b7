import sys, torch
sys.path.insert(0, ".")
from lshm_amd import _lib as L
lib = L.load(); dev = torch.device("cuda"); B = 256
def timeit(fn, iters=50):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for (Cin, Cout, H) in ((24, 48, 16), (48, 96, 8), (96, 192, 4), (12, 24, 32)):
    x = torch.randn(B, Cin, H, H, device=dev); w = torch.randn(Cout, Cin, 4, 4, device=dev) * 0.1
    b = torch.zeros(Cout, device=dev); y = torch.empty(B, Cout, H // 2, H // 2, device=dev)
    nws = lib.lshm_conv_workspace_floats(0, B, Cin, Cout, H, H); ws = torch.empty(nws, device=dev)
    P = L.ptr; st = L.stream()
    t1 = timeit(lambda: L.check(lib.lshm_conv_fwd(0, P(x), P(w), P(b), P(y), B, Cin, Cout, H, H, 0, 0, 1, P(ws), nws, st)))
    t0 = timeit(lambda: L.check(lib.lshm_conv_fwd(0, P(x), P(w), P(b), P(y), B, Cin, Cout, H, H, 0, 0, 1, None, 0, st)))
    # empty-ish kernel for launch floor
    e = timeit(lambda: L.check(lib.lshm_scale_flat(P(b), 1.0, 8, st)))
    print(f"conv {Cin}->{Cout} {H}x{H}: split {t1:.1f}us  nosplit {t0:.1f}us  (tiny kernel launch {e:.1f}us)")
