import sys, os, torch
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
def conv2(Cin, Cout, H):
    x = torch.randn(B, Cin, H, H, device=dev); w = torch.randn(Cout, Cin, 4, 4, device=dev) * 0.1
    b = torch.zeros(Cout, device=dev); y = torch.empty(B, Cout, H // 2, H // 2, device=dev)
    nws = lib.lshm_conv_workspace_floats(0, B, Cin, Cout, H, H); ws = torch.empty(max(nws, 1), device=dev)
    P = L.ptr; st = L.stream()
    return timeit(lambda: L.check(lib.lshm_conv_fwd(0, P(x), P(w), P(b), P(y), B, Cin, Cout, H, H, 0, 0, 1, P(ws), nws, st)))
print("abl", os.environ.get("LSHM_ABLATE", "0"), "conv2 %.1f us  conv3 %.1f us" % (conv2(12, 24, 32), conv2(24, 48, 16)))
