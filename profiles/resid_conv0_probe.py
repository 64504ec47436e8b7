"""conv0 of netT / netF from x and x1: one tile kernel (lshm_resid_conv0 / _keep) against lshm_residual_split + lshm_conv_fwd_pair,
each alone on an idle GPU at B = 256 (us per launch, algorithmic MB and TB/s)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshm_amd import _lib as L

lib = L.load()
dev = torch.device("cuda:0")
nb = 256
g = torch.Generator().manual_seed(1)
x, x1 = torch.randn(nb, 4, 128, 128, generator=g).to(dev), torch.randn(nb, 4, 128, 128, generator=g).to(dev)
w = [(torch.randn(8, 4, 4, generator=g) * 0.3).to(dev) for _ in range(2)]
b = [(torch.randn(8, generator=g) * 0.1).to(dev) for _ in range(2)]
y = [torch.empty(nb, 8, 4096, device=dev) for _ in range(2)]
row, col = torch.empty(nb, 4, 16384, device=dev), torch.empty(nb, 4, 16384, device=dev)
st, P = L.stream(), L.ptr
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def timed(fn, reps=20):
    ts = []
    for _ in range(reps):
        flush.max()   # evict by READING 512 MB: clean lines only (a fill would leave 256 MB of dirty lines to write back under the timed kernel)
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def fused(): L.check(lib.lshm_resid_conv0(P(x), P(x1), P(w[0]), P(b[0]), P(y[0]), P(w[1]), P(b[1]), P(y[1]), nb, st), "a")
def keep(): L.check(lib.lshm_resid_conv0_keep(P(x), P(x1), P(w[0]), P(b[0]), P(y[0]), P(w[1]), P(b[1]), P(y[1]), P(row), P(col), nb, st), "b")
def split():
    L.check(lib.lshm_residual_split(P(x), P(x1), P(row), P(col), nb * 4, 128, st), "c")
    L.check(lib.lshm_conv_fwd_pair(2, P(row), P(w[0]), P(b[0]), P(y[0]), P(col), P(w[1]), P(b[1]), P(y[1]), nb, 4, 8, 1, 16384, 0, 0, 1, None, 0, st), "d")


img = nb * 4 * 128 * 128 * 4 / 1e6
for name, fn, mb in (("resid_conv0 (nothing kept)", fused, 2 * img + img), ("resid_conv0_keep", keep, 2 * img + img + 2 * img),
                     ("residual_split + conv_fwd_pair", split, 2 * img + 2 * img + 2 * img + img)):
    for _ in range(3): fn()
    med, best = timed(fn)
    print(f"{name:34s} median {med:7.1f} us  best {best:7.1f} us   {mb:6.0f} MB algorithmic  {mb / med:5.2f} TB/s", flush=True)
