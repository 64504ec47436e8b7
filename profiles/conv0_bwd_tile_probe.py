"""Backward of conv0 of netT / netF + the combination into gx1: one tile kernel (lshm_conv0_bwd_tile) against
lshm_conv_bwd_fused x 2 + lshm_combine_dx1, each alone on an idle GPU at B = 256 (us, algorithmic MB, TB/s)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshm_amd import _lib as L

lib = L.load()
dev = torch.device("cuda:0")
nb = 256
g = torch.Generator().manual_seed(1)
r = torch.randn(nb, 4, 128, 128, generator=g).to(dev)
rcol = r.transpose(2, 3).contiguous()
dz = [torch.randn(nb, 8, 4096, generator=g).to(dev) for _ in range(2)]
w = [(torch.randn(8, 4, 4, generator=g) * 0.3).to(dev) for _ in range(2)]
gp = torch.randn(nb, 4, 128, 128, generator=g).to(dev)
gx1 = torch.empty_like(gp)
dw = [torch.empty(8, 4, 4, device=dev) for _ in range(2)]
db = [torch.empty(8, device=dev) for _ in range(2)]
dx = [torch.empty(nb, 4, 16384, device=dev) for _ in range(2)]
st, P = L.stream(), L.ptr
nws = lib.lshm_conv0_bwd_tile_workspace_floats(); ws = torch.empty(nws, device=dev)
nws2 = lib.lshm_conv_workspace_floats(2, nb, 4, 8, 1, 16384); ws2 = torch.empty(nws2, device=dev)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def timed(fn, reps=20):
    ts = []
    for _ in range(reps):
        flush.max()  # evict by reading: clean lines only
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def tile():
    L.check(lib.lshm_conv0_bwd_tile(P(r), P(dz[0]), P(dz[1]), P(w[0]), P(w[1]), P(gp), P(gx1), P(dw[0]), P(db[0]), P(dw[1]), P(db[1]), nb, P(ws), nws, 0, 0, st), "t")
def old():
    for k, xin in enumerate((r, rcol)):
        L.check(lib.lshm_conv_bwd_fused(2, P(xin), P(dz[k]), P(w[k]), P(dw[k]), P(db[k]), P(dx[k]), 0, nb, 4, 8, 1, 16384, P(ws2), nws2, st), "o")
    L.check(lib.lshm_combine_dx1(P(gp), P(dx[0]), P(dx[1]), P(gx1), nb * 4, 128, st), "c")


img = nb * 4 * 128 * 128 * 4 / 1e6
for name, fn, mb in (("conv0_bwd_tile (+ closing sums)", tile, 5 * img), ("conv_bwd_fused x 2 + combine_dx1 (each network its own launch here)", old, 9 * img)):
    for _ in range(3): fn()
    med, best = timed(fn)
    print(f"{name:70s} median {med:7.1f} us  best {best:7.1f} us   {mb:6.0f} MB algorithmic  {mb / med:5.2f} TB/s", flush=True)
