#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs /root/reference): time one ADMM iteration of the ACTUAL reference modules
(src/lofar_models.py imported unmodified, closure of src/kharmonic_lofar.py:132-202 restated on top of them,
torch.optim.Adam over all four parameter groups) and of the oracle port (oracle/lshm_oracle.py, what
bench.py's cpu_baseline runs on the GPU box) on the same cores, and print their ratio.  The ratio is what makes
the on-box "port" baseline interpretable against the reference.

Usage: python profiles/reference_cpu_timing.py [threads=8] [B=256] [K=10] [bpb=8] [steps=4]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.dont_write_bytecode = True
import make_golden as G  # noqa: E402  (import_reference / reference_augmented_loss)
from oracle import lshm_oracle as O  # noqa: E402

threads, B, K, bpb, steps = (int(v) for v in (sys.argv[1:] + ["8", "256", "10", "8", "4"][len(sys.argv) - 1:])[:5])
torch.set_num_threads(threads)
lm, _ = G.import_reference()
ref_aug = G.reference_augmented_loss()
cfg = O.StepConfig(K=K, bpb=bpb, batch_size=B // bpb)
scales = torch.tensor(O.DEFAULT_SCALES)
torch.manual_seed(0)
net = lm.AutoEncoderCNN2(latent_dim=cfg.L, channels=4, harmonic_scales=scales, rica=True)
netT = lm.AutoEncoder1DCNN(latent_dim=cfg.Lt, channels=4, harmonic_scales=scales, rica=True)
netF = lm.AutoEncoder1DCNN(latent_dim=cfg.Lt, channels=4, harmonic_scales=scales, rica=True)
mod = lm.Kmeans(latent_dim=cfg.L + 2 * cfg.Lt, K=cfg.K, p=cfg.p)
x = torch.randn(B, 4, 128, 128)
uv = 1000.0 * torch.randn(B, 2)
n = x.numel()
plist = list(net.parameters()) + list(netT.parameters()) + list(netF.parameters()) + list(mod.parameters())
opt = torch.optim.Adam(plist, lr=cfg.lr)
ys = [torch.zeros(n) for _ in range(3)]
sse = torch.nn.MSELoss(reduction="sum")


def three(xin):
    a1, z = net(xin, uv)
    h = (xin - a1) / 2
    a2, zT = netT(torch.flatten(h, 2, 3), uv)
    a2 = a2.view_as(h)
    a3, zF = netF(torch.flatten(torch.transpose(h, 2, 3), 2, 3), uv)
    a3 = torch.transpose(a3.view_as(h), 2, 3)
    return a1, h, a2, a3, z, zT, zF


def reference_iteration():
    global ys

    def closure():
        opt.zero_grad()
        a1, h, a2, a3, z, zT, zF = three(x)
        t0 = sse(a1 + a2 + a3, x) / n
        t1 = (torch.dot(ys[0], (x - a1).view(-1)) + cfg.rho / 2 * sse(x, a1)) / n
        t2 = (torch.dot(ys[1], (h - a2).view(-1)) + cfg.rho / 2 * sse(h, a2)) / n
        t3 = (torch.dot(ys[2], (h - a3).reshape(-1)) + cfg.rho / 2 * sse(h, a3)) / n
        Z = torch.cat((z, zT, zF), 1)
        lc = lambda t: torch.sum(torch.log(torch.cosh(t))) / t.numel()
        total = (t0 + t1 + t2 + t3 + cfg.alpha * mod.clustering_error(Z) + cfg.beta * mod.cluster_similarity()
                 + cfg.gamma * ref_aug(Z, cfg.bpb, cfg.batch_size) + cfg.rica_lambda * (lc(z) + lc(zT) + lc(zF)))
        total.backward(retain_graph=True)
        [v.item() for v in (t0, t1, t2, t3)]  # the script prints its terms every closure (:176-181)
        return total
    opt.step(closure)
    with torch.no_grad():
        a1, h, a2, a3, *_ = three(x)
        ys = [ys[0] + cfg.rho * (x - a1).view(-1), ys[1] + cfg.rho * (h - a2).view(-1),
              ys[2] + cfg.rho * (h - a3).reshape(-1)]


def median_time(fn):
    ts = []
    for _ in range(steps + 1):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts[1:])
    return ts[len(ts) // 2]


t_ref = median_time(reference_iteration)
params, M = O.make_params(cfg)
yo = [torch.zeros(n) for _ in range(3)]
adam = O.AdamState(O.flat_leaves(params, M), cfg.lr)


def port_iteration():
    global yo
    _, yo, _ = O.admm_iteration(params, M, x, uv, yo, cfg, adam, khm_fn=O.khm_loss_loop,
                                sim_fn=O.cluster_similarity_loop, aug_fn=O.augmented_loss_loop)


t_port = median_time(port_iteration)
print(f"threads={threads} B={B} K={K} bpb={bpb}: reference {t_ref:.3f} s/step = {B / t_ref:.1f} patches/s; "
      f"port {t_port:.3f} s/step = {B / t_port:.1f} patches/s; port/reference time ratio {t_port / t_ref:.3f}")
