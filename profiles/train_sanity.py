import sys, torch
sys.path.insert(0, ".")
from lshm_amd import KHarmonicTrainer, TrainConfig
dev = torch.device("cuda:0")
B = 72  # 8 baselines x 9 patches
tr = KHarmonicTrainer(TrainConfig(Kc=10), batch=B, batch_per_bline=9, default_batch=8, device=dev)
tr.init_parameters(seed=0)
g = torch.Generator().manual_seed(7)
for mb in range(3):                      # three minibatches x 40 ADMM iterations, multipliers reset per minibatch
    x = torch.randn(B, 4, 128, 128, generator=g); x = (x - x.mean()) / x.std()
    uv = 1000.0 * torch.randn(B, 2, generator=g)
    tr.new_minibatch(x.to(dev), uv.to(dev))
    for it in range(40):
        tr.step()
        if it % 13 == 0 or it == 39:
            t = tr.read_terms()
            assert all(v == v and abs(v) < 1e6 for v in t.values()), t
            print(mb, it, {k: round(v, 5) for k, v in t.items()})
print("finite:", bool(torch.isfinite(tr.params).all()), "param norm", float(tr.params.norm()))
