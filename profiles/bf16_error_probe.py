import sys, torch
sys.path.insert(0, '.')
from oracle import lshm_oracle as O
from lshm_amd import KHarmonicTrainer, TrainConfig
from tests.util import rel_err
DEV='cuda'
B, K, bpb, bs = 4, 5, 2, 2
ocfg = O.StepConfig(K=K, bpb=bpb, batch_size=bs)
params, M = O.make_params(ocfg)
x, uv = O.closed_form_inputs(B, 4)
y = [torch.zeros(x.numel()) for _ in range(3)]
leaves = O.flat_leaves(params, M)
for l in leaves: l.requires_grad_(True)
total, terms = O.closure_losses(params, M, x, uv, y, ocfg)
grads = torch.autograd.grad(total, leaves)
names = [f"{g}.{k}" for g in O.GROUPS for k in params[g]] + ["mod.M"]
for label, cfg in (("fp32", TrainConfig(Kc=K)), ("bf16 operands", TrainConfig(Kc=K, matrix_precision="bf16")), ("bf16 storage", TrainConfig(Kc=K, matrix_precision="bf16", activation_storage="bf16"))):
    tr = KHarmonicTrainer(cfg, batch=B, batch_per_bline=bpb, default_batch=bs, device=DEV)
    tr.load_state_dicts(params["net"], params["netT"], params["netF"], {"M": M})
    tr.new_minibatch(x.to(DEV), uv.to(DEV))
    tr.closure_only()
    t = tr.read_terms()
    terr = max(abs(t[n] - float(rf)) / (abs(float(rf)) + 1e-12) for n, rf in zip(("loss0","loss1","loss2","loss3","kdist","aug","sim","rica"), terms))
    num = sum(float((tr.view(n, tr.grads).cpu().double() - gr.double()).pow(2).sum()) for n, gr in zip(names, grads))
    den = sum(float(gr.double().pow(2).sum()) for gr in grads)
    errs = sorted((rel_err(tr.view(n, tr.grads), gr), n) for n, gr in zip(names, grads))
    print(label, "max term rel err %.2e" % terr, "total rel %.2e" % (abs(t["total"]-total.item())/abs(total.item())), "grad vec %.2e" % (num/den)**0.5, "median tensor %.2e" % errs[len(errs)//2][0], "worst", "%.2e %s" % errs[-1])
