"""LBFGSNew port vs trajectories recorded from the reference optimiser (tests/golden/lbfgs.npz):
same losses, same accepted step sizes, same parameters, same number of closure evaluations, in
batch (backtracking), full-batch (cubic strong-Wolfe) and fixed-step modes.  Host logic: CPU."""
import numpy as np
import pytest
import torch

from lshm_amd.lbfgsnew import LBFGSNew
from tests.lbfgs_problem import make_problem
from tests.util import load_golden

MODES = {"batch": dict(history_size=7, max_iter=4, line_search_fn=True, batch_mode=True),
         "full": dict(history_size=7, max_iter=6, line_search_fn=True, batch_mode=False),
         "fixed": dict(history_size=5, max_iter=3, line_search_fn=False, batch_mode=False, lr=0.05)}


def _run(mode, device="cpu", dtype=torch.float64):
    params, closure_for = make_problem(dtype, device)
    opt = LBFGSNew(params, **MODES[mode])
    losses, ts = [], []
    for it in range(4):
        l0 = opt.step(closure_for(opt, it if mode == "batch" else 0))
        losses.append(float(l0))
        ts.append(float(opt.state[opt._params[0]]["t"]))
    final = torch.cat([p.detach().reshape(-1) for p in params]).cpu().numpy()
    return losses, ts, final, opt.state[opt._params[0]]["func_evals"]


@pytest.mark.parametrize("mode", list(MODES))
def test_trajectory_matches_reference(mode):
    g = load_golden("lbfgs")
    losses, ts, final, evals = _run(mode)
    np.testing.assert_allclose(losses, g[f"{mode}/loss"], rtol=1e-9)
    np.testing.assert_allclose(ts, g[f"{mode}/t"], rtol=1e-9)
    np.testing.assert_allclose(final, g[f"{mode}/final"], rtol=1e-8, atol=1e-12)
    assert evals == int(g[f"{mode}/func_evals"])


def test_single_flat_parameter_equals_parameter_list():
    """The flat-arena fast path (one contiguous parameter) takes the same trajectory."""
    params, closure_for = make_problem()
    flat = torch.cat([p.detach().reshape(-1) for p in params]).clone().requires_grad_(True)
    shapes = [p.shape for p in params]
    X, Y = None, None

    def views():
        out, off = [], 0
        for s in shapes:
            n = int(np.prod(s))
            out.append(flat[off:off + n].view(s))
            off += n
        return out
    opt = LBFGSNew([flat], **MODES["batch"])
    from oracle import lshm_oracle as O
    Xf = O.closed_form((96, 5), "lb:X", 1.0, 0.61).double()
    Yf = torch.sin(Xf.sum(1, keepdim=True)) + 0.3 * Xf[:, :1] ** 2
    losses = []
    for it in range(4):
        xb, yb = Xf[32 * (it % 3):32 * (it % 3 + 1)], Yf[32 * (it % 3):32 * (it % 3 + 1)]

        def closure():
            if torch.is_grad_enabled():
                opt.zero_grad()
            W1, b1, W2 = views()
            loss = ((torch.tanh(xb @ W1.t() + b1) @ W2.t() - yb) ** 2).mean()
            if loss.requires_grad:
                loss.backward()
            return loss
        losses.append(float(opt.step(closure)))
    g = load_golden("lbfgs")
    np.testing.assert_allclose(losses, g["batch/loss"], rtol=1e-9)


@pytest.mark.gpu
def test_device_vector_algebra_same_trajectory():
    """On the GPU the inner products / axpys go through lshm_dot_flat etc. (fp32 problem)."""
    l_cpu, t_cpu, f_cpu, _ = _run("batch", "cpu", torch.float32)
    params, closure_for = make_problem(torch.float32, "cuda")
    flat = torch.cat([p.detach().reshape(-1) for p in params]).clone().requires_grad_(True)
    shapes = [p.shape for p in params]
    from oracle import lshm_oracle as O
    Xf = O.closed_form((96, 5), "lb:X", 1.0, 0.61).cuda()
    Yf = torch.sin(Xf.sum(1, keepdim=True)) + 0.3 * Xf[:, :1] ** 2
    opt = LBFGSNew([flat], **MODES["batch"])
    losses = []
    for it in range(4):
        xb, yb = Xf[32 * (it % 3):32 * (it % 3 + 1)], Yf[32 * (it % 3):32 * (it % 3 + 1)]

        def closure():
            if torch.is_grad_enabled():
                opt.zero_grad()
            off, vs = 0, []
            for s in shapes:
                n = int(np.prod(s))
                vs.append(flat[off:off + n].view(s))
                off += n
            loss = ((torch.tanh(xb @ vs[0].t() + vs[1]) @ vs[2].t() - yb) ** 2).mean()
            if loss.requires_grad:
                loss.backward()
            return loss
        losses.append(float(opt.step(closure)))
    np.testing.assert_allclose(losses, l_cpu, rtol=2e-3)
