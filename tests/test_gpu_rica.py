"""GPU: dictionary learning (src/rica_lofar.py:59-95) through the C ABI against the golden vectors of the
reference's own optimiser and against the CPU oracle on the same inputs."""
import numpy as np
import pytest
import torch

from oracle import lshm_oracle as O
from tests.util import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_closure_vs_golden_and_oracle():
    from lshm_amd.rica_lofar import RicaDictionary
    g = load_golden("rica")
    L, M, B = int(g["L"]), int(g["M"]), int(g["B"])
    X, A, S0 = O.rica_closed_form(L, M, B)
    rd = RicaDictionary(L, M, lambda1=0.1, eta=0.1, device=DEV, A=A)
    rd.set_minibatch(X.t().contiguous())
    St = S0.t().contiguous().to(DEV).requires_grad_(True)
    loss = float(rd.loss(St, True))
    assert abs(loss - float(g["it0/loss0"])) <= 1e-5 * float(g["it0/loss0"])          # fp32 GEMM + fp64 sums
    np.testing.assert_allclose(St.grad.t().cpu().numpy(), g["it0/grad0"], rtol=2e-4, atol=1e-6)
    with torch.no_grad():
        assert float(rd.loss(St, False)) == loss                                       # gradient-free variant
    # odd sizes (nothing a multiple of the tiles) against the oracle
    L2, M2, B2 = 4 * 9 * 9, 19, 7
    X2, A2, S2 = O.rica_closed_form(L2, M2, B2)
    rd2 = RicaDictionary(L2, M2, lambda1=0.3, eta=0.05, device=DEV, A=A2)
    rd2.set_minibatch(X2.t().contiguous())
    St2 = S2.t().contiguous().to(DEV).requires_grad_(True)
    want_l, want_g = O.rica_loss_grad(X2.double(), A2.double(), S2.double(), 0.3)
    got = float(rd2.loss(St2, True))
    assert abs(got - float(want_l)) <= 1e-5 * float(want_l)
    np.testing.assert_allclose(St2.grad.t().cpu().numpy(), want_g.float().numpy(), rtol=2e-4, atol=1e-6)
    A_new, dn = O.rica_dictionary_update(X2.double(), A2.double(), S2.double(), 0.05)
    got_dn = rd2.update_dictionary(S2)
    assert abs(got_dn - dn) <= 1e-5 * dn
    np.testing.assert_allclose(rd2.A.cpu().numpy(), A_new.float().numpy(), rtol=1e-5, atol=1e-6)


def test_two_minibatch_iterations_vs_reference_trajectory():
    """codes by LBFGSNew.step, then the dictionary update, twice (the dictionary carries over)."""
    from lshm_amd.rica_lofar import RicaDictionary
    g = load_golden("rica")
    L, M, B = int(g["L"]), int(g["M"]), int(g["B"])
    X, A, S0 = O.rica_closed_form(L, M, B)
    rd = RicaDictionary(L, M, lambda1=0.1, eta=0.1, device=DEV, A=A)
    x = X.t().contiguous().view(B, 4, 16, 16)
    for it in range(2):
        S, loss, dn = rd.iteration(x, S0 if it == 0 else S0.flip(1))
        # the line search takes the same branches; fp32 GEMM rounding moves the iterates in the 4th digit
        assert int(rd._opt.state[rd._opt._params[0]]["func_evals"]) == int(g[f"it{it}/func_evals"])
        np.testing.assert_allclose(S.cpu().numpy(), g[f"it{it}/S"], rtol=1e-2, atol=5e-4)
        assert abs(loss - float(g[f"it{it}/loss_after"])) <= 5e-3 * float(g[f"it{it}/loss_after"])
        assert abs(dn - float(g[f"it{it}/dA_norm"])) <= 5e-3 * float(g[f"it{it}/dA_norm"])
        np.testing.assert_allclose(rd.A.reshape(-1)[:64].cpu().numpy(), g[f"it{it}/A_head"], rtol=5e-3, atol=2e-4)
    assert rd.atoms(4, 16).shape == (M, 4, 16, 16)


def test_full_size_properties():
    """L = 4*128*128, M = 256 (:36-40): the loss is a quadratic in S, so along any direction
    f(S + tD) - f(S) - t <g, D> = t^2 ||A D||^2 / (B L) exactly (1-norm term aside, lambda1 = 0)."""
    from lshm_amd.rica_lofar import RicaDictionary
    L, M, B = 4 * 128 * 128, 256, 64
    gen = torch.Generator().manual_seed(3)
    A = torch.rand(L, M, generator=gen)
    x = torch.randn(B, L, generator=gen)
    rd = RicaDictionary(L, M, lambda1=0.0, eta=0.1, device=DEV, A=A)
    rd.set_minibatch(x)
    St = torch.rand(B, M, generator=gen).mul_(0.01).to(DEV).requires_grad_(True)
    D = torch.randn(B, M, generator=gen).mul_(0.01).to(DEV)
    f0 = float(rd.loss(St, True))
    gd = float((St.grad.double() * D.double()).sum())
    t = 0.5
    with torch.no_grad():
        f1 = float(rd.loss((St + t * D).contiguous(), False))
    quad = float(((D.double() @ rd.A.double().t()) ** 2).sum()) / (B * L)
    assert abs((f1 - f0 - t * gd) - t * t * quad) <= 1e-4 * abs(t * t * quad) + 1e-6 * abs(f0)
    # dictionary update: the residual energy cannot increase for a small enough step along E S^T
    before = f0
    rd.eta = 1e-3
    rd.update_dictionary(St.detach().t())
    with torch.no_grad():
        after = float(rd.loss(St, False))
    assert after < before


def test_closure_with_bf16_matrix_operands():
    """RicaDictionary(matrix_precision="bf16") runs the dictionary GEMMs on bf16 operands: loss and code gradient within bf16
    accuracy of the fp32 oracle, the 1-norm sub-gradient untouched."""
    from lshm_amd import _lib
    from lshm_amd.rica_lofar import RicaDictionary
    L, M, B = 4 * 16 * 16, 32, 24
    X, A, S0 = O.rica_closed_form(L, M, B)
    want_l, want_g = O.rica_loss_grad(X.double(), A.double(), S0.double(), 0.1)
    rd = RicaDictionary(L, M, lambda1=0.1, eta=0.1, device=DEV, A=A, matrix_precision="bf16")
    rd.set_minibatch(X.t().contiguous())
    St = S0.t().contiguous().to(DEV).requires_grad_(True)
    got = float(rd.loss(St, True))
    assert abs(got - float(want_l)) <= 2e-2 * float(want_l)
    g = St.grad.t().cpu().double()
    assert 1e-6 < float((g - want_g).norm() / want_g.norm()) < 2e-2
