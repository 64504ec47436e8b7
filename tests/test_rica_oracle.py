"""CPU: the oracle's restatement of the dictionary-learning loop body (src/rica_lofar.py:59-95) against the
golden vectors produced with the reference's own optimiser (tests/golden/make_golden_rica.py)."""
import numpy as np
import torch

from lshm_amd.lbfgsnew import LBFGSNew
from oracle import lshm_oracle as O
from tests.util import load_golden


def _problem(g):
    return O.rica_closed_form(int(g["L"]), int(g["M"]), int(g["B"]))


def test_closure_loss_and_gradient_match_reference_autograd():
    g = load_golden("rica")
    X, A, S0 = _problem(g)
    loss, grad = O.rica_loss_grad(X, A, S0, 0.1)
    assert abs(float(loss) - float(g["it0/loss0"])) <= 1e-5 * float(g["it0/loss0"])
    np.testing.assert_allclose(grad.numpy(), g["it0/grad0"], rtol=2e-4, atol=1e-6)
    # the 1-norm of a matrix is the largest column sum, not the entry-wise sum (the term is tiny but present)
    col = int(torch.argmax(S0.abs().sum(0)))
    other = (col + 1) % S0.shape[1]
    d = grad - (-2.0 * (A.t() @ (X - A @ S0)) / X.numel())
    assert d[:, col].abs().max() > 0 and d[:, other].abs().max() == 0


def test_loop_body_with_the_ported_optimiser_matches_reference_trajectory():
    """LBFGSNew port + oracle closure, two minibatch iterations: codes, loss, ||dA|| and dictionary."""
    g = load_golden("rica")
    X, A, S0 = _problem(g)
    for it in range(2):
        S = (S0 if it == 0 else S0.flip(1)).clone().requires_grad_(True)
        opt = LBFGSNew([S], history_size=7, max_iter=10, line_search_fn=True, batch_mode=True)

        def closure():
            if torch.is_grad_enabled():
                loss, grad = O.rica_loss_grad(X, A, S.detach(), 0.1)
                S.grad = grad
                return loss
            return O.rica_loss(X, A, S.detach(), 0.1)

        opt.step(closure)
        assert int(opt.state[opt._params[0]]["func_evals"]) == int(g[f"it{it}/func_evals"])
        np.testing.assert_allclose(S.detach().numpy(), g[f"it{it}/S"], rtol=5e-3, atol=2e-4)
        la = float(O.rica_loss(X, A, S.detach(), 0.1))
        assert abs(la - float(g[f"it{it}/loss_after"])) <= 2e-3 * float(g[f"it{it}/loss_after"])
        A, dn = O.rica_dictionary_update(X, A, S.detach(), 0.1)
        assert abs(dn - float(g[f"it{it}/dA_norm"])) <= 2e-3 * float(g[f"it{it}/dA_norm"])
        np.testing.assert_allclose(A.reshape(-1)[:64].numpy(), g[f"it{it}/A_head"], rtol=2e-3, atol=1e-4)
        assert abs(float(A.double().sum()) - float(g[f"it{it}/A_sum"])) <= 1e-4 * abs(float(g[f"it{it}/A_sum"]))
