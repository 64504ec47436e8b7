"""Small deterministic regression problem shared by the LBFGSNew golden generator (run with the
reference optimiser) and the parity test (run with lshm_amd.lbfgsnew)."""
import torch

from oracle import lshm_oracle as O


def make_problem(dtype=torch.float64, device="cpu"):
    W1 = (0.5 * O.closed_form((8, 5), "lb:W1", 1.0, 0.91)).to(dtype=dtype, device=device).requires_grad_(True)
    b1 = (0.1 * O.closed_form((8,), "lb:b1", 1.0, 0.53)).to(dtype=dtype, device=device).requires_grad_(True)
    W2 = (0.5 * O.closed_form((1, 8), "lb:W2", 1.0, 0.37)).to(dtype=dtype, device=device).requires_grad_(True)
    params = [W1, b1, W2]
    X = O.closed_form((96, 5), "lb:X", 1.0, 0.61).to(dtype=dtype, device=device)
    Y = torch.sin(X.sum(1, keepdim=True)) + 0.3 * X[:, :1] ** 2

    def closure_for(opt, batch):
        xb, yb = X[32 * (batch % 3):32 * (batch % 3 + 1)], Y[32 * (batch % 3):32 * (batch % 3 + 1)]

        def closure():
            if torch.is_grad_enabled():
                opt.zero_grad()
            h = torch.tanh(xb @ W1.t() + b1)
            loss = ((h @ W2.t() - yb) ** 2).mean()
            if loss.requires_grad:
                loss.backward()
            return loss
        return closure
    return params, closure_for
