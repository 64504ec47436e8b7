#!/usr/bin/env python3
"""Golden vectors for the dictionary-learning loop body of upstream ``src/rica_lofar.py:59-95``.

Runs only in the build container.  The upstream script cannot be imported (it reads LOFAR files and
trains at import time), so its loop body is restated here expression by expression on top of the
reference's own optimiser, ``src/lbfgsnew.py`` (imported unmodified), with deterministic closed-form
stand-ins for the script's random draws (oracle.rica_closed_form).  Stored: the closure's loss and
autograd gradient at the initial codes, the codes and loss after ``optimizer.step(closure)``, the
dictionary after the update and the logged ||dA||, for two consecutive minibatch iterations.

Usage:  python tests/golden/make_golden_rica.py      (writes rica.npz next to this file)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = os.environ.get("LSHM_REFERENCE", "/root/reference/src")
sys.path.insert(0, REF)


def main():
    import torch.nn as nn
    from lbfgsnew import LBFGSNew  # the reference's optimiser
    from oracle import lshm_oracle as O

    torch.manual_seed(0)
    L, M, B = 4 * 16 * 16, 32, 24
    lambda1, eta = 0.1, 0.1  # :43-44
    X, A, S0 = O.rica_closed_form(L, M, B)
    criterion = nn.MSELoss(reduction='sum')  # :52
    g = {"L": np.int64(L), "M": np.int64(M), "B": np.int64(B)}
    for it in range(2):
        nbatch = B
        S = (S0 if it == 0 else S0.flip(1)).clone().requires_grad_(True)  # :71 (fresh codes per minibatch)
        # ---- closure value / gradient at the initial codes
        loss0 = criterion(X, torch.matmul(A, S)) / (nbatch * L) + lambda1 * torch.linalg.norm(S, 1) / S.numel()
        (g0,) = torch.autograd.grad(loss0, S)
        g[f"it{it}/loss0"] = loss0.item()
        g[f"it{it}/grad0"] = g0.numpy().copy()
        optimizer = LBFGSNew([S], history_size=7, max_iter=10, line_search_fn=True, batch_mode=True)  # :73

        def closure():  # :74-81
            if torch.is_grad_enabled():
                optimizer.zero_grad()
            loss = criterion(X, torch.matmul(A, S)) / (nbatch * L) + lambda1 * torch.linalg.norm(S, 1) / S.numel()
            if loss.requires_grad:
                loss.backward()
            return loss

        optimizer.step(closure)  # :83
        with torch.no_grad():  # :85-93
            lossS = criterion(X, torch.matmul(A, S)) / (nbatch * L) + lambda1 * torch.linalg.norm(S, 1) / S.numel()
            E = X - torch.matmul(A, S)
            dA = torch.zeros((L, M))
            for ci in range(nbatch):
                dA += torch.outer(E[:, ci], S[:, ci])
            dA /= nbatch
            A = A + eta * dA
            g[f"it{it}/S"] = S.detach().numpy().copy()
            g[f"it{it}/loss_after"] = lossS.item()
            g[f"it{it}/dA_norm"] = torch.linalg.norm(dA).item()
            g[f"it{it}/A_sum"] = A.double().sum().item()
            g[f"it{it}/A_head"] = A.reshape(-1)[:64].numpy().copy()
            g[f"it{it}/func_evals"] = np.int64(optimizer.state[optimizer._params[0]]["func_evals"])
    np.savez_compressed(os.path.join(HERE, "rica.npz"), **g)
    print({k: (v if np.ndim(v) == 0 else np.shape(v)) for k, v in g.items()})


if __name__ == "__main__":
    main()
