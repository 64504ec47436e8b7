"""Dictionary learning of the reference's ``src/rica_lofar.py`` (:53-97) on the HIP kernels.

``X ~ A S`` with X (L x nbatch) the vectorised patches of a minibatch, A (L x M) the dictionary and
S (M x nbatch) the sparse codes.  Per minibatch the script draws fresh codes, minimises
``||X - A S||^2 / (nbatch L) + lambda1 ||S||_1 / S.numel()`` over S with
``LBFGSNew(history_size=7, max_iter=10, line_search_fn=True, batch_mode=True)`` and then moves the
dictionary along ``eta (X - A S) S^T / nbatch``.  :class:`RicaDictionary` keeps that loop body:
``iteration(x)`` = :71-93, with ``closure`` following the script's closure protocol (:74-81) so any
``Optimizer.step(closure)`` can drive it.

Everything stays in the loader's patch-major layout: ``x.view(-1, L)`` is X^T and the codes are held as
S^T (nbatch x M), so A S, its code gradient and the dictionary gradient are the forward, data-gradient
and weight-gradient GEMMs of one dense layer with weight A (``lshm_rica_loss_grad`` /
``lshm_rica_update_dictionary``, include/lshm.h).  There is no CPU path: without the HIP library the
constructor raises.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

from . import _lib as L_
from .lbfgsnew import LBFGSNew


class RicaDictionary:
    """State of src/rica_lofar.py: the dictionary ``A`` (L x M, ``torch.rand`` like :51 unless given) and
    the hyper-parameters ``lambda1`` (:43), ``eta`` (:44)."""

    def __init__(self, L: int, M: int = 256, lambda1: float = 0.1, eta: float = 0.1,
                 device: str = "cuda", A: Optional[torch.Tensor] = None, matrix_precision: str = "fp32"):
        if matrix_precision not in ("fp32", "bf16"):
            raise ValueError("matrix_precision must be 'fp32' or 'bf16'")
        self.bf16 = matrix_precision == "bf16"  # operands of this object's GEMMs rounded to bf16 (per call)
        self.lib = L_.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("RicaDictionary runs on a HIP device only")
        self.L, self.M, self.lambda1, self.eta = int(L), int(M), float(lambda1), float(eta)
        if A is None:
            A = torch.rand((L, M), dtype=torch.float32)
        if tuple(A.shape) != (L, M):
            raise ValueError(f"dictionary must be ({L}, {M}), got {tuple(A.shape)}")
        self.A = A.to(self.device, torch.float32).contiguous()
        self._ws = None
        self._loss = torch.zeros(1, device=self.device, dtype=torch.float64)
        self._norm = torch.zeros(1, device=self.device, dtype=torch.float64)
        self._Xt = None

    # ------------------------------------------------------------------ buffers
    def _workspace(self, B: int) -> torch.Tensor:
        n = self.lib.lshm_rica_workspace_floats(B, self.L, self.M)
        if self._ws is None or self._ws.numel() < n:
            self._ws = torch.empty(n, device=self.device, dtype=torch.float32)
        return self._ws

    def set_minibatch(self, x: torch.Tensor) -> int:
        """x: (nbatch, C, P, P) or (nbatch, L) -- ``x.view(-1, L)`` of :69 (kept transposed).  Returns nbatch."""
        Xt = x.reshape(-1, self.L).to(self.device, torch.float32).contiguous()
        self._Xt = Xt
        return Xt.shape[0]

    # ------------------------------------------------------------------ the closure (:72-81)
    def loss(self, St: torch.Tensor, want_grad: bool) -> torch.Tensor:
        """Loss at codes ``St`` = S^T (nbatch x M); with ``want_grad`` also writes ``St.grad``."""
        Xt = self._Xt
        B = Xt.shape[0]
        if tuple(St.shape) != (B, self.M) or not St.is_contiguous():
            raise ValueError("codes must be a contiguous (nbatch, M) tensor")
        ws = self._workspace(B)
        grad = None
        if want_grad:
            if St.grad is None:
                St.grad = torch.empty_like(St)
            grad = St.grad
        with L_.on_device(self.device):
            L_.check(L_.fn("lshm_rica_loss_grad", self.bf16)(L_.ptr(Xt), L_.ptr(self.A), L_.ptr(St), B, self.L, self.M,
                                                             self.lambda1, L_.ptr(self._loss), L_.ptr(grad), L_.ptr(ws),
                                                             ws.numel(), L_.stream()), "rica_loss_grad")
        return self._loss[0].clone()

    def closure_for(self, St: torch.Tensor):
        """The script's closure: gradients only when autograd is enabled (:74-81)."""
        def closure():
            return self.loss(St, torch.is_grad_enabled())
        return closure

    # ------------------------------------------------------------------ the loop body (:71-93)
    def solve_codes(self, x: torch.Tensor, S0: Optional[torch.Tensor] = None, history_size: int = 7,
                    max_iter: int = 10) -> Tuple[torch.Tensor, float]:
        """Fresh codes for this minibatch (:71, ``torch.rand`` unless ``S0`` (M x nbatch) is given) and one
        ``LBFGSNew.step`` over them (:73,83).  Returns (S (M x nbatch), loss at the solution)."""
        B = self.set_minibatch(x)
        if S0 is None:
            S0 = torch.rand((self.M, B), dtype=torch.float32)
        St = S0.to(self.device, torch.float32).t().contiguous().requires_grad_(True)
        opt = LBFGSNew([St], history_size=history_size, max_iter=max_iter, line_search_fn=True, batch_mode=True)
        opt.step(self.closure_for(St))
        with torch.no_grad():
            final = float(self.loss(St, False))
        self._St = St.detach()
        self._opt = opt
        return self._St.t().contiguous(), final

    def update_dictionary(self, S: Optional[torch.Tensor] = None) -> float:
        """:84-93 with the current minibatch: A += eta (X - A S) S^T / nbatch.  Returns the logged ||dA||."""
        St = self._St if S is None else S.to(self.device, torch.float32).t().contiguous()
        Xt = self._Xt
        B = Xt.shape[0]
        ws = self._workspace(B)
        with L_.on_device(self.device):
            L_.check(L_.fn("lshm_rica_update_dictionary", self.bf16)(L_.ptr(Xt), L_.ptr(self.A), L_.ptr(St), B, self.L,
                                                                     self.M, self.eta, L_.ptr(self._norm), L_.ptr(ws),
                                                                     ws.numel(), L_.stream()), "rica_update_dictionary")
        return math.sqrt(float(self._norm[0])) / B

    def iteration(self, x: torch.Tensor, S0: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, float, float]:
        """One minibatch of the script's loop: codes, then dictionary.  Returns (S, loss, ||dA||)."""
        S, loss = self.solve_codes(x, S0)
        return S, loss, self.update_dictionary()

    def atoms(self, channels: int, patch: int) -> torch.Tensor:
        """Columns of A as images (M, channels, patch, patch) -- what :100-103 saves."""
        return self.A.t().reshape(self.M, channels, patch, patch).contiguous()
