"""Stochastic L-BFGS optimiser with the interface of the reference's ``src/lbfgsnew.py``
(``LBFGSNew(params, lr, max_iter, max_eval, tolerance_grad, tolerance_change, history_size,
line_search_fn, batch_mode, cost_use_gradient)``, ``step(closure)``), so line 93 of
``src/kharmonic_lofar.py`` works unchanged.

Same algorithm (upstream ``lbfgsnew.py:498-759``): two-loop recursion over a bounded (s, y)
history, trust-region regularisation ``y += 1e-6 s`` and inter-batch gradient-variance step bound
in batch mode (:586-605), curvature test ``y.s > 1e-10 |s|^2`` (:610), backtracking Armijo line
search with a negative-step fallback for batch mode (:115-187) or Fletcher's cubic strong-Wolfe
search with finite-difference slopes for full-batch mode (:192-328, :330-407, :413-495).

Re-designed around ONE flat fp32 vector: parameters and gradients are gathered into flat device
buffers once per evaluation, all history vectors are flat, and on a HIP device the inner products /
axpys are the library's deterministic two-stage reductions (``lshm_dot_flat`` etc.), so every rank
of a data-parallel job takes bit-identical branches when the closure returns the all-reduced loss.
The control flow is host logic and also runs on CPU tensors (used by the CPU test-suite).
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional

import torch
from torch.optim.optimizer import Optimizer

be_verbose = False


class _TorchVec:
    """Vector algebra for CPU tensors (host-logic tests; the model kernels have no CPU path)."""

    def dot(self, a, b) -> float:
        return float(torch.dot(a, b))

    def asum(self, a) -> float:
        return float(a.abs().sum())

    def axpy(self, y, alpha, x):
        y.add_(x, alpha=alpha)

    def scale(self, x, alpha):
        x.mul_(alpha)

    def dots(self, pairs):
        return [self.dot(a, b) for a, b in pairs]

    def direction(self, old_dirs, old_stps, flat_grad, h_diag):
        return _two_loop(self, old_dirs, old_stps, flat_grad, h_diag)


def _two_loop(vec, old_dirs, old_stps, flat_grad, h_diag):
    """The two-loop recursion of src/lbfgsnew.py:632-651 on top of a vector backend's dot / axpy / scale."""
    num_old = len(old_dirs)
    ro = [1.0 / vec.dot(old_dirs[i], old_stps[i]) for i in range(num_old)]
    al = [0.0] * num_old
    q = flat_grad.neg()
    for i in range(num_old - 1, -1, -1):
        al[i] = vec.dot(old_stps[i], q) * ro[i]
        vec.axpy(q, -al[i], old_dirs[i])
    vec.scale(q, float(h_diag))
    d = q
    for i in range(num_old):
        be_i = vec.dot(old_dirs[i], d) * ro[i]
        vec.axpy(d, al[i] - be_i, old_stps[i])
    return d


class _HipVec:
    """Vector algebra through the C ABI (deterministic two-stage reductions on the device)."""

    def __init__(self, device):
        from . import _lib as L
        self.L = L
        self.lib = L.load()
        self.ws = torch.empty(1024, device=device, dtype=torch.float32)
        self.out = torch.empty(1, device=device, dtype=torch.float64)
        self.outs = torch.empty(self.MAX_PAIRS, device=device, dtype=torch.float64)
        self.dws = torch.empty(0, device=device, dtype=torch.float64)

    def dot(self, a, b) -> float:
        L = self.L
        with L.on_device(a.device):
            L.check(self.lib.lshm_dot_flat(L.ptr(a), L.ptr(b), a.numel(), L.ptr(self.out), L.ptr(self.ws), L.stream(a.device)))
        return float(self.out.item())

    def asum(self, a) -> float:
        L = self.L
        with L.on_device(a.device):
            L.check(self.lib.lshm_asum_flat(L.ptr(a), a.numel(), L.ptr(self.out), L.ptr(self.ws), L.stream(a.device)))
        return float(self.out.item())

    def axpy(self, y, alpha, x):
        L = self.L
        with L.on_device(y.device):
            L.check(self.lib.lshm_axpy_flat(L.ptr(y), L.ptr(x), float(alpha), y.numel(), L.stream(y.device)))

    def scale(self, x, alpha):
        L = self.L
        with L.on_device(x.device):
            L.check(self.lib.lshm_scale_flat(L.ptr(x), float(alpha), x.numel(), L.stream(x.device)))

    MAX_PAIRS = 16  # kMaxDots of the C ABI

    def _ptr_array(self, vs):
        import ctypes as C
        return (C.c_void_p * len(vs))(*[v.data_ptr() for v in vs])

    def dots(self, pairs) -> List[float]:
        """[a.b for a, b in pairs] with ONE host read (lshm_multi_dot_flat)."""
        if not 1 <= len(pairs) <= self.MAX_PAIRS:
            return [self.dot(a, b) for a, b in pairs]
        L = self.L
        a0 = pairs[0][0]
        need = int(self.lib.lshm_multi_dot_workspace_doubles(len(pairs)))
        if self.dws.numel() < need:
            self.dws = torch.empty(need, device=a0.device, dtype=torch.float64)
        with L.on_device(a0.device):
            L.check(self.lib.lshm_multi_dot_flat(self._ptr_array([a for a, _ in pairs]), self._ptr_array([b for _, b in pairs]),
                                                 len(pairs), a0.numel(), L.ptr(self.outs), L.ptr(self.dws),
                                                 self.dws.numel(), L.stream(a0.device)), "multi_dot")
        return self.outs[:len(pairs)].tolist()

    def direction(self, old_dirs, old_stps, flat_grad, h_diag):
        """Two-loop recursion on the device (lshm_lbfgs_direction): no host round trip per inner product."""
        m = len(old_dirs)
        if m > self.MAX_PAIRS:
            return _two_loop(self, old_dirs, old_stps, flat_grad, h_diag)
        L = self.L
        d = torch.empty_like(flat_grad)
        need = int(self.lib.lshm_lbfgs_direction_workspace_doubles(m))
        if self.dws.numel() < need:
            self.dws = torch.empty(need, device=flat_grad.device, dtype=torch.float64)
        with L.on_device(flat_grad.device):
            L.check(self.lib.lshm_lbfgs_direction(self._ptr_array(old_dirs) if m else None,
                                                  self._ptr_array(old_stps) if m else None, m, L.ptr(flat_grad),
                                                  float(h_diag), L.ptr(d), flat_grad.numel(), L.ptr(self.dws),
                                                  self.dws.numel(), L.stream(flat_grad.device)), "lbfgs_direction")
        return d


class LBFGSNew(Optimizer):
    def __init__(self, params, lr=1, max_iter=10, max_eval=None, tolerance_grad=1e-5, tolerance_change=1e-9,
                 history_size=7, line_search_fn=False, batch_mode=False, cost_use_gradient=False,
                 reuse_known_loss=False):
        """reuse_known_loss (not upstream; off by default): the line searches start by evaluating the closure at the
        point where step() has just evaluated it (src/lbfgsnew.py:277, :472).  For a deterministic closure that value
        is known: with the option on it is reused instead of recomputed (one gradient-free evaluation less per inner
        iteration; evaluation counters unchanged -- upstream does not count these either)."""
        self._reuse_known_loss = bool(reuse_known_loss)
        if max_eval is None:
            max_eval = max_iter * 5 // 4
        defaults = dict(lr=lr, max_iter=max_iter, max_eval=max_eval, tolerance_grad=tolerance_grad,
                        tolerance_change=tolerance_change, history_size=history_size,
                        line_search_fn=line_search_fn, batch_mode=batch_mode, cost_use_gradient=cost_use_gradient)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError("LBFGS doesn't support per-parameter options (parameter groups)")
        self._params = self.param_groups[0]["params"]
        self._n = sum(p.numel() for p in self._params)
        dev = self._params[0].device
        self._vec = _HipVec(dev) if dev.type == "cuda" else _TorchVec()
        # a single contiguous fp32 parameter IS the flat vector: no gather / scatter needed
        self._single = len(self._params) == 1 and self._params[0].is_contiguous()

    # ------------------------------------------------------------------ flat views
    def _flat_grad(self) -> torch.Tensor:
        if self._single:
            p = self._params[0]
            g = p.grad
            return (g.detach().reshape(-1).clone() if g is not None else torch.zeros(self._n, device=p.device))
        parts = []
        for p in self._params:
            if p.grad is None:
                parts.append(p.new_zeros(p.numel()))
            elif p.grad.is_sparse:
                parts.append(p.grad.to_dense().reshape(-1))
            else:
                parts.append(p.grad.detach().reshape(-1))
        return torch.cat(parts, 0)

    def _move(self, alpha: float, direction: torch.Tensor):
        """params += alpha * direction."""
        if self._single:
            self._vec.axpy(self._params[0].data.view(-1), alpha, direction)
            return
        off = 0
        for p in self._params:
            n = p.numel()
            p.data.add_(direction[off:off + n].view_as(p.data), alpha=alpha)
            off += n

    def _snapshot(self) -> List[torch.Tensor]:
        return [p.detach().clone(memory_format=torch.contiguous_format) for p in self._params]

    def _restore(self, snap: List[torch.Tensor]):
        with torch.no_grad():
            for p, s in zip(self._params, snap):
                p.copy_(s)

    def _st(self):
        return self.state[self._params[0]]

    # ------------------------------------------------------------------ line searches
    def _linesearch_backtrack(self, closure, pk, gk, alphabar, f_known=None):
        """Armijo backtracking from alphabar; if the decrease is too small also try negative steps."""
        c1, citer = 1e-4, 35
        alphak = alphabar
        xk = self._snapshot()
        f_old = float(closure()) if f_known is None else f_known
        self._move(alphak, pk)
        f_new = float(closure())
        prodterm = c1 * self._vec.dot(gk, pk)
        ci = 0
        while ci < citer and (math.isnan(f_new) or f_new > f_old + alphak * prodterm):
            alphak = 0.5 * alphak
            self._restore(xk)
            self._move(alphak, pk)
            f_new = float(closure())
            ci += 1
        if f_old - f_new < abs(prodterm):
            alphak1 = -alphabar
            self._restore(xk)
            self._move(alphak1, pk)
            f_new1 = float(closure())
            while ci < citer and (math.isnan(f_new1) or f_new1 > f_old + alphak1 * prodterm):
                alphak1 = 0.5 * alphak1
                self._restore(xk)
                self._move(alphak1, pk)
                f_new1 = float(closure())
                ci += 1
            if f_new1 < f_new:
                alphak = alphak1
        self._restore(xk)
        self._st()["func_evals"] += ci
        return alphak

    def _slope_here(self, closure, pk, step):
        """Central finite difference of phi at the current point; leaves params at (current - step)."""
        self._move(step, pk)
        up = float(closure())
        self._move(-2.0 * step, pk)
        dn = float(closure())
        return (up - dn) / (2.0 * step)

    def _cubic_interpolate(self, closure, xk, pk, a, b, step):
        self._restore(xk)
        st = self._st()
        self._move(a, pk)
        f0 = float(closure())
        f0d = self._slope_here(closure, pk, step)          # now at a - step
        self._move(-a + step + b, pk)                      # -> b
        f1 = float(closure())
        f1d = self._slope_here(closure, pk, step)          # now at b - step
        evals = 6
        aa = 3.0 * (f0 - f1) / (b - a) + f1d - f0d
        disc = aa * aa - f0d * f1d
        if disc > 0.0:
            cc = math.sqrt(disc)
            if (f1d - f0d + 2.0 * cc) == 0.0:
                return (a + b) * 0.5
            z0 = b - (f1d + cc - aa) * (b - a) / (f1d - f0d + 2.0 * cc)
            hi, lo = max(a, b), min(a, b)
            if z0 > hi or z0 < lo:
                fz0 = f0 + f1
            else:
                self._move(-b + step + a + z0 * (b - a), pk)
                fz0 = float(closure())
                evals += 1
            st["func_evals"] += evals
            if f0 < f1 and f0 < fz0:
                return a
            if f1 < fz0:
                return b
            return z0
        st["func_evals"] += evals
        return a if f0 < f1 else b

    def _linesearch_zoom(self, closure, xk, pk, a, b, phi_0, gphi_0, sigma, rho, t1, t2, t3, step):
        st = self._st()
        evals = 0
        aj, bj = a, b
        alphaj = a
        found = False
        for _ in range(4):
            alphaj = self._cubic_interpolate(closure, xk, pk, aj + t2 * (bj - aj), bj - t3 * (bj - aj), step)
            self._restore(xk)
            self._move(alphaj, pk)
            phi_j = float(closure())
            self._move(-alphaj + aj, pk)
            phi_aj = float(closure())
            evals += 2
            if phi_j > phi_0 + rho * alphaj * gphi_0 or phi_j >= phi_aj:
                bj = alphaj
            else:
                self._move(-aj + alphaj, pk)               # back to alphaj
                gphi_j = self._slope_here(closure, pk, step)
                evals += 2
                if (aj - alphaj) * gphi_j <= step or abs(gphi_j) <= -sigma * gphi_0:
                    found = True
                    break
                if gphi_j * (bj - aj) >= 0.0:
                    bj = aj
                aj = alphaj
        st["func_evals"] += evals
        return alphaj

    def _linesearch_cubic(self, closure, pk, step, phi_known=None):
        lr = self.param_groups[0]["lr"]
        alpha1, sigma, rho, t1, t2, t3 = 10 * lr, 0.1, 0.01, 9, 0.1, 0.5
        alphak = lr
        st = self._st()
        xk = self._snapshot()
        phi_0 = float(closure()) if phi_known is None else phi_known
        tol = min(phi_0 * 0.01, 1e-6)
        gphi_0 = self._slope_here(closure, pk, step)
        if abs(gphi_0) < 1e-12:
            return 1.0
        mu = (tol - phi_0) / (rho * gphi_0)
        if math.isnan(mu):
            return 1.0
        evals = 3
        ci = 1
        alphai, alphai1, phi_alphai1 = alpha1, 0.0, phi_0
        while ci < 4:
            self._restore(xk)
            self._move(alphai, pk)
            phi_alphai = float(closure())
            if phi_alphai < tol:
                alphak = alphai
                break
            if phi_alphai > phi_0 + alphai * gphi_0 or (ci > 1 and phi_alphai >= phi_alphai1):
                alphak = self._linesearch_zoom(closure, xk, pk, alphai1, alphai, phi_0, gphi_0, sigma, rho, t1, t2,
                                               t3, step)
                break
            gphi_i = self._slope_here(closure, pk, step)
            if abs(gphi_i) <= -sigma * gphi_0:
                alphak = alphai
                break
            if gphi_i >= 0.0:
                alphak = self._linesearch_zoom(closure, xk, pk, alphai, alphai1, phi_0, gphi_0, sigma, rho, t1, t2,
                                               t3, step)
                break
            if mu <= 2.0 * alphai - alphai1:
                alphai1, alphai = alphai, mu
            else:
                lo = 2.0 * alphai - alphai1
                hi = min(mu, alphai + t1 * (alphai - alphai1))
                alphai = self._cubic_interpolate(closure, xk, pk, lo, hi, step)
            phi_alphai1 = phi_alphai
            evals += 3
            ci += 1
        self._restore(xk)
        st["func_evals"] += evals
        return alphak

    # ------------------------------------------------------------------ step
    def step(self, closure: Callable):
        group = self.param_groups[0]
        lr, max_iter, max_eval = group["lr"], group["max_iter"], group["max_eval"]
        tol_grad, tol_change = group["tolerance_grad"], group["tolerance_change"]
        line_search, hist = group["line_search_fn"], group["history_size"]
        batch_mode, cost_use_gradient = group["batch_mode"], group["cost_use_gradient"]
        vec = self._vec
        st = self._st()
        st.setdefault("func_evals", 0)
        st.setdefault("n_iter", 0)

        orig_loss = closure()
        loss = float(orig_loss.detach()) if isinstance(orig_loss, torch.Tensor) else float(orig_loss)
        current_evals = 1
        st["func_evals"] += 1
        flat_grad = self._flat_grad()
        abs_grad_sum = vec.asum(flat_grad)
        if abs_grad_sum <= tol_grad:
            return orig_loss

        d, t = st.get("d"), st.get("t")
        old_dirs, old_stps = st.get("old_dirs"), st.get("old_stps")
        H_diag = st.get("H_diag")
        prev_flat_grad, prev_loss = st.get("prev_flat_grad"), st.get("prev_loss")
        running_avg = running_avg_sq = None
        n_iter = 0
        alphabar, lm0 = lr, 1e-6
        grad_nrm = math.sqrt(vec.dot(flat_grad, flat_grad))
        while n_iter < max_iter and not math.isnan(grad_nrm):
            n_iter += 1
            st["n_iter"] += 1
            # ---------------- direction
            if st["n_iter"] == 1:
                d = flat_grad.neg()
                old_dirs, old_stps, H_diag = [], [], 1
                if batch_mode:
                    running_avg = torch.zeros_like(flat_grad)
                    running_avg_sq = torch.zeros_like(flat_grad)
            else:
                if batch_mode:
                    running_avg, running_avg_sq = st.get("running_avg"), st.get("running_avg_sq")
                    if running_avg is None:
                        running_avg = torch.zeros_like(flat_grad)
                        running_avg_sq = torch.zeros_like(flat_grad)
                y = flat_grad.clone()
                vec.axpy(y, -1.0, prev_flat_grad)
                s = d.clone()
                vec.scale(s, t)
                if batch_mode:
                    vec.axpy(y, lm0, s)  # trust region
                ys, ss, yy = vec.dots([(y, s), (s, s), (y, y)])  # one host read for the three curvature products
                sn = math.sqrt(ss)
                batch_changed = batch_mode and (n_iter == 1 and st["n_iter"] > 1)
                if batch_changed:
                    # online inter-batch mean / second moment of the gradient -> bound on the step
                    g_old = flat_grad - running_avg
                    running_avg.add_(g_old, alpha=1.0 / st["n_iter"])
                    g_new = flat_grad - running_avg
                    running_avg_sq.addcmul_(g_new, g_old, value=1)
                    alphabar = 1.0 / (1.0 + float(running_avg_sq.sum()) / ((st["n_iter"] - 1) * grad_nrm))
                if ys > 1e-10 * sn * sn and not batch_changed:
                    if len(old_dirs) == hist:
                        old_dirs.pop(0)
                        old_stps.pop(0)
                    old_dirs.append(y)
                    old_stps.append(s)
                    H_diag = ys / yy
                if isinstance(H_diag, float) and math.isnan(H_diag):
                    print("Warning H_diag nan")
                d = vec.direction(old_dirs, old_stps, flat_grad, H_diag)
            if prev_flat_grad is None:
                prev_flat_grad = flat_grad.clone()
            else:
                prev_flat_grad.copy_(flat_grad)
            prev_loss = loss
            # ---------------- step length
            t = min(1.0, 1.0 / abs_grad_sum) * lr if st["n_iter"] == 1 else lr
            gtd = vec.dot(flat_grad, d)
            if math.isnan(gtd):
                print("Warning grad norm infinite")
            ls_func_evals = 0
            if line_search:
                grad_was = torch.is_grad_enabled()
                if not cost_use_gradient:
                    torch.set_grad_enabled(False)
                try:
                    known = loss if self._reuse_known_loss else None  # the closure value at the current point
                    if batch_mode:
                        t = self._linesearch_backtrack(closure, d, flat_grad, alphabar, known)
                    else:
                        t = self._linesearch_cubic(closure, d, 1e-6, known)
                finally:
                    torch.set_grad_enabled(grad_was)
                if math.isnan(t):
                    print("Warning: stepsize nan")
                    t = lr
                self._move(t, d)
                if be_verbose:
                    print("step size=%f" % t)
            else:
                self._move(t, d)
            if n_iter != max_iter:
                loss = float(closure())
                flat_grad = self._flat_grad()
                abs_grad_sum = vec.asum(flat_grad)
                if math.isnan(abs_grad_sum):
                    print("Warning: gradient nan")
                    break
                ls_func_evals = 1
            current_evals += ls_func_evals
            st["func_evals"] += ls_func_evals
            # ---------------- stopping rules
            if n_iter == max_iter or current_evals >= max_eval:
                break
            if abs_grad_sum <= tol_grad or gtd > -tol_change:
                break
            if abs(t) * vec.asum(d) <= tol_change:
                break
            if abs(loss - prev_loss) < tol_change:
                break
        st["d"], st["t"] = d, t
        st["old_dirs"], st["old_stps"] = old_dirs, old_stps
        st["H_diag"] = H_diag
        st["prev_flat_grad"], st["prev_loss"] = prev_flat_grad, prev_loss
        if batch_mode:
            if running_avg is None:
                running_avg = torch.zeros_like(flat_grad)
                running_avg_sq = torch.zeros_like(flat_grad)
            st["running_avg"], st["running_avg_sq"] = running_avg, running_avg_sq
        return orig_loss
