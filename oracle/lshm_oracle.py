"""CPU oracle for the LSHM cascaded-AE + k-harmonic training step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``lshm_amd/`` may import this module;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and only as the checker / baseline, never as the product.

This is a from-scratch *restatement* (functional, state-dict driven, torch-CPU
tensor ops) of the reference's hot path.  Every function cites the reference
file:line it follows (paths relative to the upstream repo root).  The
reference's arithmetic lives in third-party PyTorch ATen ops (version unpinned
upstream; this image has torch 2.10.0 CPU), so the restatement composes the same
ATen ops on CPU; `oracle/lshm_oracle_c.c` is an independent plain-C restatement
of the per-op math used to cross-check this file.

PINNING: the reference ships no tests / golden vectors for this path
(SURVEY.md section 4).  The oracle is pinned against outputs of the reference
itself, produced in the build container by ``tests/golden/make_golden.py``
(imports upstream ``lofar_models.py`` on CPU) and committed as
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this file
against those fixtures.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
StateDict = Dict[str, Tensor]

CH = (4, 8, 12, 24, 48, 96, 192)  # channel ladder, src/lofar_models.py:31-41
DEFAULT_SCALES = (1e-4, 1e-3, 1e-2, 1e-1)  # src/kharmonic_lofar.py:57
EPS_KHM = 1e-9  # src/lofar_models.py:195


# --------------------------------------------------------------------------
# deterministic closed-form fills (shared by golden generator, tests, smoke)
# --------------------------------------------------------------------------
def _key_phase(key: str) -> float:
    return 0.1 * (sum(ord(c) * (i % 7 + 1) for i, c in enumerate(key)) % 997)


def closed_form(shape: Sequence[int], key: str, scale: float = 1.0,
                freq: float = 0.7390851332151607, dtype=torch.float32) -> Tensor:
    """w.flat[i] = scale * sin(freq*i + phase(key)), evaluated in float64."""
    n = int(np.prod(shape)) if len(shape) else 1
    i = np.arange(n, dtype=np.float64)
    v = scale * np.sin(freq * i + _key_phase(key))
    return torch.from_numpy(v.reshape(tuple(shape))).to(dtype)


def ae_param_shapes(latent_dim: int, channels: int, ndim: int, rica: bool,
                    harmonic_dim: int = 16) -> Dict[str, Tuple[int, ...]]:
    """Parameter names / torch layouts of AutoEncoderCNN2 (ndim=2,
    src/lofar_models.py:31-57) and AutoEncoder1DCNN (ndim=1, :115-142)."""
    ch = (channels,) + CH[1:]
    k = (4, 4) if ndim == 2 else (4,)
    shapes: Dict[str, Tuple[int, ...]] = {}
    for i in range(6):
        shapes[f"conv{i}.weight"] = (ch[i + 1], ch[i]) + k
        shapes[f"conv{i}.bias"] = (ch[i + 1],)
    shapes["fcuv1.weight"] = (harmonic_dim, harmonic_dim)
    shapes["fcuv1.bias"] = (harmonic_dim,)
    shapes["fcuv3.weight"] = (harmonic_dim, harmonic_dim)
    shapes["fcuv3.bias"] = (harmonic_dim,)
    shapes["fc1.weight"] = (latent_dim, 768 + harmonic_dim)
    shapes["fc1.bias"] = (latent_dim,)
    if rica:
        shapes["fc2in.weight"] = (latent_dim, latent_dim)
        shapes["fc2in.bias"] = (latent_dim,)
        shapes["fc2out.weight"] = (latent_dim, latent_dim)
        shapes["fc2out.bias"] = (latent_dim,)
    shapes["fc3.weight"] = (768, latent_dim + harmonic_dim)
    shapes["fc3.bias"] = (768,)
    for i in range(6):
        shapes[f"tconv{i}.weight"] = (ch[6 - i], ch[5 - i]) + k
        shapes[f"tconv{i}.bias"] = (ch[5 - i],)
    return shapes


def _fan_in(name: str, shape: Tuple[int, ...]) -> int:
    if name.endswith("bias"):
        return 0
    if name.startswith("tconv"):  # torch uses weight.size(1)*receptive field
        return int(shape[1] * np.prod(shape[2:]))
    return int(np.prod(shape[1:]))


def closed_form_state_dict(latent_dim: int, channels: int, ndim: int, rica: bool,
                           tag: str = "ae", gain: float = 1.0,
                           dtype=torch.float32) -> StateDict:
    """Deterministic parameters with roughly default-init magnitudes."""
    shapes = ae_param_shapes(latent_dim, channels, ndim, rica)
    sd: StateDict = {}
    fan = {}
    for name, shp in shapes.items():
        if name.endswith("weight"):
            fan[name[:-7]] = _fan_in(name, shp)
    for name, shp in shapes.items():
        base = name.rsplit(".", 1)[0]
        s = gain * math.sqrt(3.0) / math.sqrt(fan[base])
        if name.endswith("bias"):
            s = gain / math.sqrt(fan[base])
        sd[name] = closed_form(shp, tag + ":" + name, scale=s, dtype=dtype)
    return sd


def closed_form_inputs(B: int, channels: int = 4, patch: int = 128,
                       dtype=torch.float32) -> Tuple[Tensor, Tensor]:
    """Standardised-looking patches and wavelength-scale uv coordinates."""
    n = B * channels * patch * patch
    i = np.arange(n, dtype=np.float64)
    x = 1.1 * np.sin(0.0137 * i + 0.3) + 0.6 * np.sin(1.7 * i + 1.1) \
        + 0.4 * np.cos(0.00091 * i * (1 + (i % 5)))
    x = x.reshape(B, channels, patch, patch)
    j = np.arange(B * 2, dtype=np.float64)
    uv = 1000.0 * np.sin(2.3 * j + 0.5) + 300.0 * np.cos(0.37 * j)
    uv = uv.reshape(B, 2)
    return torch.from_numpy(x).to(dtype), torch.from_numpy(uv).to(dtype)


# --------------------------------------------------------------------------
# a2: harmonic features           (src/lofar_models.py:60-62, 145-147)
# --------------------------------------------------------------------------
def uv_harmonics(scales: Tensor, uv: Tensor) -> Tensor:
    """(B,2) -> (B,4H): [sin(s0 u),sin(s0 v),sin(s1 u),...,cos(s0 u),...].

    kron of a (H,) vector with a (B,2) matrix is a (B,2H) matrix whose column
    2h+c equals scales[h]*uv[:,c]."""
    arg = (scales.reshape(1, -1, 1) * uv.reshape(uv.shape[0], 1, 2)).reshape(uv.shape[0], -1)
    return torch.cat((torch.sin(arg), torch.cos(arg)), dim=1)


# --------------------------------------------------------------------------
# a3-a6: the autoencoders
# --------------------------------------------------------------------------
def _conv(ndim: int):
    return F.conv2d if ndim == 2 else F.conv1d


def _tconv(ndim: int):
    return F.conv_transpose2d if ndim == 2 else F.conv_transpose1d


def ae_encode(sd: StateDict, x: Tensor, uvh: Tensor, ndim: int) -> Tensor:
    """src/lofar_models.py:71-84 (2D: k4 s2 p1) / :156-169 (1D: k4 s4 p1)."""
    stride = 2 if ndim == 2 else 4
    h = x
    for i in range(6):
        h = F.elu(_conv(ndim)(h, sd[f"conv{i}.weight"], sd[f"conv{i}.bias"],
                              stride=stride, padding=1))
    h = h.flatten(1)
    u = F.elu(F.linear(uvh, sd["fcuv1.weight"], sd["fcuv1.bias"]))
    return F.elu(F.linear(torch.cat((h, u), 1), sd["fc1.weight"], sd["fc1.bias"]))


def ae_decode(sd: StateDict, z: Tensor, uvh: Tensor, ndim: int) -> Tensor:
    """src/lofar_models.py:86-99 (2D: k4 s2 p1) / :171-184 (1D: k4 s4 p0)."""
    stride, pad = (2, 1) if ndim == 2 else (4, 0)
    u = F.elu(F.linear(uvh, sd["fcuv3.weight"], sd["fcuv3.bias"]))
    h = F.linear(torch.cat((z, u), 1), sd["fc3.weight"], sd["fc3.bias"])
    h = h.reshape((-1, 192, 2, 2) if ndim == 2 else (-1, 192, 4))
    for i in range(6):
        h = _tconv(ndim)(h, sd[f"tconv{i}.weight"], sd[f"tconv{i}.bias"],
                         stride=stride, padding=pad)
        if i < 5:
            h = F.elu(h)
    return h


def ae_forward(sd: StateDict, x: Tensor, uv: Tensor, scales: Tensor, ndim: int,
               rica: bool) -> Tuple[Tensor, Tensor]:
    """forward(): src/lofar_models.py:59-69 (2D) / :144-154 (1D).

    With rica the returned latent is the post-fc2in code.  The upstream 1D
    class cannot run with rica=False (:150 drops `uv`); here that mode simply
    passes the harmonic features, which is the evident intent."""
    uvh = uv_harmonics(scales, uv)
    mu = ae_encode(sd, x, uvh, ndim)
    if not rica:
        return ae_decode(sd, mu, uvh, ndim), mu
    mu = F.elu(F.linear(mu, sd["fc2in.weight"], sd["fc2in.bias"]))
    mup = F.elu(F.linear(mu, sd["fc2out.weight"], sd["fc2out.bias"]))
    return ae_decode(sd, mup, uvh, ndim), mu


# --------------------------------------------------------------------------
# a8-a10: k-harmonic means
# --------------------------------------------------------------------------
def khm_loss_loop(X: Tensor, M: Tensor, p: float, eps: float = EPS_KHM) -> Tensor:
    """Faithful evaluation order of Kmeans.forward (src/lofar_models.py:199-209):
    per sample, per centroid, norm -> pow -> reciprocal accumulation."""
    B, D = X.shape
    K = M.shape[0]
    total = 0
    for i in range(B):
        e = 0
        for k in range(K):
            d = torch.linalg.norm(M[k] - X[i], 2)
            e = e + 1.0 / (d.pow(p) + eps)
        total = total + K / (e + eps)
    return total / (B * K * D)


def khm_loss(X: Tensor, M: Tensor, p: float, eps: float = EPS_KHM) -> Tensor:
    """Vectorised, autograd-safe form of the same sum.
    Uses (d^2)^(p/2) so the gradient at d == 0 is finite (for p >= 2), matching
    the closed form the HIP kernel implements."""
    B, D = X.shape
    K = M.shape[0]
    s = ((X[:, None, :] - M[None, :, :]) ** 2).sum(-1)
    g = s.pow(p / 2.0) + eps
    e = (1.0 / g).sum(1)
    return (K / (e + eps)).sum() / (B * K * D)


def khm_grads(X: Tensor, M: Tensor, p: float, eps: float = EPS_KHM
              ) -> Tuple[Tensor, Tensor, Tensor]:
    """Closed-form loss, dX, dM, float64 internally."""
    Xd, Md = X.double(), M.double()
    B, D = Xd.shape
    K = Md.shape[0]
    s = ((Xd[:, None, :] - Md[None, :, :]) ** 2).sum(-1)
    g = s.pow(p / 2.0) + eps
    e = (1.0 / g).sum(1)
    c = 1.0 / (B * K * D)
    loss = c * (K / (e + eps)).sum()
    W = (c * K / (e + eps) ** 2)[:, None] * p * s.pow(p / 2.0 - 1.0) / g ** 2
    dX = W.sum(1)[:, None] * Xd - W @ Md
    dM = W.sum(0)[:, None] * Md - W.t() @ Xd
    return loss, dX, dM


def khm_offline_partials(X: Tensor, M: Tensor, p: float, eps: float = EPS_KHM
                         ) -> Tuple[Tensor, Tensor]:
    """Intent of Kmeans.offline_update (src/lofar_models.py:231-261; upstream has
    a typo at :248 and cannot run): numerator (K,D) and denominator (K) of
    Zhang's generalised KHM recursion.  float64.  No reference golden exists."""
    Xd, Md = X.double(), M.double()
    s = ((Xd[:, None, :] - Md[None, :, :]) ** 2).sum(-1)
    e = (1.0 / (s.pow(p / 2.0) + eps)).sum(1)
    alpha = 1.0 / (e ** 2 + eps)
    Q = alpha[:, None] / (s.pow((p + 2) / 2.0) + eps)
    return Q.t() @ Xd, Q.sum(0)


def khm_offline_update(X: Tensor, M: Tensor, p: float, eps: float = EPS_KHM) -> Tensor:
    num, den = khm_offline_partials(X, M, p, eps)
    return num / den[:, None]


def cluster_similarity_loop(M: Tensor, eps: float = EPS_KHM) -> Tensor:
    """Faithful order of Kmeans.cluster_similarity (src/lofar_models.py:214-229)."""
    K, D = M.shape
    total = 0
    for i in range(K):
        ni = torch.linalg.norm(M[i], 2)
        den = torch.exp(torch.dot(M[i], M[i]) / (ni * ni + eps))
        num = 0
        for j in range(K):
            if j != i:
                nj = torch.linalg.norm(M[j], 2)
                num = num + torch.exp(torch.dot(M[i], M[j]) / (ni * nj + eps))
        total = total + num / (den + eps)
    return total / (K * D)


def cluster_similarity(M: Tensor, eps: float = EPS_KHM) -> Tensor:
    """Vectorised form."""
    K, D = M.shape
    n = torch.linalg.norm(M, dim=1)
    G = M @ M.t()
    E = torch.exp(G / (n[:, None] * n[None, :] + eps))
    diag = torch.diagonal(E)
    num = E.sum(1) - diag
    return (num / (diag + eps)).sum() / (K * D)


# --------------------------------------------------------------------------
# a11: augmented loss              (src/kharmonic_lofar.py:97-110)
# --------------------------------------------------------------------------
def augmented_loss_loop(mu: Tensor, bpb: int, batch_size: int) -> Tensor:
    loss = torch.zeros(1, dtype=mu.dtype)
    for g in range(batch_size):
        Z = mu[g * bpb:(g + 1) * bpb]
        acc = torch.zeros(1, dtype=mu.dtype)
        for i in range(Z.shape[0]):
            zi = Z[i] / (torch.norm(Z[i]) + 1e-6)
            for j in range(i + 1, Z.shape[0]):
                zj = Z[j] / (torch.norm(Z[j]) + 1e-6)
                acc = acc + torch.exp(-torch.dot(zi, zj))
        loss = loss + acc / bpb
    return loss / (batch_size * bpb)


def augmented_loss(mu: Tensor, bpb: int, batch_size: int) -> Tensor:
    """Vectorised.  Rows beyond batch_size*bpb are ignored;
    a trailing partial / empty group contributes only its existing pairs, as the
    slicing in the loop form does."""
    rows = min(mu.shape[0], batch_size * bpb)
    Zh = mu[:rows] / (torch.linalg.norm(mu[:rows], dim=1, keepdim=True) + 1e-6)
    total = torch.zeros(1, dtype=mu.dtype)
    for g in range((rows + bpb - 1) // bpb):
        Z = Zh[g * bpb:(g + 1) * bpb]
        E = torch.exp(-(Z @ Z.t()))
        total = total + torch.triu(E, diagonal=1).sum() / bpb
    return total / (batch_size * bpb)


# --------------------------------------------------------------------------
# a16: FFT feature step (Demo.ipynb:169-175, src/lofar_tools.py:24-30)
# --------------------------------------------------------------------------
def fft_features(r: Tensor, clamp: float = 10.0) -> Tensor:
    """fftn(dims 2,3, ortho) -> roll by size//2 on dims 2,3 -> cat(Re,Im) -> clamp."""
    Fc = torch.fft.fftn(r, dim=(2, 3), norm="ortho")
    re, im = Fc.real, Fc.imag
    for d in (2, 3):
        re = torch.roll(re, shifts=re.shape[d] // 2, dims=d)
        im = torch.roll(im, shifts=im.shape[d] // 2, dims=d)
    return torch.cat((re, im), dim=1).clamp(-clamp, clamp)


# --------------------------------------------------------------------------
# a7, a12-a15, a17: one ADMM iteration  (src/kharmonic_lofar.py:128-202)
# --------------------------------------------------------------------------
class StepConfig:
    """Script constants, src/kharmonic_lofar.py:25-57,92."""

    def __init__(self, L=224, Lt=16, K=10, p=4, alpha=0.01, beta=0.01, gamma=0.01,
                 rho=1.0, rica=True, rica_lambda=0.01, bpb=8, batch_size=None,
                 lr=1e-4, channels=4, scales=DEFAULT_SCALES):
        self.L, self.Lt, self.K, self.p = L, Lt, K, p
        self.alpha, self.beta, self.gamma, self.rho = alpha, beta, gamma, rho
        self.rica, self.rica_lambda = rica, rica_lambda
        self.bpb, self.batch_size = bpb, batch_size
        self.lr, self.channels = lr, channels
        self.scales = tuple(scales)


def three_ae_forward(params: Dict[str, StateDict], x: Tensor, uv: Tensor,
                     cfg: StepConfig):
    """src/kharmonic_lofar.py:135-150 (and the no-grad copy at :188-198)."""
    sc = torch.tensor(cfg.scales, dtype=x.dtype)
    x1, mu = ae_forward(params["net"], x, uv, sc, 2, cfg.rica)
    x11 = (x - x1) / 2
    yT, muT = ae_forward(params["netT"], x11.flatten(2, 3), uv, sc, 1, cfg.rica)
    x2 = yT.view_as(x11)
    yF, muF = ae_forward(params["netF"], x11.transpose(2, 3).flatten(2, 3), uv, sc, 1, cfg.rica)
    x3 = yF.view_as(x11).transpose(2, 3)
    return x1, x11, x2, x3, mu, muT, muF


def closure_losses(params: Dict[str, StateDict], M: Tensor, x: Tensor, uv: Tensor,
                   y: Sequence[Tensor], cfg: StepConfig, khm_fn=khm_loss,
                   sim_fn=cluster_similarity, aug_fn=augmented_loss):
    """src/kharmonic_lofar.py:132-182.  Returns (total, terms) where terms is the
    logged tuple (loss0, loss1, loss2, loss3, kdist, aug, sim, rica)."""
    n = x.numel()
    x1, x11, x2, x3, mu, muT, muF = three_ae_forward(params, x, uv, cfg)
    xr = x1 + x2 + x3
    sse = lambda a, b: ((a - b) ** 2).sum()
    loss0 = sse(xr, x) / n
    loss1 = (torch.dot(y[0], (x - x1).reshape(-1)) + cfg.rho / 2 * sse(x, x1)) / n
    loss2 = (torch.dot(y[1], (x11 - x2).reshape(-1)) + cfg.rho / 2 * sse(x11, x2)) / n
    loss3 = (torch.dot(y[2], (x11 - x3).reshape(-1)) + cfg.rho / 2 * sse(x11, x3)) / n
    Mu = torch.cat((mu, muT, muF), 1)
    bs = cfg.batch_size if cfg.batch_size is not None else x.shape[0] // cfg.bpb
    kdist = cfg.alpha * khm_fn(Mu, M, cfg.p)
    sim = cfg.beta * sim_fn(M)
    aug = cfg.gamma * aug_fn(Mu, cfg.bpb, bs).reshape(())
    total = loss0 + loss1 + loss2 + loss3 + kdist + aug + sim
    rica = torch.zeros((), dtype=x.dtype)
    if cfg.rica:
        lc = lambda t: torch.log(torch.cosh(t)).sum() / t.numel()
        rica = cfg.rica_lambda * (lc(mu) + lc(muT) + lc(muF))
        total = total + rica
    return total, (loss0, loss1, loss2, loss3, kdist, aug, sim, rica)


def multiplier_update(params, x, uv, y, cfg: StepConfig) -> List[Tensor]:
    """src/kharmonic_lofar.py:187-202."""
    with torch.no_grad():
        x1, x11, x2, x3, *_ = three_ae_forward(params, x, uv, cfg)
        return [y[0] + cfg.rho * (x - x1).reshape(-1),
                y[1] + cfg.rho * (x11 - x2).reshape(-1),
                y[2] + cfg.rho * (x11 - x3).reshape(-1)]


GROUPS = ("net", "netT", "netF")


def make_params(cfg: StepConfig, dtype=torch.float32, gain: float = 1.0):
    """Closed-form parameters for the four modules of src/kharmonic_lofar.py:60-65."""
    params = {
        "net": closed_form_state_dict(cfg.L, cfg.channels, 2, cfg.rica, "net", gain, dtype),
        "netT": closed_form_state_dict(cfg.Lt, cfg.channels, 1, cfg.rica, "netT", gain, dtype),
        "netF": closed_form_state_dict(cfg.Lt, cfg.channels, 1, cfg.rica, "netF", gain, dtype),
    }
    D = cfg.L + 2 * cfg.Lt
    M = (0.5 + 0.5 * closed_form((cfg.K, D), "khm:M", 1.0, 0.61803398875, dtype))
    return params, M


def flat_leaves(params, M, groups=("net", "netT", "netF", "mod")):
    leaves = []
    for g in GROUPS:
        if g in groups:
            leaves += [params[g][k] for k in params[g]]
    if "mod" in groups:
        leaves.append(M)
    return leaves


class AdamState:
    """torch.optim.Adam defaults (src/kharmonic_lofar.py:92): betas (0.9,0.999),
    eps 1e-8, no weight decay, bias-corrected."""

    def __init__(self, leaves: Iterable[Tensor], lr: float):
        self.leaves = list(leaves)
        self.m = [torch.zeros_like(t) for t in self.leaves]
        self.v = [torch.zeros_like(t) for t in self.leaves]
        self.t = 0
        self.lr = lr

    def step(self, grads: Sequence[Tensor]):
        self.t += 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        c1 = 1 - b1 ** self.t
        c2 = 1 - b2 ** self.t
        with torch.no_grad():
            for p, g, m, v in zip(self.leaves, grads, self.m, self.v):
                m.mul_(b1).add_(g, alpha=1 - b1)
                v.mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (v.sqrt() / math.sqrt(c2)).add_(eps)
                p.addcdiv_(m, denom, value=-self.lr / c1)


def admm_iteration(params, M, x, uv, y, cfg: StepConfig, adam: AdamState,
                   khm_fn=khm_loss, sim_fn=cluster_similarity, aug_fn=augmented_loss):
    """One `for admm in range(Nadmm)` body: closure + backward + Adam step
    (:185), no-grad forward and multiplier update (:187-202).
    Returns (terms, new_y, grads)."""
    for t in adam.leaves:
        t.requires_grad_(True)
        t.grad = None
    total, terms = closure_losses(params, M, x, uv, y, cfg, khm_fn, sim_fn, aug_fn)
    grads = torch.autograd.grad(total, adam.leaves, allow_unused=True)
    grads = [g if g is not None else torch.zeros_like(t) for g, t in zip(grads, adam.leaves)]
    for t in adam.leaves:
        t.requires_grad_(False)
    adam.step(grads)
    y_new = multiplier_update(params, x, uv, y, cfg)
    return [float(t.detach()) for t in terms], y_new, grads


# --------------------------------------------------------------------------
# dictionary learning X ~ A S  (src/rica_lofar.py:53-97), SURVEY 8 f4
# --------------------------------------------------------------------------
def rica_closed_form(L: int, M: int, B: int, dtype=torch.float32):
    """Deterministic stand-ins for the script's random draws: the minibatch X (L x B) (:69), the dictionary
    A = torch.rand((L, M)) (:51) and the codes S = torch.rand((M, B)) (:71); A and S in [0, 1) like rand."""
    X = closed_form((L, B), "rica.X", 1.0).to(dtype)
    A = (0.5 + 0.5 * closed_form((L, M), "rica.A", 1.0)).to(dtype)
    S = (0.5 + 0.5 * closed_form((M, B), "rica.S", 1.0)).to(dtype)
    return X, A, S


def rica_loss(X: Tensor, A: Tensor, S: Tensor, lambda1: float) -> Tensor:
    """The closure's loss (:76): MSELoss(reduction='sum')(X, A S) / (nbatch L) + lambda1 ||S||_1 / S.numel(),
    with torch.linalg.norm(S, 1) of a 2-D tensor = the matrix 1-norm (largest column sum of |S|)."""
    L, B = X.shape
    sse = ((X - A @ S) ** 2).sum()
    norm1 = S.abs().sum(dim=0).max()
    return sse / (B * L) + lambda1 * norm1 / S.numel()


def rica_loss_grad(X: Tensor, A: Tensor, S: Tensor, lambda1: float) -> Tuple[Tensor, Tensor]:
    """Closed-form gradient of rica_loss w.r.t. S: -2 A^T (X - A S) / (B L) plus lambda1 / numel * sign(S) on
    the arg-max column of the 1-norm (first one on ties)."""
    L, B = X.shape
    E = X - A @ S
    g = -2.0 * (A.t() @ E) / (B * L)
    col = int(torch.argmax(S.abs().sum(dim=0)))
    g[:, col] += lambda1 / S.numel() * torch.sign(S[:, col])
    return rica_loss(X, A, S, lambda1), g


def rica_dictionary_update(X: Tensor, A: Tensor, S: Tensor, eta: float) -> Tuple[Tensor, float]:
    """:84-93: E = X - A S; dA = sum_b outer(E[:, b], S[:, b]) / nbatch = E S^T / nbatch; A += eta dA.
    Returns (new A, ||dA||_F) -- the value the script prints."""
    B = X.shape[1]
    dA = (X - A @ S) @ S.t() / B
    return A + eta * dA, float(torch.linalg.norm(dA))
