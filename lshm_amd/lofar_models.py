"""Drop-in modules for the reference's ``src/lofar_models.py``.

Same class names, constructor signatures, ``forward`` contracts, attribute names
and ``state_dict`` keys / tensor layouts as upstream (``AutoEncoderCNN2`` :12-99,
``AutoEncoder1DCNN`` :103-184, ``Kmeans`` :189-261), so checkpoints written by
either side load in the other.  The arithmetic runs in hand-written gfx950
kernels (through ``lshm_amd.functional``); the ``torch.nn`` layer objects below
only hold parameters (and give the reference's default initialisation, drawn in
the reference's order, so a given ``torch.manual_seed`` yields the same weights).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as Fh

_CHANNELS = (8, 12, 24, 48, 96, 192)  # widths after each encoder stage (upstream :31-41)
_FLAT = 768                           # 192*2*2 == 192*4 (upstream :45-46, :129-130)


class _CascadeAE(nn.Module):
    """Shared body of the 2-D and 1-D autoencoders (they differ only in the conv flavour)."""

    _ndim = 2
    # extension: "bf16" rounds the operands of this module's GEMM-shaped layers to bf16 (fp32 accumulation and
    # storage); a per-module attribute, passed per call to the kernels -- nothing process-wide
    matrix_precision = "fp32"

    def __init__(self, latent_dim=128, channels=3, harmonic_scales=None, rica=False):
        super().__init__()
        self.rica = rica
        self.latent_dim = latent_dim
        self.harmonic_scales = harmonic_scales
        # as upstream (:29): a missing scale vector is an AttributeError at construction
        self.harmonic_dim = (self.harmonic_scales.size()[0]) * 2 * 2
        widths = (channels,) + _CHANNELS
        if self._ndim == 2:
            conv = lambda i, o: nn.Conv2d(i, o, 4, stride=2, padding=1)
            tconv = lambda i, o: nn.ConvTranspose2d(i, o, 4, stride=2, padding=1)
            self._kinds = (Fh.CONV2D, Fh.TCONV2D)
            self._bottleneck = (192, 2, 2)
        else:
            conv = lambda i, o: nn.Conv1d(i, o, 4, stride=4, padding=1)
            tconv = lambda i, o: nn.ConvTranspose1d(i, o, 4, stride=4, padding=0, output_padding=0)
            self._kinds = (Fh.CONV1D, Fh.TCONV1D)
            self._bottleneck = (192, 4)
        for i in range(6):
            setattr(self, f"conv{i}", conv(widths[i], widths[i + 1]))
        self.fcuv1 = nn.Linear(self.harmonic_dim, self.harmonic_dim)
        self.fcuv3 = nn.Linear(self.harmonic_dim, self.harmonic_dim)
        self.fc1 = nn.Linear(_FLAT + self.harmonic_dim, self.latent_dim)
        if self.rica:
            self.fc2in = nn.Linear(self.latent_dim, self.latent_dim)
            self.fc2out = nn.Linear(self.latent_dim, self.latent_dim)
        self.fc3 = nn.Linear(self.latent_dim + self.harmonic_dim, _FLAT)
        for i in range(6):
            setattr(self, f"tconv{i}", tconv(widths[6 - i], widths[5 - i]))

    # -- helpers ---------------------------------------------------------
    def _lin(self, layer, x, act=True):
        return Fh.linear_act(x, layer.weight, layer.bias, act, self.matrix_precision == "bf16")

    def _harmonics(self, uv):
        return Fh.uv_harmonics(self.harmonic_scales, uv)

    # -- reference API ---------------------------------------------------
    def forward(self, x, uv):
        uvh = self._harmonics(uv)
        mu = self.encode(x, uvh)
        if not self.rica:
            # upstream's 1-D class drops `uv` here (:150) and cannot run; the intent is obvious
            return self.decode(mu, uvh), mu
        mu = self._lin(self.fc2in, mu)
        return self.decode(self._lin(self.fc2out, mu), uvh), mu

    def encode(self, x, uv):
        h = x
        for i in range(6):
            layer = getattr(self, f"conv{i}")
            h = Fh.conv_act(h, layer.weight, layer.bias, self._kinds[0], True, self.matrix_precision == "bf16")
        h = torch.flatten(h, start_dim=1)
        u = self._lin(self.fcuv1, uv)
        return self._lin(self.fc1, torch.cat((h, u), dim=1))

    def decode(self, z, uv):
        u = self._lin(self.fcuv3, uv)
        h = self._lin(self.fc3, torch.cat((z, u), dim=1), act=False)
        h = torch.reshape(h, (-1,) + self._bottleneck)
        for i in range(6):
            layer = getattr(self, f"tconv{i}")
            h = Fh.conv_act(h, layer.weight, layer.bias, self._kinds[1], i < 5, self.matrix_precision == "bf16")
        return h


class AutoEncoderCNN2(_CascadeAE):
    """2-D conv autoencoder over (B,C,128,128) patches: forward(x, uv) -> (xhat, mu)."""
    _ndim = 2


# BASELINE.json's north star calls the class AutoEncoderCNN; upstream v2 only has AutoEncoderCNN2.
AutoEncoderCNN = AutoEncoderCNN2


class AutoEncoder1DCNN(_CascadeAE):
    """1-D conv autoencoder over (B,C,16384) vectorised patches: forward(x, uv) -> (xhat, mu)."""
    _ndim = 1


class Kmeans(nn.Module):
    """K-harmonic-means clustering module (upstream :189-261)."""

    def __init__(self, latent_dim=128, K=10, p=2):
        super().__init__()
        self.latent_dim = latent_dim
        self.K = K
        self.p = p
        self.EPS = 1e-9
        self.M = torch.nn.Parameter(torch.rand(self.K, self.latent_dim), requires_grad=True)

    def forward(self, X):
        return Fh.khm_loss(X, self.M, self.p, self.EPS)

    def clustering_error(self, X):
        return self.forward(X)

    def cluster_similarity(self):
        return Fh.cluster_similarity(self.M, self.EPS)

    def offline_partials(self, X):
        """(numerator (K,D), denominator (K)) of the centroid recursion: the pair a
        data-parallel job sums over ranks before dividing."""
        return Fh.khm_offline_partials(X, self.M, self.p, self.EPS)

    def offline_update(self, X):
        """Zhang's generalised K-harmonic-means recursion, eqs 7.1-7.5.  Upstream's body
        (:231-261) cannot run (typo at :248, in-place writes into a leaf that requires
        grad); this implements what it states."""
        num, den = self.offline_partials(X)
        with torch.no_grad():
            self.M.copy_(num / den[:, None])

    def mean_distances(self, X):
        """dist[k] = mean_n ||X_n - M_k||^p, the per-baseline statistic of
        src/evaluate_clustering.py:111-115."""
        return Fh.khm_mean_distances(X, self.M, self.p)

    def assign(self, X):
        """(dist (K,), cluster id, soft labels) of the patches X of ONE baseline: dist as mean_distances,
        id = argmin_k dist (src/evaluate_clustering.py:111-119), labels = softmax(-dist / dist.mean())
        (src/train_graph_stat.py:206-210)."""
        dist = self.mean_distances(X)
        idx, prob = Fh.khm_assign(dist)
        return dist, idx, prob
