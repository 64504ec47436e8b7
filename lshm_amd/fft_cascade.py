"""The notebook-era second stage (``Demo.ipynb:116-119,169-175``; ``src/EvaluateClusters.ipynb:237-243``):
a 2-D autoencoder over the clamped, shifted 2-D FFT (real || imaginary, 2C channels) of the first
autoencoder's residual.  At v2 the training script uses the row/column 1-D autoencoders instead
(``src/kharmonic_lofar.py:137-147``); this keeps the FFT feature path available as
``feature_mode="fft"`` with the reference's module API.

    net  = AutoEncoderCNN2(latent_dim=L,  channels=4, harmonic_scales=hs, rica=...)
    fnet = AutoEncoderCNN2(latent_dim=Lf, channels=8, harmonic_scales=hs, rica=...)
    xhat, mu, Fx, Fhat, fmu = fft_cascade_forward(net, fnet, x, uv)
"""
from __future__ import annotations

import torch

from .functional import fft_features


def fft_cascade_forward(net, fnet, x: torch.Tensor, uv: torch.Tensor, clamp: float = 10.0,
                        detach_residual: bool = True):
    """(xhat, mu) = net(x, uv);  F = clamp(cat(Re, Im)(fftshift(fftn(x - xhat, ortho))));
    (Fhat, fmu) = fnet(F, uv).  The notebooks run the FFT step on detached residuals (the default here);
    with detach_residual=False the second stage is trainable end to end: the feature op has a backward
    (lshm_fft2_backward), so gradients of a loss on Fhat / fmu reach `net` through the residual."""
    xhat, mu = net(x, uv)
    resid = x - xhat
    if detach_residual:
        resid = resid.detach()
    Fx = fft_features(resid, clamp)
    Fhat, fmu = fnet(Fx, uv)
    return xhat, mu, Fx, Fhat, fmu


def fft_cascade_latents(net, fnet, x, uv, clamp: float = 10.0) -> torch.Tensor:
    """Concatenated latent [mu | fmu] per patch, the clustering input of the notebook pipeline."""
    with torch.no_grad():
        _, mu, _, _, fmu = fft_cascade_forward(net, fnet, x, uv, clamp)
    return torch.cat((mu, fmu), dim=1)
