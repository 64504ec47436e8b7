"""Minibatch construction of the reference's ``src/lofar_tools.py`` (``get_data_minibatch`` :51-211) with
the tensor work on the GPU.

The H5 container is only *read* on the host (``h5py`` if installed); visibilities stay int8 until they
are on the device, where one kernel dequantises, zero-pads, cuts the 50 %-overlap patches in the
reference's patch-major order, clamps and accumulates the minibatch moments, and a second pass
normalises (``lshm_patches_from_vis``).  ``torch_fftshift`` and the u,v computation are kept with the
reference's signatures.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib as L

C_LIGHT = 2.99792458e8


def torch_fftshift(real, imag):
    """src/lofar_tools.py:24-30 (roll dims >= 2 by size//2); plain tensor indexing, any device."""
    for dim in range(2, real.dim()):
        real = torch.roll(real, dims=dim, shifts=real.size(dim) // 2)
        imag = torch.roll(imag, dims=dim, shifts=imag.size(dim) // 2)
    return real, imag


def patches_from_visibilities(vis: torch.Tensor, scale: torch.Tensor, patch_size: int = 128,
                              normalize_data: bool = False, clamp: float = 1e3, num_channels: int = 4,
                              process_group=None):
    """vis (nb, ntime, nfreq, 4, 2) int8 and scale (nb, nfreq, 4) fp32, both on the device ->
    (patchx, patchy, y (patchx*patchy*nb, num_channels, P, P), [mean, std] as a device tensor).
    process_group: data-parallel job whose ranks each hold some baselines of ONE minibatch -- the
    normalisation then uses the moments of the whole minibatch (sum, sum of squares and count are
    all-reduced), exactly what the single process of upstream computes at :190-193."""
    if not vis.is_cuda or vis.dtype != torch.int8:
        raise RuntimeError("vis must be an int8 tensor on a HIP device")
    L.require_device(scale)
    vis, scale = vis.contiguous(), scale.contiguous()
    nb, ntime, nfreq, npol, nc = vis.shape
    if npol != 4 or nc != 2 or tuple(scale.shape) != (nb, nfreq, 4):
        raise RuntimeError("expected vis (nb, ntime, nfreq, 4, 2) and scale (nb, nfreq, 4)")
    P = patch_size
    T, F = max(ntime, P), max(nfreq, P)
    px, py = (T - P) // (P // 2) + 1, (F - P) // (P // 2) + 1
    if num_channels not in (4, 8):
        raise AssertionError("num_channels==4 or num_channels==8")  # upstream :70
    lib = L.load()
    y = torch.empty((px * py * nb, num_channels, P, P), device=vis.device, dtype=torch.float32)
    ms = torch.empty(2, device=vis.device, dtype=torch.float64)
    mom = torch.empty(3, device=vis.device, dtype=torch.float64)
    ws = torch.empty(lib.lshm_patches_workspace_floats(), device=vis.device, dtype=torch.float32)
    world = 1
    if process_group is not None:
        import torch.distributed as dist
        world = dist.get_world_size(process_group)
    local_norm = bool(normalize_data) and world == 1
    with L.on_device(vis.device):
        L.check(lib.lshm_patches_from_vis_ex(L.ptr(vis), L.ptr(scale), nb, ntime, nfreq, P, num_channels, float(clamp),
                                             int(local_norm), L.ptr(y), L.ptr(ms), L.ptr(mom), L.ptr(ws), L.stream()),
                "patches_from_vis")
        if normalize_data and world > 1:
            import torch.distributed as dist
            dist.all_reduce(mom, op=dist.ReduceOp.SUM, group=process_group)   # [sum, sum of squares, count]
            L.check(lib.lshm_patches_normalize(L.ptr(y), y.numel(), L.ptr(mom), L.stream()), "patches_normalize")
            mean = mom[0] / mom[2]
            ms = torch.stack((mean, ((mom[1] - mom[2] * mean * mean) / (mom[2] - 1)).clamp_min(0).sqrt()))
    return px, py, y, ms


def uv_wavelengths(xyz: np.ndarray, baselines: np.ndarray, sel: np.ndarray, start_time: str, freq0: float):
    """u,v of the selected baselines in wavelengths: station XY difference rotated by the start-time
    hour angle, divided by lambda (src/lofar_tools.py:90-106,143-151).  start_time: 'date hh:mm:ss'."""
    hms = start_time.split()[1].split(sep=":")
    hours = float(hms[0]) + float(hms[1]) / 60.0 + float(hms[2]) / 3600
    theta = hours / 24.0 * (2 * math.pi)
    inv_lambda = freq0 / C_LIGHT
    rot00, rot01 = math.cos(theta) * inv_lambda, math.sin(theta) * inv_lambda
    uv = np.zeros((len(sel), 2), dtype=np.float32)
    for k, b in enumerate(sel):
        xx = xyz[baselines[b][0]][0] - xyz[baselines[b][1]][0]
        yy = xyz[baselines[b][0]][1] - xyz[baselines[b][1]][1]
        uv[k, 0] = xx * rot00 + yy * rot01
        uv[k, 1] = -xx * rot01 + yy * rot00
    return uv


def minibatch_from_sap(sap, info, batch_size=2, patch_size=32, normalize_data=False, num_channels=8,
                       uvdist=False, device="cuda", baselinelist=None, process_group=None):
    """get_data_minibatch on an already opened SAP group (h5py group or a dict of numpy arrays with the
    keys 'visibilities', 'visibility_scale_factors', 'central_frequencies', 'baselines',
    'antenna_locations'/'XYZ'; info['start_time'][0] bytes).  Baselines are drawn with
    np.random.randint exactly as upstream (:88) unless given.  Returns (patchx, patchy, y[, uv1])."""
    g, h = sap["visibilities"], sap["visibility_scale_factors"]
    nbase = g.shape[0]
    if baselinelist is None:
        baselinelist = np.random.randint(0, nbase, batch_size)
    baselinelist = np.asarray(baselinelist)
    vis = torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(g[int(b)]) for b in baselinelist]))).to(device)
    sc = torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(h[int(b)]) for b in baselinelist]),
                                               dtype=np.float32)).to(device)
    px, py, y, _ = patches_from_visibilities(vis, sc, patch_size, normalize_data, num_channels=num_channels,
                                             process_group=process_group)
    if not uvdist:
        return px, py, y
    frq = np.asarray(sap["central_frequencies"])
    uv = uv_wavelengths(np.asarray(sap["antenna_locations"]["XYZ"]), np.asarray(sap["baselines"]), baselinelist,
                        info["start_time"][0].decode("ascii"), float(frq[frq.shape[0] // 2]))
    # upstream repeats each baseline's uv for its px*py patches in BASELINE-major order (:175-178) while
    # the patches themselves are patch-major (:170-173); kept as is for parity
    uv1 = torch.from_numpy(np.repeat(uv, px * py, axis=0)).to(device)
    return px, py, y, uv1


def get_data_minibatch(file_list, SAP_list, batch_size=2, patch_size=32, normalize_data=False, num_channels=8,
                       transform=None, uvdist=False, device="cuda"):
    """Same call as upstream :51; needs h5py for the container (not part of the GPU path)."""
    try:
        import h5py
    except ImportError as e:  # pragma: no cover - h5py is absent in the build image
        raise ImportError("get_data_minibatch reads LOFAR H5 files and needs h5py; use minibatch_from_sap "
                          "with arrays otherwise") from e
    if transform is not None:
        raise NotImplementedError("torchvision transforms are not part of the GPU pipeline")
    assert len(file_list) == len(SAP_list)
    file_id = np.random.randint(0, len(file_list))
    f = h5py.File(file_list[file_id], "r")
    return minibatch_from_sap(f["measurement"]["saps"][SAP_list[file_id]], f["measurement"]["info"], batch_size,
                              patch_size, normalize_data, num_channels, uvdist, device)
