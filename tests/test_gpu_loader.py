"""Minibatch patch pipeline vs the golden produced by the reference's own
get_data_minibatch on a synthetic SAP (tests/h5_fixture.py)."""
import numpy as np
import pytest
import torch

from tests.h5_fixture import drawn_baselines, make_sap
from tests.util import assert_close, assert_probe, load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nchan", [4, 8])
@pytest.mark.parametrize("normalize", [False, True])
def test_minibatch_matches_reference_loader(normalize, nchan):
    from lshm_amd.lofar_tools import minibatch_from_sap
    g = load_golden("minibatch")
    tag = ("norm" if normalize else "raw") + ("8" if nchan == 8 else "")
    sap, info = make_sap()
    sel = drawn_baselines(123, 1, sap["visibilities"].shape[0], 3)
    px, py, y, uv1 = minibatch_from_sap(sap, info, batch_size=3, patch_size=128, normalize_data=normalize,
                                        num_channels=nchan, uvdist=True, baselinelist=sel)
    assert [px, py] == list(g[f"{tag}/patchxy"])
    assert y.shape == (px * py * 3, nchan, 128, 128)
    assert_probe(g, f"{tag}/y", y, 2e-6, 1e-5)
    assert_close(y[0, :, 60:64, 60:64], g[f"{tag}/y_first"], 2e-6, 1e-5)
    assert_close(uv1, g[f"{tag}/uv"], 1e-6, 1e-4)
    if not normalize:
        assert float(y.abs().max()) <= 1000.0


def test_short_spectra_are_zero_padded():
    from lshm_amd.lofar_tools import patches_from_visibilities
    sap, _ = make_sap(nbase=2, ntime=70, nfreq=100)
    vis = torch.from_numpy(sap["visibilities"]).cuda()
    sc = torch.from_numpy(sap["visibility_scale_factors"]).cuda()
    px, py, y, ms = patches_from_visibilities(vis, sc, 128, False)
    assert (px, py) == (1, 1) and y.shape == (2, 4, 128, 128)
    assert float(y[:, :, 70:, :].abs().max()) == 0.0 and float(y[:, :, :, 100:].abs().max()) == 0.0
    ref = sap["visibilities"][0, :, :, 3, 1].astype(np.float32) * sap["visibility_scale_factors"][0, :, 3][None, :]
    np.testing.assert_allclose(y[0, 3, :70, :100].cpu().numpy(), np.clip(ref, -1e3, 1e3), rtol=1e-6)
    # baseline 1, frequency 5, pol 0 carries a scale of 40: |int8| * 40 exceeds the +-1e3 clamp
    raw = sap["visibilities"][1, :, 5, 0, 0].astype(np.float32) * 40.0
    assert np.abs(raw).max() > 1000.0
    np.testing.assert_allclose(y[1, 0, :70, 5].cpu().numpy(), np.clip(raw, -1e3, 1e3), rtol=1e-6)
    assert abs(ms[0].item() - y.double().mean().item()) < 1e-9
    assert abs(ms[1].item() - y.double().std().item()) < 1e-7


def test_global_batch_normalisation_from_rank_moments():
    """Data-parallel loader: each rank cuts the patches of ITS baselines without normalising and
    reports [sum, sum of squares, count]; after the three doubles are summed over ranks every rank normalises
    with the moments of the whole minibatch -- the same numbers the reference's single process gets from
    y.mean() / y.std() over all patches (src/lofar_tools.py:190-193)."""
    from lshm_amd import _lib as L
    lib = L.load()
    g = load_golden("minibatch")
    sap, info = make_sap()
    sel = drawn_baselines(123, 1, sap["visibilities"].shape[0], 3)
    vis = torch.from_numpy(np.stack([sap["visibilities"][int(b)] for b in sel])).cuda()
    sc = torch.from_numpy(np.stack([sap["visibility_scale_factors"][int(b)] for b in sel])).cuda()
    nb, ntime, nfreq = vis.shape[:3]
    ws = torch.empty(lib.lshm_patches_workspace_floats(), device="cuda")
    ms = torch.empty(2, device="cuda", dtype=torch.float64)
    parts, moms = [], []
    for lo, hi in ((0, 2), (2, 3)):     # "rank 0" holds two baselines, "rank 1" one
        v, s_ = vis[lo:hi].contiguous(), sc[lo:hi].contiguous()
        px = (max(ntime, 128) - 128) // 64 + 1
        py = (max(nfreq, 128) - 128) // 64 + 1
        y = torch.empty((px * py * (hi - lo), 4, 128, 128), device="cuda")
        mom = torch.empty(3, device="cuda", dtype=torch.float64)
        L.check(lib.lshm_patches_from_vis_ex(L.ptr(v), L.ptr(s_), hi - lo, ntime, nfreq, 128, 4, 1e3, 0, L.ptr(y),
                                             L.ptr(ms), L.ptr(mom), L.ptr(ws), L.stream()))
        parts.append(y)
        moms.append(mom)
    total = moms[0] + moms[1]          # what the SUM all-reduce leaves on every rank
    for y in parts:
        L.check(lib.lshm_patches_normalize(L.ptr(y), y.numel(), L.ptr(total), L.stream()))
    torch.cuda.synchronize()
    # patch-major order of the global minibatch: patch ck of baseline b sits at ck * nb + b
    npatch = parts[0].shape[0] // 2
    full = torch.empty((npatch * 3, 4, 128, 128), device="cuda")
    for ck in range(npatch):
        full[ck * 3:ck * 3 + 2] = parts[0][ck * 2:ck * 2 + 2]
        full[ck * 3 + 2] = parts[1][ck]
    assert_probe(g, "norm/y", full, 2e-6, 1e-5)
    assert_close(full[0, :, 60:64, 60:64], g["norm/y_first"], 2e-6, 1e-5)
