"""Minibatch patch pipeline vs the golden produced by the reference's own
get_data_minibatch on a synthetic SAP (tests/h5_fixture.py)."""
import numpy as np
import pytest
import torch

from tests.h5_fixture import drawn_baselines, make_sap
from tests.util import assert_close, assert_probe, load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("normalize", [False, True])
def test_minibatch_matches_reference_loader(normalize):
    from lshm_amd.lofar_tools import minibatch_from_sap
    g = load_golden("minibatch")
    tag = "norm" if normalize else "raw"
    sap, info = make_sap()
    sel = drawn_baselines(123, 1, sap["visibilities"].shape[0], 3)
    px, py, y, uv1 = minibatch_from_sap(sap, info, batch_size=3, patch_size=128, normalize_data=normalize,
                                        num_channels=4, uvdist=True, baselinelist=sel)
    assert [px, py] == list(g[f"{tag}/patchxy"])
    assert y.shape == (px * py * 3, 4, 128, 128)
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
