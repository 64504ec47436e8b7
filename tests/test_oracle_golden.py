"""Pins oracle/lshm_oracle.py against fixtures produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import lshm_oracle as O
from tests.util import assert_close, assert_probe, load_golden

SC = torch.tensor(O.DEFAULT_SCALES)


def test_harmonics():
    g = load_golden("harmonics")
    out = O.uv_harmonics(SC, torch.from_numpy(g["harmonics/uv"]))
    assert_close(out, g["harmonics/out"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("name,L,C,nd,rica", [("ae2d_rica", 224, 4, 2, True),
                                              ("ae2d_norica", 224, 4, 2, False),
                                              ("ae1d_rica", 16, 4, 1, True),
                                              ("fnet8", 64, 8, 2, False)])
def test_autoencoders(name, L, C, nd, rica):
    g = load_golden("autoencoders")
    sd = O.closed_form_state_dict(L, C, nd, rica, name)
    for t in sd.values():
        t.requires_grad_(True)
    x, uv = O.closed_form_inputs(2, C)
    if nd == 1:
        x = x.flatten(2, 3)
    x.requires_grad_(True)
    xhat, mu = O.ae_forward(sd, x, uv, SC, nd, rica)
    assert_probe(g, name + "/xhat", xhat, 1e-5, 1e-6)
    assert_close(mu, g[name + "/mu"], 1e-5, 1e-6)
    gy = O.closed_form(tuple(xhat.shape), name + ":gy", 1.0, 0.3331)
    gm = O.closed_form(tuple(mu.shape), name + ":gm", 1.0, 0.7717)
    ((xhat * gy).sum() + (mu * gm).sum()).backward()
    assert_probe(g, name + "/dx", x.grad, 1e-4, 1e-6)
    for k, t in sd.items():
        ref = float(g[f"{name}/gnorm/{k}"])
        assert abs(t.grad.double().norm().item() - ref) <= 1e-4 * ref + 1e-7, k
        assert_close(t.grad.reshape(-1)[:8], g[f"{name}/ghead/{k}"], 1e-4, 1e-5 * ref + 1e-7, k)


def test_khm_loop_closed_and_grads():
    g = load_golden("latent_losses")
    for ci in range(6):
        Bk, K, D, p, deg = [int(v) for v in g[f"khm{ci}/cfg"]]
        M = 0.5 + 0.5 * O.closed_form((K, D), f"khm{ci}:M", 1.0, 0.618)
        X = 0.8 * O.closed_form((Bk, D), f"khm{ci}:X", 1.0, 0.4142) + 0.3
        if deg:
            X[0] = M[0]
        ref = float(g[f"khm{ci}/loss"])
        if Bk * K <= 640:
            assert abs(O.khm_loss_loop(X, M, p).item() - ref) <= 1e-6 * abs(ref)
        assert abs(O.khm_loss(X, M, p).item() - ref) <= 2e-6 * abs(ref)
        loss, dX, dM = O.khm_grads(X, M, p)
        assert abs(loss.item() - ref) <= 2e-6 * abs(ref)
        if not deg:
            assert_close(dX, g[f"khm{ci}/dX"], 2e-5, 1e-6 * np.abs(g[f"khm{ci}/dX"]).max())
            assert_close(dM, g[f"khm{ci}/dM"], 2e-5, 1e-6 * np.abs(g[f"khm{ci}/dM"]).max())
        else:
            assert torch.isfinite(dX).all() and torch.isfinite(dM).all()


def test_cluster_similarity():
    g = load_golden("latent_losses")
    for K in (4, 10, 64):
        M = (0.5 + 0.5 * O.closed_form((K, 256), f"sim{K}:M", 1.0, 0.618)).requires_grad_(True)
        ref = float(g[f"sim{K}/loss"])
        s = O.cluster_similarity(M)
        assert abs(s.item() - ref) <= 1e-6 * abs(ref)
        if K <= 10:
            assert abs(O.cluster_similarity_loop(M).item() - ref) <= 1e-6 * abs(ref)
        s.backward()
        assert_close(M.grad, g[f"sim{K}/dM"], 1e-4, 1e-9)


def test_augmented_loss():
    g = load_golden("latent_losses")
    for bpb, bs in ((4, 2), (8, 8), (9, 12)):
        Z = (0.8 * O.closed_form((bpb * bs, 256), f"aug{bpb}_{bs}:Z", 1.0, 0.4142) + 0.1)
        Z.requires_grad_(True)
        ref = float(g[f"aug{bpb}_{bs}/loss"])
        a = O.augmented_loss(Z, bpb, bs)
        assert a.shape == (1,)
        assert abs(a.item() - ref) <= 2e-6 * abs(ref)
        a.sum().backward()
        assert_close(Z.grad, g[f"aug{bpb}_{bs}/dZ"], 1e-4, 1e-9)
        assert abs(O.augmented_loss_loop(Z.detach(), bpb, bs).item() - ref) <= 1e-6 * abs(ref)


def test_fft_features():
    g = load_golden("fft")
    r, _ = O.closed_form_inputs(2, 4)
    out = O.fft_features(r * 3.0)
    assert out.shape == (2, 8, 128, 128)
    assert_probe(g, "fft/out", out, 1e-5, 1e-5)
    assert_close(out[:, :, 60:68, 60:68], g["fft/crop"], 1e-5, 1e-5)


def test_closure_and_adam_trajectory():
    g = load_golden("closure")
    cfg = O.StepConfig(K=4, bpb=4, batch_size=2)
    params, M = O.make_params(cfg)
    x, uv = O.closed_form_inputs(8, 4)
    y = [torch.zeros(x.numel()) for _ in range(3)]
    leaves = O.flat_leaves(params, M)
    names = [f"{gname}.{k}" for gname in O.GROUPS for k in params[gname]] + ["mod.M"]
    adam = O.AdamState(leaves, cfg.lr)
    terms_ref = g["terms"]
    for it in range(4):
        if it < 3:
            terms, y_new, grads = O.admm_iteration(params, M, x, uv, y, cfg, adam)
        else:
            for t in leaves:
                t.requires_grad_(True)
            total, tt = O.closure_losses(params, M, x, uv, y, cfg)
            grads = torch.autograd.grad(total, leaves)
            terms = [float(v) for v in tt]
        np.testing.assert_allclose(terms, terms_ref[it][:8], rtol=1e-4 if it == 0 else 1e-3,
                                   atol=1e-7)
        if it in (0, 3):
            tol = 1e-4 if it == 0 else 2e-3
            np.testing.assert_allclose([t.double().norm().item() for t in y], g[f"it{it}/ynorm"],
                                       rtol=tol)
            for nm, gr in zip(names, grads):
                ref = float(g[f"it{it}/gnorm/{nm}"])
                assert abs(gr.double().norm().item() - ref) <= tol * ref + 1e-9, (it, nm)
        if it < 3:
            y = y_new
