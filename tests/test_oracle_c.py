"""Cross-check of the two CPU restatements: the plain-C per-op oracle (oracle/lshm_oracle_c.c,
no PyTorch) against the torch-CPU oracle that is pinned to reference-generated goldens."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import lshm_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_build", "liblshm_oracle_c.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(SO):
        subprocess.check_call(["make", "oracle"], cwd=ROOT)
    l = C.CDLL(SO)
    l.oc_khm.restype = C.c_double
    l.oc_cluster_similarity.restype = C.c_double
    l.oc_augmented_loss.restype = C.c_double
    return l


def fp(t):
    return t.numpy().ctypes.data_as(C.c_void_p)


def test_conv_flavours(lib):
    B = 2
    x = O.closed_form((B, 4, 16, 16), "c:x", 1.0, 0.2113)
    w = O.closed_form((6, 4, 4, 4), "c:w", 0.2)
    b = O.closed_form((6,), "c:b", 0.1)
    y = torch.empty(B, 6, 8, 8)
    lib.oc_conv2d_k4s2p1(fp(x), fp(w), fp(b), fp(y), B, 4, 6, 16, 16, 1)
    assert torch.allclose(y, F.elu(F.conv2d(x, w, b, stride=2, padding=1)), atol=2e-6)
    wt = O.closed_form((4, 5, 4, 4), "c:wt", 0.2)
    bt = O.closed_form((5,), "c:bt", 0.1)
    yt = torch.empty(B, 5, 32, 32)
    lib.oc_tconv2d_k4s2p1(fp(x), fp(wt), fp(bt), fp(yt), B, 4, 5, 16, 16, 0)
    assert torch.allclose(yt, F.conv_transpose2d(x, wt, bt, stride=2, padding=1), atol=2e-6)
    x1 = O.closed_form((B, 4, 64), "c:x1", 1.0, 0.2113)
    w1 = O.closed_form((6, 4, 4), "c:w1", 0.3)
    y1 = torch.empty(B, 6, 16)
    lib.oc_conv1d_k4s4p1(fp(x1), fp(w1), fp(b), fp(y1), B, 4, 6, 64, 1)
    assert torch.allclose(y1, F.elu(F.conv1d(x1, w1, b, stride=4, padding=1)), atol=2e-6)
    wt1 = O.closed_form((4, 5, 4), "c:wt1", 0.3)
    yt1 = torch.empty(B, 5, 256)
    lib.oc_tconv1d_k4s4p0(fp(x1), fp(wt1), fp(bt), fp(yt1), B, 4, 5, 64, 1)
    assert torch.allclose(yt1, F.elu(F.conv_transpose1d(x1, wt1, bt, stride=4, padding=0)), atol=2e-6)


def test_linear_and_harmonics(lib):
    x = O.closed_form((3, 20), "l:x")
    w = O.closed_form((7, 20), "l:w", 0.3)
    b = O.closed_form((7,), "l:b", 0.1)
    y = torch.empty(3, 7)
    lib.oc_linear(fp(x), fp(w), fp(b), fp(y), 3, 20, 7, 1)
    assert torch.allclose(y, F.elu(F.linear(x, w, b)), atol=2e-6)
    uv = torch.tensor([[1234.5, -987.25], [0.0, 3.5], [-20000.0, 15.125]])
    sc = torch.tensor(O.DEFAULT_SCALES)
    out = torch.empty(3, 16)
    lib.oc_uv_harmonics(fp(uv), fp(sc), 4, 3, fp(out))
    assert torch.allclose(out, O.uv_harmonics(sc, uv), atol=1e-6)


@pytest.mark.parametrize("B,K,D,p", [(8, 4, 64, 4), (5, 10, 256, 2), (6, 3, 32, 3)])
def test_khm(lib, B, K, D, p):
    X = 0.8 * O.closed_form((B, D), "k:X", 1.0, 0.4142) + 0.3
    M = 0.5 + 0.5 * O.closed_form((K, D), "k:M", 1.0, 0.618)
    dX = np.zeros((B, D))
    dM = np.zeros((K, D))
    loss = lib.oc_khm(fp(X), fp(M), B, K, D, C.c_double(p), C.c_double(1e-9),
                      dX.ctypes.data_as(C.c_void_p), dM.ctypes.data_as(C.c_void_p))
    lo, dXo, dMo = O.khm_grads(X, M, p)
    assert abs(loss - lo.item()) <= 1e-10 * abs(lo.item())
    assert np.allclose(dX, dXo.numpy(), rtol=1e-9, atol=1e-14) and np.allclose(dM, dMo.numpy(), rtol=1e-9, atol=1e-14)
    assert abs(loss - O.khm_loss_loop(X, M, p).item()) <= 2e-6 * abs(loss)


def test_similarity_and_augmented(lib):
    M = 0.5 + 0.5 * O.closed_form((10, 256), "s:M", 1.0, 0.618)
    v = lib.oc_cluster_similarity(fp(M), 10, 256, C.c_double(1e-9))
    assert abs(v - O.cluster_similarity_loop(M).item()) <= 2e-6 * abs(v)
    for rows, bpb, bs in ((16, 4, 4), (20, 4, 3), (10, 4, 3)):
        Z = 0.8 * O.closed_form((rows, 64), "a:Z", 1.0, 0.4142) + 0.1
        v = lib.oc_augmented_loss(fp(Z), rows, 64, bpb, bs)
        assert abs(v - O.augmented_loss_loop(Z, bpb, bs).item()) <= 2e-6 * abs(v) + 1e-12
        assert abs(v - O.augmented_loss(Z, bpb, bs).item()) <= 2e-6 * abs(v) + 1e-12


def test_fft_features(lib):
    r = 3.0 * O.closed_form((1, 2, 128, 128), "f:x", 1.0, 0.0137)
    re = torch.empty(2, 128, 128)
    im = torch.empty(2, 128, 128)
    lib.oc_fft2_features(fp(r), fp(re), fp(im), 2, 128, C.c_double(10.0))
    ref = O.fft_features(r)
    assert torch.allclose(re, ref[0, :2], atol=2e-5) and torch.allclose(im, ref[0, 2:], atol=2e-5)


def test_c_oracle_is_clean_under_sanitizers():
    """`make sanitize`: every function of the C restatement on exactly-sized heap buffers under AddressSanitizer +
    UndefinedBehaviorSanitizer (oracle/sanitize_main.c).  CPU only -- the GPU pool offers no sanitizer and the HIP library
    has no CPU build; the checker itself must at least be free of out-of-bounds reads."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-s", "sanitize"], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "sanitize_main: ok" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr
