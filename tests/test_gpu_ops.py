"""GPU parity tests, per op: HIP kernels (through the C ABI) vs golden vectors produced by the
reference and vs the CPU oracle on identical inputs.  fp32 tolerances are written per test."""
import numpy as np
import pytest
import torch

from oracle import lshm_oracle as O
from tests.util import assert_close, assert_probe, load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
KINDS = {"conv2d": 0, "tconv2d": 1, "conv1d": 2, "tconv1d": 3}


def _F():
    from lshm_amd import functional as Fh
    return Fh


def test_library_loaded():
    from lshm_amd import _lib
    assert _lib.load().lshm_version() >= 100


def test_harmonics():
    g = load_golden("harmonics")
    out = _F().uv_harmonics(torch.tensor(O.DEFAULT_SCALES, device=DEV), torch.from_numpy(g["harmonics/uv"]).to(DEV))
    # device sin/cos vs libm at arguments up to 2000 rad
    assert_close(out, g["harmonics/out"], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("kind", list(KINDS))
@pytest.mark.parametrize("i", range(6))
def test_conv_layer_fwd_bwd_vs_golden(kind, i):
    _check_conv_layer(kind, i)


@pytest.mark.parametrize("cfg", range(22))
def test_every_gemm_tile_configuration(cfg):
    """The implicit-GEMM launcher chooses among 22 (M tile, K chunk, split-K / K-over-wavefronts) configurations; pin each
    one in turn so that whichever the tuner picks on a given box has been checked against the goldens."""
    from lshm_amd import _lib
    lib = _lib.load()
    lib.lshm_set_tuning(0, cfg)
    try:
        for kind in KINDS:
            for i in (1, 3, 5):
                _check_conv_layer(kind, i)
        test_linear_vs_torch(5, 784, 224, True)
        test_linear_vs_torch(64, 32, 768, False)
    finally:
        lib.lshm_set_tuning(0, -1)


def _check_conv_layer(kind, i):
    g = load_golden("convops")
    Fh = _F()
    ch = O.CH
    B = 2
    nd = 2 if kind.endswith("2d") else 1
    tr = kind.startswith("t")
    if nd == 2:
        hw = 128 >> i if not tr else 2 << i
        ishape = (B, ch[i] if not tr else ch[6 - i], hw, hw)
    else:
        Ls = [16384, 4096, 1024, 256, 64, 16, 4]
        ishape = (B, ch[i] if not tr else ch[6 - i], Ls[i] if not tr else Ls[6 - i])
    cin = ishape[1]
    cout = ch[i + 1] if not tr else ch[5 - i]
    kk = (4, 4) if nd == 2 else (4,)
    wshape = ((cout, cin) if not tr else (cin, cout)) + kk
    key = f"{kind}{i}"
    fan = (cin if not tr else cout) * int(np.prod(kk))
    x = O.closed_form(ishape, key + ":x", 1.0, 0.2113).to(DEV).requires_grad_(True)
    w = O.closed_form(wshape, key + ":w", (3.0 / fan) ** 0.5).to(DEV).requires_grad_(True)
    b = O.closed_form((cout,), key + ":b", fan ** -0.5).to(DEV).requires_grad_(True)
    y = Fh.conv_act(x, w, b, KINDS[kind], True)
    gy = O.closed_form(tuple(y.shape), key + ":gy", 1.0, 0.3331).to(DEV)
    (y * gy).sum().backward()
    assert_probe(g, key + "/y", y, 1e-5, 1e-6)
    assert_probe(g, key + "/dx", x.grad, 1e-4, 1e-6)
    assert_probe(g, key + "/dw", w.grad, 1e-4, 1e-4)
    ref_db = g[key + "/db"]
    assert_close(b.grad, ref_db, 1e-4, 1e-5 * np.abs(ref_db).max() + 1e-5)


@pytest.mark.parametrize("B,K,N,act", [(2, 16, 16, True), (5, 784, 224, True), (3, 240, 768, False), (64, 32, 768, False)])
def test_linear_vs_torch(B, K, N, act):
    Fh = _F()
    x = O.closed_form((B, K), "lin:x", 1.0, 0.31).to(DEV).requires_grad_(True)
    w = O.closed_form((N, K), "lin:w", K ** -0.5, 0.77).to(DEV).requires_grad_(True)
    b = O.closed_form((N,), "lin:b", 0.1, 0.53).to(DEV).requires_grad_(True)
    gy = O.closed_form((B, N), "lin:gy", 1.0, 0.91).to(DEV)
    y = Fh.linear_act(x, w, b, act)
    (y * gy).sum().backward()
    xr, wr, br = (t.detach().cpu().double().requires_grad_(True) for t in (x, w, b))
    yr = torch.nn.functional.linear(xr, wr, br)
    if act:
        yr = torch.nn.functional.elu(yr)
    (yr * gy.cpu().double()).sum().backward()
    # fp32 sums of up to 784 terms against fp64: a few 1e-6 whatever the association order of the tile
    # configuration (sequential chain, split-K slabs, K over the wavefronts)
    assert rel_err(y, yr) < 4e-6
    assert rel_err(x.grad, xr.grad) < 1e-5
    assert rel_err(w.grad, wr.grad) < 1e-5
    assert rel_err(b.grad, br.grad) < 1e-5


@pytest.mark.parametrize("name,L,C,nd,rica", [("ae2d_rica", 224, 4, 2, True), ("ae2d_norica", 224, 4, 2, False),
                                              ("ae1d_rica", 16, 4, 1, True), ("fnet8", 64, 8, 2, False)])
def test_autoencoder_modules_vs_golden(name, L, C, nd, rica):
    from lshm_amd.lofar_models import AutoEncoder1DCNN, AutoEncoderCNN2
    g = load_golden("autoencoders")
    cls = AutoEncoderCNN2 if nd == 2 else AutoEncoder1DCNN
    net = cls(latent_dim=L, channels=C, harmonic_scales=torch.tensor(O.DEFAULT_SCALES), rica=rica)
    net.load_state_dict(O.closed_form_state_dict(L, C, nd, rica, name))
    net = net.to(DEV)
    net.harmonic_scales = net.harmonic_scales.to(DEV)
    x, uv = O.closed_form_inputs(2, C)
    if nd == 1:
        x = x.flatten(2, 3)
    x = x.to(DEV).requires_grad_(True)
    xhat, mu = net(x, uv.to(DEV))
    assert_probe(g, name + "/xhat", xhat, 2e-5, 2e-6)
    assert_close(mu, g[name + "/mu"], 2e-5, 2e-6)
    gy = O.closed_form(tuple(xhat.shape), name + ":gy", 1.0, 0.3331).to(DEV)
    gm = O.closed_form(tuple(mu.shape), name + ":gm", 1.0, 0.7717).to(DEV)
    ((xhat * gy).sum() + (mu * gm).sum()).backward()
    assert_probe(g, name + "/dx", x.grad, 2e-4, 2e-6)
    # Parameter gradients.  Bias / first-layer gradients are sums of ~1e5 terms that cancel to
    # ~1e-3 of their magnitude, so fp32 summation order alone moves them by ~1e-3 relative (the
    # reference's own fp32 result is that far from exact).  Criterion: (a) golden norms within
    # 5e-3 plus the golden's own distance from the fp64 norm; (b) every element within "as accurate as the fp32 reference": the error against an
    # fp64 oracle is within an order of magnitude (10x) of the fp32 oracle's own error (+1e-5 of the tensor norm);
    # sequential fmaf chains (MFMA) carry ~sqrt(K) eps vs the blocked sums of the CPU kernels.
    sd64 = {k: v.double().requires_grad_(True) for k, v in O.closed_form_state_dict(L, C, nd, rica, name).items()}
    sd32 = {k: v.requires_grad_(True) for k, v in O.closed_form_state_dict(L, C, nd, rica, name).items()}
    xc, uvc = O.closed_form_inputs(2, C)
    if nd == 1:
        xc = xc.flatten(2, 3)
    for sdx, dt in ((sd64, torch.float64), (sd32, torch.float32)):
        xo, mo = O.ae_forward(sdx, xc.to(dt), uvc.to(dt), torch.tensor(O.DEFAULT_SCALES, dtype=dt), nd, rica)
        ((xo * gy.cpu().to(dt)).sum() + (mo * gm.cpu().to(dt)).sum()).backward()
    for k, p in net.named_parameters():
        ref = float(g[f"{name}/gnorm/{k}"])
        exact = sd64[k].grad
        ref_off = abs(ref - exact.norm().item())
        assert abs(p.grad.double().norm().item() - ref) <= 5e-3 * ref + 3 * ref_off + 1e-7, k
        err_ref = (sd32[k].grad.double() - exact).norm().item()
        err = (p.grad.double().cpu() - exact).norm().item()
        assert err <= 10 * err_ref + 1e-5 * exact.norm().item(), (k, err, err_ref)


def test_autoencoder_rejects_wrong_patch_size():
    from lshm_amd.lofar_models import AutoEncoderCNN2
    net = AutoEncoderCNN2(224, 4, torch.tensor(O.DEFAULT_SCALES, device=DEV), True).to(DEV)
    with pytest.raises(RuntimeError):  # upstream: mat1 and mat2 shapes cannot be multiplied
        net(torch.zeros(2, 4, 64, 64, device=DEV), torch.zeros(2, 2, device=DEV))


def test_khm_vs_golden_and_oracle():
    Fh = _F()
    g = load_golden("latent_losses")
    for ci in range(6):
        Bk, K, D, p, deg = [int(v) for v in g[f"khm{ci}/cfg"]]
        M = (0.5 + 0.5 * O.closed_form((K, D), f"khm{ci}:M", 1.0, 0.618))
        X = 0.8 * O.closed_form((Bk, D), f"khm{ci}:X", 1.0, 0.4142) + 0.3
        if deg:
            X[0] = M[0]
        Xg, Mg = X.to(DEV).requires_grad_(True), M.to(DEV).requires_grad_(True)
        loss = Fh.khm_loss(Xg, Mg, p)
        loss.backward()
        ref = float(g[f"khm{ci}/loss"])
        assert abs(loss.item() - ref) <= 3e-6 * abs(ref), (ci, loss.item(), ref)
        _, dXo, dMo = O.khm_grads(X, M, p)
        assert rel_err(Xg.grad, dXo) < 1e-5, ci
        assert rel_err(Mg.grad, dMo) < 1e-5, ci
        if not deg:
            assert_close(Xg.grad, g[f"khm{ci}/dX"], 3e-5, 2e-6 * np.abs(g[f"khm{ci}/dX"]).max())
            assert_close(Mg.grad, g[f"khm{ci}/dM"], 3e-5, 2e-6 * np.abs(g[f"khm{ci}/dM"]).max())


def test_khm_known_answer_equal_distances():
    # all d_ik = r  =>  loss = K/(K/(r^p+eps)+eps) * B / (B*K*D)   
    Fh = _F()
    B, K, D, p, r = 8, 4, 64, 4, 2.0
    X = torch.zeros(B, D)
    M = torch.zeros(K, D)
    for k in range(K):
        M[k, k] = r  # |x - m_k| = r for x = 0
    loss = Fh.khm_loss(X.to(DEV), M.to(DEV), p).item()
    eps = 1e-9
    ref = (K / (K / (r ** p + eps) + eps)) * B / (B * K * D)
    assert abs(loss - ref) <= 1e-6 * ref


def test_khm_ragged_sizes_and_offline_partials():
    Fh = _F()
    # K > 16 takes the row-split kernel: one, two, four and eight wavefronts per row (D <= 64, 128, 256, 512)
    for (N, K, D, p) in [(1, 1, 8, 2), (7, 3, 100, 4), (130, 17, 256, 4), (33, 64, 256, 3), (1000, 10, 256, 4),
                         (9, 20, 40, 4), (700, 33, 100, 2), (5, 64, 300, 4), (21, 17, 512, 4)]:
        X = 0.8 * O.closed_form((N, D), "rag:X", 1.0, 0.4142) + 0.3
        M = 0.5 + 0.5 * O.closed_form((K, D), "rag:M", 1.0, 0.618)
        Xg, Mg = X.to(DEV).requires_grad_(True), M.to(DEV).requires_grad_(True)
        loss = Fh.khm_loss(Xg, Mg, p)
        loss.backward()
        lo, dXo, dMo = O.khm_grads(X, M, p)
        assert abs(loss.item() - lo.item()) <= 5e-6 * abs(lo.item()), (N, K, D, p)
        assert rel_err(Xg.grad, dXo) < 2e-5 and rel_err(Mg.grad, dMo) < 2e-5, (N, K, D, p)
        num, den = Fh.khm_offline_partials(X.to(DEV), M.to(DEV), p)
        no, do = O.khm_offline_partials(X, M, p)
        assert rel_err(num, no) < 2e-5 and rel_err(den, do) < 2e-5, (N, K, D, p)
        dist = Fh.khm_mean_distances(X.to(DEV), M.to(DEV), p)
        dref = ((X[:, None, :].double() - M[None].double()) ** 2).sum(-1).pow(p / 2).mean(0)
        assert rel_err(dist, dref) < 1e-5


@pytest.mark.parametrize("K", [1, 4, 10, 37, 64])
def test_distance_epilogues_argmin_and_softmax(K):
    """Cluster id = torch.min over the mean distances (src/evaluate_clustering.py:111-119) and the soft labels
    softmax(-dist / dist.mean()) (src/train_graph_stat.py:206-210), written here with upstream's own torch
    expressions on the CPU."""
    from lshm_amd.lofar_models import Kmeans
    D, p, N = 256, 4, 9
    X = 0.8 * O.closed_form((N, D), "asg:X", 1.0, 0.4142) + 0.3
    M = 0.5 + 0.5 * O.closed_form((K, D), f"asg:M{K}", 1.0, 0.618)
    mod = Kmeans(latent_dim=D, K=K, p=p)
    mod.load_state_dict({"M": M})
    mod = mod.to(DEV)
    dist, idx, prob = mod.assign(X.to(DEV))
    dref = torch.zeros(K, dtype=torch.float64)
    for ck in range(K):
        for cn in range(N):
            dref[ck] = dref[ck] + torch.sum(torch.pow(torch.linalg.norm(X[cn, :].double() - M[ck, :].double(), 2), p))
    dref = dref / N
    (values, indices) = torch.min(dref.view(K, 1), 0)
    assert rel_err(dist, dref) < 1e-5
    assert int(idx) == int(indices[0])
    assert rel_err(prob, torch.softmax(-dref / dref.mean(), 0)) < 1e-5 and abs(float(prob.sum()) - 1.0) < 1e-6
    # ties: the first index wins, as torch.min does
    Fh = _F()
    tie = torch.full((K,), 2.0, device=DEV)
    i2, p2 = Fh.khm_assign(tie)
    assert int(i2) == 0 and rel_err(p2, torch.full((K,), 1.0 / K)) < 1e-6


@pytest.mark.parametrize("N,K,p", [(1, 17, 4), (64, 64, 4), (65, 33, 2), (1000, 48, 4), (257, 64, 3)])
def test_khm_matrix_core_kernel_vs_fp64(N, K, p):
    """16 < K <= 64 at latent_dim 256 runs on the matrix cores with the squared distance expanded as
    |x|^2 + |m|^2 - 2 x.m (lshm_amd/csrc/khm.hip: khm_mfma_kernel).  On separated data (every distance well above
    rounding of the norms) loss, both gradients, the offline partial sums and the evaluation distances match the fp64
    formula to the same 2e-5 as the exact-difference kernels; the stated loss of accuracy is for points that coincide
    with a centroid: there the expansion's absolute error ~1e-7 (|x|^2 + |m|^2) shows, and the test bounds it."""
    Fh = _F()
    D = 256
    X = 0.8 * O.closed_form((N, D), f"mm:X{N}", 1.0, 0.4142) + 0.3
    M = 0.5 + 0.5 * O.closed_form((K, D), f"mm:M{K}", 1.0, 0.618)
    Xg, Mg = X.to(DEV).requires_grad_(True), M.to(DEV).requires_grad_(True)
    loss = Fh.khm_loss(Xg, Mg, p)
    loss.backward()
    lo, dXo, dMo = O.khm_grads(X.double(), M.double(), p)
    assert abs(loss.item() - lo.item()) <= 1e-5 * abs(lo.item())
    assert rel_err(Xg.grad, dXo) < 2e-5 and rel_err(Mg.grad, dMo) < 2e-5
    num, den = Fh.khm_offline_partials(X.to(DEV), M.to(DEV), p)
    no, do = O.khm_offline_partials(X.double(), M.double(), p)
    assert rel_err(num, no) < 2e-5 and rel_err(den, do) < 2e-5
    dist = Fh.khm_mean_distances(X.to(DEV), M.to(DEV), p)
    dref = ((X[:, None, :].double() - M[None].double()) ** 2).sum(-1).pow(p / 2).mean(0)
    assert rel_err(dist, dref) < 1e-5
    # a row that IS a centroid: squared distance 0 up to the cancellation error, everything stays finite
    X2 = X.clone()
    X2[0] = M[1]
    d2 = Fh.khm_mean_distances(X2[:1].to(DEV), M.to(DEV), 2)   # p = 2: the squared distance itself
    assert abs(float(d2[1])) <= 2e-6 * float((M[1].double() ** 2).sum() * 2)   # a few ulp of |x|^2 + |m|^2 (measured 4e-7 x)
    Xc, Mc = X2.to(DEV).requires_grad_(True), M.to(DEV).requires_grad_(True)
    Fh.khm_loss(Xc, Mc, p).backward()
    assert torch.isfinite(Xc.grad).all() and torch.isfinite(Mc.grad).all()


def test_kmeans_module_offline_update_and_state_dict():
    from lshm_amd.lofar_models import Kmeans
    mod = Kmeans(latent_dim=256, K=10, p=4)
    assert list(mod.state_dict().keys()) == ["M"]
    M0 = 0.5 + 0.5 * O.closed_form((10, 256), "off:M", 1.0, 0.618)
    X = 0.8 * O.closed_form((64, 256), "off:X", 1.0, 0.4142) + 0.3
    mod.load_state_dict({"M": M0})
    mod = mod.to(DEV)
    mod.offline_update(X.to(DEV))
    assert rel_err(mod.M, O.khm_offline_update(X, M0, 4)) < 2e-5


def test_cluster_similarity_vs_golden():
    Fh = _F()
    g = load_golden("latent_losses")
    for K in (4, 10, 64):
        M = (0.5 + 0.5 * O.closed_form((K, 256), f"sim{K}:M", 1.0, 0.618)).to(DEV).requires_grad_(True)
        s = Fh.cluster_similarity(M)
        s.backward()
        ref = float(g[f"sim{K}/loss"])
        assert abs(s.item() - ref) <= 3e-6 * abs(ref)
        assert_close(M.grad, g[f"sim{K}/dM"], 2e-4, 2e-5 * np.abs(g[f"sim{K}/dM"]).max())


def test_cluster_similarity_known_answers():
    Fh = _F()
    K, D = 6, 32
    M = torch.zeros(K, D)
    for k in range(K):
        M[k, k] = 1.0 + k  # orthogonal rows: (K-1) e^0 / (e^1+eps) * K / (K D)
    val = Fh.cluster_similarity(M.to(DEV)).item()
    ref = (K - 1) / (np.e + 1e-9) * K / (K * D)
    assert abs(val - ref) <= 2e-6 * ref


def test_augmented_loss_vs_golden():
    Fh = _F()
    g = load_golden("latent_losses")
    for bpb, bs in ((4, 2), (8, 8), (9, 12)):
        Z = (0.8 * O.closed_form((bpb * bs, 256), f"aug{bpb}_{bs}:Z", 1.0, 0.4142) + 0.1).to(DEV)
        Z.requires_grad_(True)
        a = Fh.augmented_loss(Z, bpb, bs)
        assert a.shape == (1,)
        a.sum().backward()
        ref = float(g[f"aug{bpb}_{bs}/loss"])
        assert abs(a.item() - ref) <= 3e-6 * abs(ref)
        assert_close(Z.grad, g[f"aug{bpb}_{bs}/dZ"], 2e-4, 2e-5 * np.abs(g[f"aug{bpb}_{bs}/dZ"]).max())


def test_augmented_loss_ragged_rows():
    # more rows than batch_size*bpb (ignored, zero gradient) and a trailing partial group
    Fh = _F()
    for rows, bpb, bs in ((20, 4, 3), (10, 4, 3), (3, 4, 2)):
        Z = (0.8 * O.closed_form((rows, 64), "augr:Z", 1.0, 0.4142) + 0.1)
        Zg = Z.to(DEV).requires_grad_(True)
        a = Fh.augmented_loss(Zg, bpb, bs)
        a.sum().backward()
        Zo = Z.clone().requires_grad_(True)
        ao = O.augmented_loss(Zo, bpb, bs)
        ao.sum().backward()
        assert abs(a.item() - ao.item()) <= 3e-6 * abs(ao.item()) + 1e-9, (rows, bpb, bs)
        assert_close(Zg.grad, Zo.grad, 2e-4, 1e-7)


def test_fft_features_vs_golden_and_known_answers():
    Fh = _F()
    g = load_golden("fft")
    r, _ = O.closed_form_inputs(2, 4)
    out = Fh.fft_features((r * 3.0).to(DEV))
    assert out.shape == (2, 8, 128, 128)
    assert_probe(g, "fft/out", out, 2e-5, 2e-5)
    assert_close(out[:, :, 60:68, 60:68], g["fft/crop"], 2e-5, 2e-5)
    assert rel_err(out, O.fft_features(r * 3.0)) < 2e-6
    # delta at the origin -> flat spectrum 1/128 in the real plane, zero imaginary plane
    d = torch.zeros(1, 1, 128, 128)
    d[0, 0, 0, 0] = 1.0
    o = Fh.fft_features(d.to(DEV)).cpu()
    assert torch.allclose(o[0, 0], torch.full((128, 128), 1.0 / 128), atol=1e-7)
    assert o[0, 1].abs().max() < 1e-7
    # one plane wave exp(2 pi i (3 y + 5 x)/128) real part -> two shifted peaks of height 64 (clamped to 10)
    yy, xx = torch.meshgrid(torch.arange(128.), torch.arange(128.), indexing="ij")
    w = torch.cos(2 * np.pi * (3 * yy + 5 * xx) / 128)[None, None]
    o = Fh.fft_features(w.to(DEV), clamp=100.0).cpu()
    assert abs(o[0, 0, 64 + 3, 64 + 5].item() - 64.0) < 1e-3 and abs(o[0, 0, 64 - 3, 64 - 5].item() - 64.0) < 1e-3
    o[0, 0, 64 + 3, 64 + 5] = 0
    o[0, 0, 64 - 3, 64 - 5] = 0
    assert o.abs().max() < 1e-3


@pytest.mark.parametrize("B,C,scale,clamp", [(2, 4, 3.0, 10.0), (1, 1, 1.0, 1e9), (3, 8, 3.0, 2.0)])
def test_fft_features_backward_vs_autograd(B, C, scale, clamp):
    """The feature step is differentiable (lshm_fft2_backward): gradient w.r.t. the real input against torch autograd
    through torch.fft.fftn + roll + cat + clamp (the oracle's restatement of Demo.ipynb:169-175 and
    src/lofar_tools.py:24-30), with the clamp active on part of the spectrum."""
    Fh = _F()
    r = O.closed_form((B, C, 128, 128), f"fftb:{B}{C}", scale, 0.37)
    gy = O.closed_form((B, 2 * C, 128, 128), f"fftb:g{B}{C}", 1.0, 0.91)
    rc = r.clone().requires_grad_(True)
    (O.fft_features(rc, clamp) * gy).sum().backward()
    rg = r.to(DEV).requires_grad_(True)
    out = Fh.fft_features(rg, clamp)
    (out * gy.to(DEV)).sum().backward()
    if clamp < 1e8:
        frac = float((out.detach().abs() >= clamp).float().mean())
        assert 0.0 < frac < 0.9, frac         # the clamp bites somewhere and not everywhere
    assert rel_err(rg.grad, rc.grad) < 5e-6
    # adjointness without the clamp: <F x, g> == <x, F^T g>
    if clamp >= 1e8:
        lhs = float((out.detach().double() * gy.to(DEV).double()).sum())
        rhs = float((r.to(DEV).double() * rg.grad.double()).sum())
        assert abs(lhs - rhs) <= 1e-5 * abs(lhs)


def test_fft_cascade_trains_end_to_end():
    """feature_mode='fft' as a trainable second stage: with detach_residual=False a loss on the second autoencoder's
    output back-propagates through the FFT feature op into the first autoencoder (same gradient as the oracle
    composition built from torch ops)."""
    from lshm_amd.fft_cascade import fft_cascade_forward
    from lshm_amd.lofar_models import AutoEncoderCNN2
    hs = torch.tensor(O.DEFAULT_SCALES)
    torch.manual_seed(0)
    net = AutoEncoderCNN2(32, 4, hs, True)
    fnet = AutoEncoderCNN2(16, 8, hs, True)
    x, uv = O.closed_form_inputs(2, 4)
    sd, fsd = net.state_dict(), fnet.state_dict()
    # oracle side (CPU, torch ops)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xhat, mu = O.ae_forward(p, x, uv, hs, 2, True)
    Fx = O.fft_features(x - xhat, 10.0)
    Fhat, fmu = O.ae_forward({k: v.clone() for k, v in fsd.items()}, Fx, uv, hs, 2, True)
    loss_ref = ((Fhat - Fx) ** 2).mean() + fmu.mean()
    gref = torch.autograd.grad(loss_ref, [p["conv0.weight"], p["tconv5.bias"]])
    net, fnet = net.to(DEV), fnet.to(DEV)
    net.harmonic_scales = fnet.harmonic_scales = hs.to(DEV)
    xh, mu_, Fx_, Fhat_, fmu_ = fft_cascade_forward(net, fnet, x.to(DEV), uv.to(DEV), detach_residual=False)
    loss = ((Fhat_ - Fx_) ** 2).mean() + fmu_.mean()
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) <= 1e-4 * abs(loss_ref.item())
    assert rel_err(net.conv0.weight.grad, gref[0]) < 2e-3
    assert rel_err(net.tconv5.bias.grad, gref[1]) < 2e-3


def test_full_size_conv_linearity_and_adjointness():
    """B=256 layer kernels (direct LDS-patch paths) through size-independent properties:
    linearity in the input and <conv(x), y> == <x, conv^T(y)> (the transposed conv is the adjoint)."""
    Fh = _F()
    g = torch.Generator().manual_seed(3)
    B = 256
    for cin, cout, hw in ((4, 8, 128), (8, 12, 64)):
        w = (0.1 * torch.randn(cout, cin, 4, 4, generator=g)).to(DEV)
        x1 = torch.randn(B, cin, hw, hw, generator=g).to(DEV)
        x2 = torch.randn(B, cin, hw, hw, generator=g).to(DEV)
        y = torch.randn(B, cout, hw // 2, hw // 2, generator=g).to(DEV)
        c1 = Fh.conv_act(x1, w, None, Fh.CONV2D, False)
        c2 = Fh.conv_act(x2, w, None, Fh.CONV2D, False)
        c12 = Fh.conv_act(x1 + 2.0 * x2, w, None, Fh.CONV2D, False)
        assert rel_err(c12, c1 + 2.0 * c2) < 2e-6
        # ConvTranspose2d with weight (cout, cin, 4, 4) viewed as (Cin_t=cout, Cout_t=cin) is conv^T
        ty = Fh.conv_act(y, w, None, Fh.TCONV2D, False)
        lhs = float((c1.double() * y.double()).sum())
        rhs = float((x1.double() * ty.double()).sum())
        assert abs(lhs - rhs) <= 2e-5 * max(abs(lhs), abs(rhs)) + 1e-3
    # 1-D pair: conv1d k4 s4 p1 vs transposed k4 s4 p0 are not adjoint (different padding): linearity only
    w1 = (0.1 * torch.randn(8, 4, 4, generator=g)).to(DEV)
    a = torch.randn(B, 4, 16384, generator=g).to(DEV)
    b2 = torch.randn(B, 4, 16384, generator=g).to(DEV)
    assert rel_err(Fh.conv_act(a - b2, w1, None, Fh.CONV1D, False),
                   Fh.conv_act(a, w1, None, Fh.CONV1D, False) - Fh.conv_act(b2, w1, None, Fh.CONV1D, False)) < 2e-6


def test_operand_tensor_of_4gib_is_refused_not_wrapped():
    """The implicit-GEMM fetch addresses a tensor with 32-bit byte offsets: a (strided) operand that spans 4 GiB
    or more must come back as an error from the C ABI, and the same layer just below the limit must run."""
    from lshm_amd import _lib as L
    lib = L.load()
    B, cin, cout, hw = 5, 12, 24, 16
    bs = 1 << 28  # floats between samples: B * bs * 4 bytes = 5 GiB (really allocated, so nothing can fault)
    x = torch.empty(B * bs, device=DEV)
    xv = torch.as_strided(x, (B, cin, hw, hw), (bs, hw * hw, hw, 1))
    xv.normal_()
    w = (0.1 * torch.randn(cout, cin, 4, 4)).to(DEV)
    y = torch.empty(B, cout, hw // 2, hw // 2, device=DEV)
    ws = torch.empty(max(int(lib.lshm_conv_workspace_floats(0, B, cin, cout, hw, hw)), 1), device=DEV)
    args = lambda nb: (0, L.ptr(x), L.ptr(w), None, L.ptr(y), nb, cin, cout, hw, hw, bs, y[0].numel(), 0, L.ptr(ws),
                       ws.numel(), L.stream())
    assert lib.lshm_conv_fwd(*args(B)) == -3  # LSHM_ERR_UNSUPPORTED
    assert b"4 GiB" in lib.lshm_last_error_string()
    L.check(lib.lshm_conv_fwd(*args(4)), "conv_fwd")  # 3 * 2^28 + 3072 floats: fits
    want = torch.nn.functional.conv2d(xv[:4].contiguous(), w, None, stride=2, padding=1)
    assert rel_err(y[:4], want) < 5e-6
    del x


def test_conv_fwd_pair_matches_two_single_launches():
    """lshm_conv_fwd_pair (netT and netF share every launch) must be bitwise the two single-problem results."""
    from lshm_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(11)
    B = 16
    for kind, cin, cout, win in ((2, 4, 8, 16384), (2, 24, 48, 256), (3, 48, 24, 16), (3, 8, 4, 4096)):
        xs = [torch.randn(B, cin, win, generator=g).to(DEV) for _ in range(2)]
        wshape = (cout, cin, 4) if kind == 2 else (cin, cout, 4)
        ws_ = [(0.2 * torch.randn(*wshape, generator=g)).to(DEV) for _ in range(2)]
        bs = [torch.randn(cout, generator=g).to(DEV) for _ in range(2)]
        wo = (win - 2) // 4 + 1 if kind == 2 else win * 4
        nws = lib.lshm_conv_workspace_floats(kind, B, cin, cout, 1, win)
        wk = torch.empty(2 * nws + 16, device=DEV)
        single = [torch.empty(B, cout, wo, device=DEV) for _ in range(2)]
        pair = [torch.empty(B, cout, wo, device=DEV) for _ in range(2)]
        for i in range(2):
            L.check(lib.lshm_conv_fwd(kind, L.ptr(xs[i]), L.ptr(ws_[i]), L.ptr(bs[i]), L.ptr(single[i]), B, cin, cout,
                                      1, win, 0, 0, 1, L.ptr(wk), nws, L.stream()))
        L.check(lib.lshm_conv_fwd_pair(kind, L.ptr(xs[0]), L.ptr(ws_[0]), L.ptr(bs[0]), L.ptr(pair[0]),
                                       L.ptr(xs[1]), L.ptr(ws_[1]), L.ptr(bs[1]), L.ptr(pair[1]), B, cin, cout,
                                       1, win, 0, 0, 1, L.ptr(wk), 2 * nws, L.stream()))
        torch.cuda.synchronize()
        for i in range(2):
            ref = torch.nn.functional.elu(
                torch.nn.functional.conv1d(xs[i].double().cpu(), ws_[i].double().cpu(), bs[i].double().cpu(), stride=4, padding=1)
                if kind == 2 else
                torch.nn.functional.conv_transpose1d(xs[i].double().cpu(), ws_[i].double().cpu(), bs[i].double().cpu(), stride=4))
            assert rel_err(pair[i], ref.float()) < 2e-6
            # the tile configuration is chosen per (shape, group count): same values up to summation order
            assert rel_err(pair[i], single[i]) < 1e-6


@pytest.mark.parametrize("K", [10, 64], ids=["K10-stream-kernel", "K64-matrix-core-kernel"])
def test_full_size_khm_streaming_properties(K):
    """N = 2^20 rows (the streaming shape): loss of the concatenation is the mean of the two halves'
    losses, dM adds, dX rows are independent; K-harmonic loss is invariant to a row permutation.  K = 64
    (config 5) takes the matrix-core kernel, K = 10 the streaming one."""
    Fh = _F()
    g = torch.Generator().manual_seed(9)
    N, D = 1 << 20, 256
    X = torch.rand(N, D, generator=g).to(DEV)
    M = torch.rand(K, D, generator=g).to(DEV)
    Xg, Mg = X.clone().requires_grad_(True), M.clone().requires_grad_(True)
    loss = Fh.khm_loss(Xg, Mg, 4)
    loss.backward()
    halves = []
    for h in range(2):
        xh = X[h * (N // 2):(h + 1) * (N // 2)].clone().requires_grad_(True)
        mh = M.clone().requires_grad_(True)
        lh = Fh.khm_loss(xh, mh, 4)
        lh.backward()
        halves.append((lh.item(), xh.grad, mh.grad))
    assert abs(loss.item() - 0.5 * (halves[0][0] + halves[1][0])) <= 2e-6 * abs(loss.item())
    assert rel_err(Mg.grad, 0.5 * (halves[0][2] + halves[1][2])) < 2e-5
    assert rel_err(Xg.grad[:N // 2], 0.5 * halves[0][1]) < 2e-6
    perm = torch.randperm(N, generator=g).to(DEV)
    lp = Fh.khm_loss(X[perm], M, 4)
    assert abs(lp.item() - loss.item()) <= 2e-6 * abs(loss.item())


def test_fft_cascade_matches_oracle_composition():
    """feature_mode="fft": net -> FFT features of the residual -> 8-channel second-stage AE."""
    from lshm_amd.fft_cascade import fft_cascade_forward, fft_cascade_latents
    from lshm_amd.lofar_models import AutoEncoderCNN2
    hs = torch.tensor(O.DEFAULT_SCALES)
    sd1 = O.closed_form_state_dict(224, 4, 2, True, "ae2d_rica")
    sd2 = O.closed_form_state_dict(64, 8, 2, False, "fnet8")
    net = AutoEncoderCNN2(224, 4, hs, True)
    fnet = AutoEncoderCNN2(64, 8, hs, False)
    net.load_state_dict(sd1)
    fnet.load_state_dict(sd2)
    net, fnet = net.to(DEV), fnet.to(DEV)
    net.harmonic_scales = fnet.harmonic_scales = hs.to(DEV)
    x, uv = O.closed_form_inputs(2, 4)
    xhat, mu, Fx, Fhat, fmu = fft_cascade_forward(net, fnet, x.to(DEV), uv.to(DEV))
    xo, muo = O.ae_forward(sd1, x, uv, hs, 2, True)
    Fo = O.fft_features(x - xo)
    Fho, fmuo = O.ae_forward(sd2, Fo, uv, hs, 2, False)
    assert rel_err(xhat, xo) < 1e-5 and rel_err(Fx, Fo) < 1e-5
    assert rel_err(Fhat, Fho) < 2e-5 and rel_err(fmu, fmuo) < 2e-5
    Z = fft_cascade_latents(net, fnet, x.to(DEV), uv.to(DEV))
    assert Z.shape == (2, 224 + 64) and rel_err(Z, torch.cat((muo, fmuo), 1)) < 2e-5


def test_recon_losses_and_gradient_free_variant():
    """Reconstruction terms of the closure (src/kharmonic_lofar.py:150-158) against the formulas in fp64, and
    the variant without gradient images (a closure under no_grad) giving the same seven sums."""
    from lshm_amd import _lib as L
    lib = L.load()
    planes, P, rho = 6, 64, 0.7
    g = torch.Generator().manual_seed(5)
    x, x1, x2, x3, y1, y2, y3 = [torch.randn(planes, P, P, generator=g) for _ in range(7)]
    x3c = x3.transpose(1, 2).contiguous()
    dev = [t.to(DEV) for t in (x, x1, x2, x3c, y1, y2, y3)]
    nws = lib.lshm_recon_workspace_floats(planes, P)
    ws = torch.empty(nws, device=DEV)
    sums = torch.zeros(8, device=DEV, dtype=torch.float64)
    grads = [torch.empty(planes, P, P, device=DEV) for _ in range(3)]
    L.check(lib.lshm_recon_losses_fwd_bwd(*[L.ptr(t) for t in dev], rho, planes, P, L.ptr(sums), *[L.ptr(t) for t in grads],
                                          L.ptr(ws), L.stream()))
    X, X1, X2, X3, Y1, Y2, Y3 = [t.double() for t in (x, x1, x2, x3, y1, y2, y3)]
    r1 = X - X1; h = 0.5 * r1; r2 = h - X2; r3 = h - X3; e = X1 + X2 + X3 - X
    want = [float((e * e).sum()), float((Y1 * r1).sum()), float((r1 * r1).sum()), float((Y2 * r2).sum()),
            float((r2 * r2).sum()), float((Y3 * r3).sum()), float((r3 * r3).sum())]
    got = sums[:7].cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-3)
    n = planes * P * P
    t2, t3 = Y2 + rho * r2, Y3 + rho * r3
    assert_close(grads[1], ((2 * e - t2) / n).float().numpy(), rtol=1e-5, atol=1e-9)
    assert_close(grads[2].transpose(1, 2), ((2 * e - t3) / n).float().numpy(), rtol=1e-5, atol=1e-9)
    assert_close(grads[0], (((2 * e - Y1 - rho * r1) - 0.5 * (t2 + t3)) / n).float().numpy(), rtol=1e-5, atol=1e-9)
    sums2 = torch.zeros(8, device=DEV, dtype=torch.float64)
    L.check(lib.lshm_recon_losses_fwd_bwd(*[L.ptr(t) for t in dev], rho, planes, P, L.ptr(sums2), None, None, None,
                                          L.ptr(ws), L.stream()))
    assert torch.equal(sums[:7], sums2[:7])
    # mixed NULL / non-NULL gradient images are refused
    assert lib.lshm_recon_losses_fwd_bwd(*[L.ptr(t) for t in dev], rho, planes, P, L.ptr(sums2), L.ptr(grads[0]), None, None,
                                         L.ptr(ws), L.stream()) != 0


def test_flat_dot_alignment_and_tail():
    """lshm_dot_flat (src/lbfgsnew.py inner products): float4 path with an n % 4 tail, and the scalar path
    for a pointer that is not 16-byte aligned."""
    from lshm_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(9)
    base_a = torch.randn(100003 + 1, generator=g).to(DEV)
    base_b = torch.randn(100003 + 1, generator=g).to(DEV)
    out = torch.zeros(1, device=DEV, dtype=torch.float64)
    ws = torch.empty(512, device=DEV)
    for off in (0, 1):
        a, b = base_a[off:off + 100003], base_b[off:off + 100003]
        L.check(lib.lshm_dot_flat(L.ptr(a), L.ptr(b), a.numel(), L.ptr(out), L.ptr(ws), L.stream()))
        want = float((a.double() * b.double()).sum())
        assert abs(float(out[0]) - want) <= 1e-9 * max(1.0, abs(want))


@pytest.mark.parametrize("n,off", [(100003, 0), (100003, 1), (4096, 0), (7, 0)])
def test_multi_dot_and_device_two_loop_recursion(n, off):
    """lshm_multi_dot_flat and lshm_lbfgs_direction (src/lbfgsnew.py:632-651 on the device) against the host loop
    in fp64, for m = 0, 1, 3, 7 stored pairs, aligned and unaligned vectors, an n % 4 tail, and through the
    optimiser's own vector backends (the device backend and the host-loop backend must agree)."""
    from lshm_amd import lbfgsnew as LB
    g = torch.Generator().manual_seed(31 + n + off)
    mk = lambda: torch.randn(n + off, generator=g).to(DEV)[off:]
    vec = LB._HipVec(torch.device(DEV))
    pairs = [(mk(), mk()) for _ in range(5)]
    got = vec.dots(pairs)
    for (a, b), v in zip(pairs, got):
        want = float((a.double() * b.double()).sum())
        assert abs(v - want) <= 1e-9 * max(1.0, abs(want))
    for m in (0, 1, 3, 7):
        ys = [mk() for _ in range(m)]
        ss = [y * 0.3 + 0.1 * mk() for y in ys]  # <y_i, s_i> > 0, as the curvature test guarantees
        grad = mk()
        h = 0.37
        d = vec.direction(ys, ss, grad, h)
        # the host loop of the reference, in fp64
        q = -grad.double()
        ro = [1.0 / float((y.double() * s_.double()).sum()) for y, s_ in zip(ys, ss)]
        al = [0.0] * m
        for i in range(m - 1, -1, -1):
            al[i] = float((ss[i].double() * q).sum()) * ro[i]
            q -= al[i] * ys[i].double()
        q *= h
        for i in range(m):
            be = float((ys[i].double() * q).sum()) * ro[i]
            q += (al[i] - be) * ss[i].double()
        assert rel_err(d, q.float()) < 5e-6, m
        # and the fp32 host loop on the device primitives (what the optimiser ran before): same numbers
        d_loop = LB._two_loop(vec, ys, ss, grad, h)
        assert rel_err(d, d_loop) < 5e-6, m
        assert ys[0].data_ptr() != d.data_ptr() if m else True
    # argument checks: too many pairs, aliasing
    lib = vec.lib
    import ctypes as C
    big = (C.c_void_p * 17)(*[grad.data_ptr()] * 17)
    ws = torch.empty(int(lib.lshm_lbfgs_direction_workspace_doubles(16)), device=DEV, dtype=torch.float64)
    assert lib.lshm_lbfgs_direction(big, big, 17, grad.data_ptr(), 1.0, d.data_ptr(), n, ws.data_ptr(), ws.numel(), None) != 0
    one = (C.c_void_p * 1)(d.data_ptr())
    assert lib.lshm_lbfgs_direction(one, one, 1, grad.data_ptr(), 1.0, d.data_ptr(), n, ws.data_ptr(), ws.numel(), None) != 0


def _bf16_round(t):
    return t.to(torch.bfloat16).to(t.dtype)


@pytest.mark.parametrize("cfg", [-1, 0, 3, 6, 9, 12, 13, 14, 15, 16, 19, 20])
def test_bf16_matrix_operands(cfg):
    """The `_bf16` entry points, BASELINE configs[2]: the GEMM-shaped kernels round their operands to bf16
    (nearest even) and accumulate in fp32.  Checked against fp64 products of the bf16-rounded operands
    (that is the exact result up to fp32 summation), tight, and against the fp32 result, loose -- and the
    difference from the fp32 result must be visible, i.e. the switch really reached the kernel."""
    from lshm_amd import _lib
    lib = _lib.load()
    Fh = _F()
    B, K, N = 48, 784, 224
    x = O.closed_form((B, K), "bf:x", 1.0, 0.31).to(DEV)
    w = O.closed_form((N, K), "bf:w", K ** -0.5, 0.77).to(DEV)
    b = O.closed_form((N,), "bf:b", 0.1, 0.53).to(DEV)
    gy = O.closed_form((B, N), "bf:gy", 1.0, 0.91).to(DEV)

    def run(bf16=False):
        xs, ws, bs = (t.clone().requires_grad_(True) for t in (x, w, b))
        y = Fh.linear_act(xs, ws, bs, False, bf16)
        (y * gy).sum().backward()
        return y.detach(), xs.grad, ws.grad, bs.grad
    y32, dx32, dw32, db32 = run()
    lib.lshm_set_tuning(0, cfg)
    try:
        y16, dx16, dw16, db16 = run(True)
        # a convolution through the same template: conv2d layer 3 (24 -> 48 channels at 16 x 16)
        xc = O.closed_form((2, 24, 16, 16), "bf:xc", 1.0, 0.2113).to(DEV)
        wc = O.closed_form((48, 24, 4, 4), "bf:wc", (3.0 / 384) ** 0.5).to(DEV)
        bc = O.closed_form((48,), "bf:bc", 0.05).to(DEV)
        yc16 = Fh.conv_act(xc, wc, bc, KINDS["conv2d"], False, True)
        yc32 = Fh.conv_act(xc, wc, bc, KINDS["conv2d"], False)   # the precision is per call: fp32 right after
    finally:
        lib.lshm_set_tuning(0, -1)
    assert rel_err(yc32, torch.nn.functional.conv2d(xc.double().cpu(), wc.double().cpu(), bc.double().cpu(),
                                                    stride=2, padding=1)) < 5e-6
    xd, wd, gd = _bf16_round(x).double().cpu(), _bf16_round(w).double().cpu(), _bf16_round(gy).double().cpu()
    assert rel_err(y16, xd @ wd.t() + b.double().cpu()) < 5e-6
    assert rel_err(dx16, gd @ wd) < 5e-6
    assert rel_err(dw16, gd.t() @ xd) < 5e-6
    assert rel_err(db16, db32) < 1e-6                      # column sums are not a GEMM: unchanged
    for a16, a32 in ((y16, y32), (dx16, dx32), (dw16, dw32)):
        assert 1e-4 < rel_err(a16, a32) < 5e-2             # bf16 operands: 2^-9 per factor, cancelling sums
    yc = torch.nn.functional.conv2d(_bf16_round(xc).double().cpu(), _bf16_round(wc).double().cpu(), bc.double().cpu(),
                                    stride=2, padding=1)
    assert rel_err(yc16, yc) < 5e-6
