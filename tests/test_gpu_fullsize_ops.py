"""The implicit-GEMM policies at the layer shapes of BASELINE.json's configuration (B = 256), not the B = 2 shapes of
the golden fixtures.  Operands are fetched with raw buffer loads whose out-of-range offsets read as ZERO, so a wrong
edge predicate at full size would produce a plausible number instead of a fault: every policy is therefore compared
here against an fp64 evaluation of the same layer (torch CPU conv; the layer math of src/lofar_models.py:31-57,
115-142) on whole samples that include the first and the last batch entry -- first / last rows of the GEMM, first /
last k-block, the padded border -- with the tile table shipped for these shapes in effect."""
import pytest
import torch
import torch.nn.functional as TF

from tests.util import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
B = 256
CH = (4, 8, 12, 24, 48, 96, 192)
SAMPLES = (0, 1, 127, 254, 255)


def _shapes(kind, i):
    nd2 = kind < 2
    tr = kind in (1, 3)
    cin, cout = (CH[6 - i], CH[5 - i]) if tr else (CH[i], CH[i + 1])
    if nd2:
        hw = (2 << i) if tr else (128 >> i)
        ishape = (B, cin, hw, hw)
    else:
        L = (4 << (2 * i)) if tr else (16384 >> (2 * i))
        ishape = (B, cin, L)
    wshape = ((cin, cout) if tr else (cout, cin)) + ((4, 4) if nd2 else (4,))
    return ishape, wshape, cin, cout


def _ref_layer(kind, x, w, b):
    if kind == 0:
        return TF.conv2d(x, w, b, stride=2, padding=1)
    if kind == 1:
        return TF.conv_transpose2d(x, w, b, stride=2, padding=1)
    if kind == 2:
        return TF.conv1d(x, w, b, stride=4, padding=1)
    return TF.conv_transpose1d(x, w, b, stride=4, padding=0)


@pytest.mark.parametrize("i", [2, 3, 4, 5])
@pytest.mark.parametrize("kind", [0, 1, 2, 3], ids=["conv2d", "tconv2d", "conv1d", "tconv1d"])
def test_gemm_shaped_layers_at_full_batch(kind, i):
    """Forward (bias + ELU), data gradient with the ELU' multiply of a saved activation, weight and bias gradient of
    conv2-5 / tconv0-3 (decoder index i <-> 5 - i counted from the output) at B = 256."""
    from lshm_amd import _lib as L
    lib = L.load()
    if kind in (1, 3):
        i = 5 - i  # tconv0..3 are the GEMM-shaped transposed layers
    ishape, wshape, cin, cout = _shapes(kind, i)
    g = torch.Generator().manual_seed(100 * kind + i)
    x = torch.randn(ishape, generator=g)
    fan = (cout if kind in (1, 3) else cin) * (16 if kind < 2 else 4)  # (any scale will do: the products are linear in w)
    w = torch.randn(wshape, generator=g) * (3.0 / fan) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    Hin, Win = (ishape[2], ishape[3]) if kind < 2 else (1, ishape[2])
    nws = lib.lshm_conv_workspace_floats(kind, B, cin, cout, Hin, Win)
    ws = torch.empty(max(nws, 1), device=DEV)
    st = L.stream()
    yref = TF.elu(_ref_layer(kind, x[list(SAMPLES)].double(), w.double(), b.double()))
    y = torch.empty((B,) + tuple(yref.shape[1:]), device=DEV)
    L.check(lib.lshm_conv_fwd(kind, L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(y), B, cin, cout, Hin, Win, 0, 0, 1, L.ptr(ws),
                              nws, st), "conv_fwd")
    assert rel_err(y[list(SAMPLES)], yref) < 2e-5
    # data gradient, multiplied by ELU'(saved input) as inside the backward pass (the saved input is an ELU output)
    dz = torch.randn(y.shape, generator=g)
    saved = TF.elu(torch.randn(ishape, generator=g))
    dzd, savedd = dz.to(DEV), saved.to(DEV)
    dx = torch.empty(ishape, device=DEV)
    L.check(lib.lshm_conv_dgrad(kind, L.ptr(dzd), L.ptr(wd), L.ptr(dx), L.ptr(savedd), B, cin, cout, Hin, Win, 0, 0,
                                L.ptr(ws), nws, st), "conv_dgrad")
    xs = x[list(SAMPLES)].double().requires_grad_(True)
    _ref_layer(kind, xs, w.double(), None).backward(dz[list(SAMPLES)].double())
    sv = saved[list(SAMPLES)].double()
    dxref = xs.grad * torch.where(sv > 0, torch.ones_like(sv), sv + 1.0)
    assert rel_err(dx[list(SAMPLES)], dxref) < 5e-5
    # weight / bias gradient: a sum over the WHOLE batch -- fp64 on the CPU over all 256 samples
    dw, db = torch.empty(wshape, device=DEV), torch.empty(cout, device=DEV)
    L.check(lib.lshm_conv_wgrad(kind, L.ptr(xd), L.ptr(dzd), L.ptr(dw), L.ptr(db), B, cin, cout, Hin, Win, 0, 0, L.ptr(ws),
                                nws, 0, st), "conv_wgrad")
    wref = w.double().requires_grad_(True)
    bref = b.double().requires_grad_(True)
    _ref_layer(kind, x.double(), wref, bref).backward(dz.double())
    assert rel_err(dw, wref.grad) < 1e-4
    assert rel_err(db, bref.grad) < 1e-4


@pytest.mark.parametrize("i", [2, 3, 4, 5])
@pytest.mark.parametrize("kind", [2, 3], ids=["conv1d", "tconv1d"])
def test_paired_1d_launches_at_full_batch(kind, i):
    """netT and netF share every launch of their GEMM-shaped layers (grid.z carries the problem): both halves of a
    paired forward at B = 256 against fp64."""
    from lshm_amd import _lib as L
    lib = L.load()
    if kind == 3:
        i = 5 - i
    ishape, wshape, cin, cout = _shapes(kind, i)
    g = torch.Generator().manual_seed(7 + 10 * kind + i)
    xs = [torch.randn(ishape, generator=g) for _ in range(2)]
    fan = (cout if kind == 3 else cin) * 4
    wsx = [torch.randn(wshape, generator=g) * (3.0 / fan) ** 0.5 for _ in range(2)]
    bs = [torch.randn(cout, generator=g) * 0.1 for _ in range(2)]
    nws = lib.lshm_conv_workspace_floats(kind, B, cin, cout, 1, ishape[2])
    ws = torch.empty(max(2 * nws, 1), device=DEV)
    refs = [TF.elu(_ref_layer(kind, xs[q][list(SAMPLES)].double(), wsx[q].double(), bs[q].double())) for q in range(2)]
    d = [t.to(DEV) for t in xs + wsx + bs]
    ys = [torch.empty((B,) + tuple(refs[0].shape[1:]), device=DEV) for _ in range(2)]
    L.check(lib.lshm_conv_fwd_pair(kind, L.ptr(d[0]), L.ptr(d[2]), L.ptr(d[4]), L.ptr(ys[0]), L.ptr(d[1]), L.ptr(d[3]),
                                   L.ptr(d[5]), L.ptr(ys[1]), B, cin, cout, 1, ishape[2], 0, 0, 1, L.ptr(ws), 2 * nws,
                                   L.stream()), "conv_fwd_pair")
    for q in range(2):
        assert rel_err(ys[q][list(SAMPLES)], refs[q]) < 2e-5


@pytest.mark.parametrize("K,N,act", [(784, 224, 1), (224, 224, 1), (240, 768, 0), (784, 16, 1), (32, 768, 0)])
def test_dense_layers_at_full_batch(K, N, act):
    """The four strided-GEMM policies (forward, data gradient with addend + ELU', weight gradient) at B = 256 and the
    dense-layer shapes of the three autoencoders (fc1, fc2in / fc2out, fc3), every row compared."""
    from lshm_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(K + N)
    x = torch.randn(B, K, generator=g)
    w = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g) * 0.1
    dz = torch.randn(B, N, generator=g)
    saved = TF.elu(torch.randn(B, K, generator=g))
    xd, wd, bd, dzd, sd = (t.to(DEV) for t in (x, w, b, dz, saved))
    nws = lib.lshm_linear_workspace_floats(B, K, N)
    ws = torch.empty(nws, device=DEV)
    st = L.stream()
    y = torch.empty(B, N, device=DEV)
    L.check(lib.lshm_linear_fwd(L.ptr(xd), K, L.ptr(wd), L.ptr(bd), L.ptr(y), N, B, K, N, act, L.ptr(ws), nws, st))
    yref = x.double() @ w.double().t() + b.double()
    assert rel_err(y, TF.elu(yref) if act else yref) < 2e-5
    dx = torch.empty(B, K, device=DEV)
    L.check(lib.lshm_linear_dgrad(L.ptr(dzd), N, L.ptr(wd), L.ptr(dx), K, L.ptr(sd), K, B, K, N, L.ptr(ws), nws, st))
    sv = saved.double()
    assert rel_err(dx, (dz.double() @ w.double()) * torch.where(sv > 0, torch.ones_like(sv), sv + 1.0)) < 5e-5
    dw, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
    L.check(lib.lshm_linear_wgrad(L.ptr(xd), K, L.ptr(dzd), N, L.ptr(dw), L.ptr(db), B, K, N, L.ptr(ws), nws, st))
    assert rel_err(dw, dz.double().t() @ x.double()) < 5e-5
    assert rel_err(db, dz.double().sum(0)) < 5e-5


@pytest.mark.parametrize("kind,i,elu_grad,form", [(3, 5, 1, ""), (3, 4, 1, "lds"), (2, 1, 1, "lds"), (3, 4, 0, "lds"), (2, 1, 0, "lds"),
                                                  (2, 0, 0, "lds"), (2, 0, 1, "lds"),
                                                  (2, 0, 0, "reg84"), (2, 0, 1, "reg84"), (3, 5, 0, ""), (1, 5, 1, ""), (1, 5, 0, ""),
                                                  (1, 4, 1, ""), (1, 4, 0, ""), (0, 1, 1, ""), (0, 1, 0, "")],
                         ids=["tconv5", "tconv4", "conv1", "tconv4-noelu", "conv1-noelu",
                              "conv0", "conv0-elu", "conv0-registers", "conv0-elu-registers", "tconv5-noelu", "2d-tconv5",
                              "2d-tconv5-noelu", "2d-tconv4", "2d-tconv4-noelu", "2d-conv1", "2d-conv1-noelu"])
def test_one_pass_backward_of_outer_layers(kind, i, elu_grad, form):
    """lshm_conv_bwd_fused (weight + bias + data gradient from one read of dz and the saved input; backward of
    src/lofar_models.py:115-117,140-142) against the pair of kernels it replaces (lshm_conv_wgrad + lshm_conv_dgrad)
    and against fp64, at B = 256: the weight / bias gradients come from the same MFMA sequence as the stand-alone
    weight-gradient kernel (fp32 rounding apart: the pad-1 windows are assembled from other loads), the data
    gradient agrees to fp32 rounding.  "registers": 1-D conv0 on the register form (LSHM_SCHED_NO_BWD_LDS_8_4 through
    lshm_conv_bwd_fused_ex: the kernel bf16 storage runs)."""
    from lshm_amd import _lib as L
    lib = L.load()
    ishape, wshape, cin, cout = _shapes(kind, i)
    word = L.SCHEDULE_BITS["no_bwd_lds_8_4"] if form == "reg84" else 0
    g = torch.Generator().manual_seed(31 * kind + i)
    x = TF.elu(torch.randn(ishape, generator=g))  # the saved input of a layer behind an ELU
    fan = (cout if kind in (1, 3) else cin) * (16 if kind < 2 else 4)  # (any scale will do: the products are linear in w)
    w = torch.randn(wshape, generator=g) * (3.0 / fan) ** 0.5
    if kind < 2:
        Hin, Win = ishape[2], ishape[3]
        dz = torch.randn(B, cout, 2 * Hin, 2 * Win, generator=g) if kind == 1 else torch.randn(B, cout, Hin // 2, Win // 2, generator=g)
    else:
        Hin, Win = 1, ishape[2]
        dz = torch.randn(B, cout, Win * 4 if kind == 3 else Win // 4, generator=g)
    xd, wd, dzd = x.to(DEV), w.to(DEV), dz.to(DEV)
    nws = lib.lshm_conv_workspace_floats(kind, B, cin, cout, Hin, Win)
    ws = torch.empty(nws, device=DEV)
    st = L.stream()
    dw, db, dx = torch.empty(wshape, device=DEV), torch.empty(cout, device=DEV), torch.full(ishape, float("nan"), device=DEV)
    L.check(lib.lshm_conv_bwd_fused_ex(kind, L.ptr(xd), L.ptr(dzd), L.ptr(wd), L.ptr(dw), L.ptr(db), L.ptr(dx), elu_grad, B, cin,
                                       cout, Hin, Win, L.ptr(ws), nws, word, st), "conv_bwd_fused")
    dw2, db2, dx2 = torch.empty_like(dw), torch.empty_like(db), torch.empty_like(dx)
    L.check(lib.lshm_conv_wgrad(kind, L.ptr(xd), L.ptr(dzd), L.ptr(dw2), L.ptr(db2), B, cin, cout, Hin, Win, 0, 0, L.ptr(ws),
                                nws, 0, st), "conv_wgrad")
    L.check(lib.lshm_conv_dgrad(kind, L.ptr(dzd), L.ptr(wd), L.ptr(dx2), L.ptr(xd) if elu_grad else None, B, cin, cout, Hin,
                                Win, 0, 0, L.ptr(ws), nws, st), "conv_dgrad")
    torch.cuda.synchronize()
    assert torch.isfinite(dx).all()
    assert rel_err(dw, dw2) < 2e-6 and rel_err(db, db2) < 2e-6
    assert rel_err(dx, dx2) < 2e-6
    # fp64: whole batch for the parameter gradients, five samples for the data gradient
    wref = w.double().requires_grad_(True)
    bref = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    xs = x.double().requires_grad_(True)
    _ref_layer(kind, xs, wref, bref).backward(dz.double())
    assert rel_err(dw, wref.grad) < 1e-4 and rel_err(db, bref.grad) < 1e-4
    sv = x[list(SAMPLES)].double()
    dxref = xs.grad[list(SAMPLES)] * (torch.where(sv > 0, torch.ones_like(sv), sv + 1.0) if elu_grad else 1.0)
    assert rel_err(dx[list(SAMPLES)], dxref) < 5e-5


def test_one_pass_backward_refuses_other_layers():
    from lshm_amd import _lib as L
    lib = L.load()
    t = torch.zeros(16, device=DEV)
    rc = lib.lshm_conv_bwd_fused(2, L.ptr(t), L.ptr(t), L.ptr(t), L.ptr(t), L.ptr(t), L.ptr(t), 1, 2, 24, 48, 1, 256, L.ptr(t), 16,
                                 L.stream())
    assert rc == -3 and b"one-pass" in lib.lshm_last_error_string()


@pytest.mark.parametrize("mode", ["down-forward", "down-dgrad", "up-forward", "up-dgrad"])
def test_three_layer_chain_matches_the_three_launches(mode):
    """lshm_conv1d_chain3 (conv2 -> conv3 -> conv4 / tconv1 -> tconv2 -> tconv3 of AutoEncoder1DCNN and the data-gradient
    chains through the same layers, activations resident in LDS, src/lofar_models.py:119-123,138-140) against the three
    separate launches it replaces and against fp64, B = 256: every stage's output."""
    import ctypes as C
    from lshm_amd import _lib as L
    lib = L.load()
    up = mode.startswith("up")
    fwd = mode.endswith("forward")
    g = torch.Generator().manual_seed({"down-forward": 1, "down-dgrad": 2, "up-forward": 3, "up-dgrad": 4}[mode])
    ch = (96, 48, 24, 12) if up else (12, 24, 48, 96)
    Ls = [16, 64, 256, 1024] if up else [1024, 256, 64, 16]
    # forward: conv (down) / tconv (up) layers; dgrad: the data gradient of tconv (down direction) / conv (up direction)
    kind = (3 if up else 2) if fwd else (2 if up else 3)
    x = torch.randn(B, ch[0], Ls[0], generator=g)
    ws_, bs_, saved = [], [], []
    for k in range(3):
        cin, cout = ch[k], ch[k + 1]
        if fwd:
            shape = (cin, cout, 4) if up else (cout, cin, 4)
        else:  # the layer whose data gradient this stage is maps ch[k+1] -> ch[k] channels
            shape = (ch[k], ch[k + 1], 4) if up else (ch[k + 1], ch[k], 4)
        ws_.append(torch.randn(shape, generator=g) * (3.0 / (4 * max(cin, cout))) ** 0.5)
        bs_.append(torch.randn(cout, generator=g) * 0.1)
        saved.append(TF.elu(torch.randn(B, cout, Ls[k + 1], generator=g)))
    xd = x.to(DEV)
    wd, bd, sd = [t.to(DEV) for t in ws_], [t.to(DEV) for t in bs_], [t.to(DEV) for t in saved]
    outs = [torch.full((B, ch[k + 1], Ls[k + 1]), float("nan"), device=DEV) for k in range(3)]
    arr = lambda ts: (C.c_void_p * 3)(*[t.data_ptr() for t in ts])
    pad = (0 if up else 1) if fwd else (1 if up else 0)
    L.check(lib.lshm_conv1d_chain3(int(up), L.ptr(xd), arr(wd), arr(bd) if fwd else None, arr(outs), None if fwd else arr(sd),
                                   int(fwd), pad, B, L.stream()), "conv1d_chain3")
    torch.cuda.synchronize()
    # the separate launches
    cur = xd
    for k in range(3):
        ref = torch.empty_like(outs[k])
        if fwd:
            cin, cout, Lin = ch[k], ch[k + 1], Ls[k]
            nws = lib.lshm_conv_workspace_floats(kind, B, cin, cout, 1, Lin)
            wsb = torch.empty(max(nws, 1), device=DEV)
            L.check(lib.lshm_conv_fwd(kind, L.ptr(cur), L.ptr(wd[k]), L.ptr(bd[k]), L.ptr(ref), B, cin, cout, 1, Lin, 0, 0, 1,
                                      L.ptr(wsb), nws, L.stream()))
        else:  # data gradient of the layer (ch[k+1] -> ch[k] channels, input length Ls[k+1])
            cin, cout, Lin = ch[k + 1], ch[k], Ls[k + 1]
            nws = lib.lshm_conv_workspace_floats(kind, B, cin, cout, 1, Lin)
            wsb = torch.empty(max(nws, 1), device=DEV)
            L.check(lib.lshm_conv_dgrad(kind, L.ptr(cur), L.ptr(wd[k]), L.ptr(ref), L.ptr(sd[k]), B, cin, cout, 1, Lin, 0, 0,
                                        L.ptr(wsb), nws, L.stream()))
        torch.cuda.synchronize()
        assert torch.isfinite(outs[k]).all(), (mode, k)
        assert rel_err(outs[k], ref) < 2e-6, (mode, k)
        cur = ref
    # fp64 on five samples
    cur = x[list(SAMPLES)].double()
    for k in range(3):
        if fwd:
            y = TF.elu(_ref_layer(kind, cur, ws_[k].double(), bs_[k].double()))
        else:
            xin = torch.zeros(len(SAMPLES), ch[k + 1], Ls[k + 1], dtype=torch.float64, requires_grad=True)
            _ref_layer(kind, xin, ws_[k].double(), None).backward(cur)
            sv = saved[k][list(SAMPLES)].double()
            y = xin.grad * torch.where(sv > 0, torch.ones_like(sv), sv + 1.0)
        assert rel_err(outs[k][list(SAMPLES)], y) < 2e-5, (mode, k)
        cur = y.detach()


@pytest.mark.parametrize("nb", [256, 37])
@pytest.mark.parametrize("Ld", [16], ids=["latent16"])
def test_dense_middle_of_the_1d_autoencoder_as_one_launch(nb, Ld):
    """lshm_dense1d_fwd / lshm_dense1d_bwd (fc1 -> fc2in -> fc2out -> fc3 of AutoEncoder1DCNN(latent_dim=16, rica=True) and the
    data gradients back through them, src/lofar_models.py:127-135,165-176) against fp64, every output, at the full batch and at a
    ragged one (37 rows: the last workgroup has 5)."""
    import ctypes as C
    from lshm_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(5 + nb)
    R = lambda *s: torch.randn(*s, generator=g)
    cat1 = TF.elu(R(nb, 784))
    W = {"fc1": R(Ld, 784) * 784 ** -0.5, "fc2in": R(Ld, Ld) * Ld ** -0.5, "fc2out": R(Ld, Ld) * Ld ** -0.5,
         "fc3": R(768, Ld + 16) * (Ld + 16) ** -0.5}
    Bv = {k: R(v.shape[0]) * 0.1 for k, v in W.items()}
    uvh = TF.elu(R(nb, 16))
    D = 288
    import functools
    fwd, bwd = lib.lshm_dense1d_fwd, lib.lshm_dense1d_bwd
    Mu = torch.zeros(nb, D)
    cat3 = torch.zeros(nb, Ld + 16)
    cat3[:, Ld:] = uvh
    dev = lambda t: t.to(DEV).contiguous()
    cat1d, Mud, cat3d = dev(cat1), dev(Mu), dev(cat3)
    z1d, d0d = torch.empty(nb, Ld, device=DEV), torch.empty(nb, 768, device=DEV)
    wb = [dev(t) for k in ("fc1", "fc2in", "fc2out", "fc3") for t in (W[k], Bv[k])]
    arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    mu_col = 272 if Ld == 16 else 0
    L.check(fwd(L.ptr(cat1d), arr(wb), L.ptr(z1d), Mud.data_ptr() + 4 * mu_col, D, L.ptr(cat3d), L.ptr(d0d), nb,
                                 L.stream()), "dense1d_fwd")
    torch.cuda.synchronize()
    d = lambda t: t.double()
    z1 = TF.elu(d(cat1) @ d(W["fc1"]).t() + d(Bv["fc1"]))
    mu = TF.elu(z1 @ d(W["fc2in"]).t() + d(Bv["fc2in"]))
    c = TF.elu(mu @ d(W["fc2out"]).t() + d(Bv["fc2out"]))
    c3 = torch.cat((c, d(uvh)), 1)
    d0 = c3 @ d(W["fc3"]).t() + d(Bv["fc3"])
    assert rel_err(z1d, z1) < 1e-5 and rel_err(Mud[:, mu_col:mu_col + Ld], mu) < 1e-5
    assert rel_err(cat3d, c3) < 1e-5 and rel_err(d0d, d0) < 1e-5
    assert torch.equal(Mud[:, :mu_col].cpu(), Mu[:, :mu_col]) and torch.equal(Mud[:, mu_col + Ld:].cpu(), Mu[:, mu_col + Ld:])  # nothing outside the latent columns was touched
    # backward
    dd0, gmu = R(nb, 768), R(nb, D) * 0.1
    outs = [torch.full(s, float("nan"), device=DEV) for s in ((nb, Ld + 16), (nb, Ld), (nb, Ld), (nb, 784))]
    ws_ = [dev(W[k]) for k in ("fc1", "fc2in", "fc2out", "fc3")]
    gmud = dev(gmu)
    L.check(bwd(L.ptr(dev(dd0)), L.ptr(cat3d), Mud.data_ptr() + 4 * mu_col, D, gmud.data_ptr() + 4 * mu_col, D,
                                 L.ptr(z1d), L.ptr(cat1d), arr(ws_), *[L.ptr(t) for t in outs], nb, L.stream()), "dense1d_bwd")
    torch.cuda.synchronize()
    eg = lambda y: torch.where(y > 0, torch.ones_like(y), y + 1.0)
    dcat3 = (d(dd0) @ d(W["fc3"])) * eg(c3)
    dzmu = (dcat3[:, :Ld] @ d(W["fc2out"]) + d(gmu[:, mu_col:mu_col + Ld])) * eg(mu)
    dz1 = (dzmu @ d(W["fc2in"])) * eg(z1)
    dcat1 = (dz1 @ d(W["fc1"])) * eg(d(cat1))
    for got, ref in zip(outs, (dcat3, dzmu, dz1, dcat1)):
        assert torch.isfinite(got).all() and rel_err(got, ref) < 2e-5


@pytest.mark.parametrize("nb,keep", [(256, False), (256, True), (3, False), (13, True), (8, True)])
def test_first_1d_layers_straight_from_the_images_are_bitwise_the_two_launches(nb, keep):
    """lshm_resid_conv0 / lshm_resid_conv0_keep (netT.conv0 and netF.conv0 of the row- / column-vectorised residual
    (x - x1) / 2 in one launch from one 64 x 64 tile per workgroup; nothing materialised, or -- keep -- both vectorisations
    written too; src/kharmonic_lofar.py:142-147, src/lofar_models.py:115) against lshm_residual_split followed by
    lshm_conv_fwd_pair: bit for bit, every output; and against fp64."""
    from lshm_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(77 + nb)
    x, x1 = torch.randn(nb, 4, 128, 128, generator=g), torch.randn(nb, 4, 128, 128, generator=g)
    w = [torch.randn(8, 4, 4, generator=g) * 0.3 for _ in range(2)]
    b = [torch.randn(8, generator=g) * 0.1 for _ in range(2)]
    xd, x1d = x.to(DEV), x1.to(DEV)
    wd, bd = [t.to(DEV) for t in w], [t.to(DEV) for t in b]
    st, P = L.stream(), L.ptr
    y = [torch.full((nb, 8, 4096), float("nan"), device=DEV) for _ in range(2)]
    kept = [torch.full((nb, 4, 16384), float("nan"), device=DEV) for _ in range(2)]
    if keep:
        L.check(lib.lshm_resid_conv0_keep(P(xd), P(x1d), P(wd[0]), P(bd[0]), P(y[0]), P(wd[1]), P(bd[1]), P(y[1]), P(kept[0]), P(kept[1]),
                                          nb, st), "resid_conv0_keep")
    else:
        L.check(lib.lshm_resid_conv0(P(xd), P(x1d), P(wd[0]), P(bd[0]), P(y[0]), P(wd[1]), P(bd[1]), P(y[1]), nb, st), "resid_conv0")
    row, col = torch.empty(nb, 4, 16384, device=DEV), torch.empty(nb, 4, 16384, device=DEV)
    L.check(lib.lshm_residual_split(P(xd), P(x1d), P(row), P(col), nb * 4, 128, st), "residual_split")
    z = [torch.empty(nb, 8, 4096, device=DEV) for _ in range(2)]
    L.check(lib.lshm_conv_fwd_pair(2, P(row), P(wd[0]), P(bd[0]), P(z[0]), P(col), P(wd[1]), P(bd[1]), P(z[1]), nb, 4, 8, 1, 16384, 0, 0, 1,
                                   None, 0, st), "conv_fwd_pair")
    torch.cuda.synchronize()
    assert torch.equal(y[0], z[0]) and torch.equal(y[1], z[1])
    if keep:
        assert torch.equal(kept[0], row) and torch.equal(kept[1], col)
    r = (x.double() - x1.double()) / 2
    refT = TF.elu(TF.conv1d(r.reshape(nb, 4, -1), w[0].double(), b[0].double(), stride=4, padding=1))
    refF = TF.elu(TF.conv1d(r.transpose(2, 3).reshape(nb, 4, -1), w[1].double(), b[1].double(), stride=4, padding=1))
    assert rel_err(y[0], refT) < 1e-5 and rel_err(y[1], refF) < 1e-5


@pytest.mark.parametrize("nb", [256, 5, 8])
def test_first_1d_layers_backward_and_the_combination_as_one_launch(nb):
    """lshm_conv0_bwd_tile (weight, bias and data gradients of netT.conv0 and netF.conv0 and gx1 = gx1p - (dT + dF^T) / 2 from one
    pass over 32 x 64 image tiles; backward of src/lofar_models.py:115 on both networks, src/kharmonic_lofar.py:142-147) against the
    launches it replaces (lshm_conv_bwd_fused on each network, then lshm_combine_dx1) and against fp64.  The column-vectorised
    residual is NOT given to the new entry: netF's windows are read column-wise from the row image."""
    from lshm_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(500 + nb)
    r = torch.randn(nb, 4, 128, 128, generator=g)                    # the residual (x - x1) / 2 as the image
    dz = [torch.randn(nb, 8, 4096, generator=g) for _ in range(2)]  # netT, netF: gradients w.r.t. conv0's pre-activations
    w = [torch.randn(8, 4, 4, generator=g) * 0.3 for _ in range(2)]
    gx1p = torch.randn(nb, 4, 128, 128, generator=g)
    rd, gpd = r.to(DEV), gx1p.to(DEV)
    rcol = r.transpose(2, 3).contiguous().to(DEV)
    dzd, wd = [t.to(DEV) for t in dz], [t.to(DEV) for t in w]
    st, P = L.stream(), L.ptr
    nws = lib.lshm_conv0_bwd_tile_workspace_floats()
    ws = torch.empty(nws, device=DEV)
    dw = [torch.full((8, 4, 4), float("nan"), device=DEV) for _ in range(2)]
    db = [torch.full((8,), float("nan"), device=DEV) for _ in range(2)]
    gx1 = torch.full((nb, 4, 128, 128), float("nan"), device=DEV)
    L.check(lib.lshm_conv0_bwd_tile(P(rd), P(dzd[0]), P(dzd[1]), P(wd[0]), P(wd[1]), P(gpd), P(gx1), P(dw[0]), P(db[0]), P(dw[1]), P(db[1]),
                                    nb, P(ws), nws, 0, 0, st), "conv0_bwd_tile")
    # the replaced launches
    nws2 = lib.lshm_conv_workspace_floats(2, nb, 4, 8, 1, 16384)
    ws2 = torch.empty(nws2, device=DEV)
    dw2 = [torch.empty(8, 4, 4, device=DEV) for _ in range(2)]
    db2 = [torch.empty(8, device=DEV) for _ in range(2)]
    dx2 = [torch.empty(nb, 4, 16384, device=DEV) for _ in range(2)]
    for k, xin in enumerate((rd, rcol)):
        L.check(lib.lshm_conv_bwd_fused(2, P(xin), P(dzd[k]), P(wd[k]), P(dw2[k]), P(db2[k]), P(dx2[k]), 0, nb, 4, 8, 1, 16384, P(ws2), nws2,
                                        st), "conv_bwd_fused")
    gx1_2 = torch.empty_like(gx1)
    L.check(lib.lshm_combine_dx1(P(gpd), P(dx2[0]), P(dx2[1]), P(gx1_2), nb * 4, 128, st), "combine_dx1")
    torch.cuda.synchronize()
    assert torch.isfinite(gx1).all()
    for k in range(2):
        assert rel_err(dw[k], dw2[k]) < 2e-6 and rel_err(db[k], db2[k]) < 2e-6
    assert rel_err(gx1, gx1_2) < 2e-6
    # fp64
    seqs = (r.double().reshape(nb, 4, -1), r.double().transpose(2, 3).reshape(nb, 4, -1))
    gref = gx1p.double().clone()
    for k in range(2):
        xs = seqs[k].clone().requires_grad_(True)
        wr = w[k].double().requires_grad_(True)
        br = torch.zeros(8, dtype=torch.float64, requires_grad=True)
        TF.conv1d(xs, wr, br, stride=4, padding=1).backward(dz[k].double())
        assert rel_err(dw[k], wr.grad) < 1e-4 and rel_err(db[k], br.grad) < 1e-4
        d = xs.grad.reshape(nb, 4, 128, 128)
        gref -= 0.5 * (d if k == 0 else d.transpose(2, 3))
    assert rel_err(gx1, gref) < 1e-5


@pytest.mark.parametrize("nb", [256, 2])
def test_reconstruction_pass_from_the_last_layers_input_is_bitwise_the_two_launches(nb):
    """lshm_recon_losses_from_a (the reconstruction terms and gradients with x2 / x3c formed inside the pass from the input of
    netT / netF's last layer, src/lofar_models.py:142, src/kharmonic_lofar.py:150-158) against lshm_conv_fwd_pair of that
    layer followed by lshm_recon_losses_fwd_bwd: the seven sums and the three gradient images, bit for bit."""
    from lshm_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(11 + nb)
    C, P = 4, 128
    planes = nb * C
    x, x1 = torch.randn(nb, C, P, P, generator=g), torch.randn(nb, C, P, P, generator=g)
    a = [TF.elu(torch.randn(nb, 8, P * P // 4, generator=g)) for _ in range(2)]
    w = [torch.randn(8, C, 4, generator=g) * 0.3 for _ in range(2)]
    b = [torch.randn(C, generator=g) * 0.1 for _ in range(2)]
    ys = [torch.randn(nb, C, P, P, generator=g) * 0.1 for _ in range(3)]
    dv = lambda t: t.to(DEV).contiguous()
    xd, x1d, ad, wd, bd, yd = dv(x), dv(x1), [dv(t) for t in a], [dv(t) for t in w], [dv(t) for t in b], [dv(t) for t in ys]
    st, Pt = L.stream(), L.ptr
    nws = lib.lshm_recon_workspace_floats(planes, P)
    ws = torch.empty(nws, device=DEV)
    # reference: run the layer, then the pass
    x2, x3c = torch.empty(nb, C, P * P, device=DEV), torch.empty(nb, C, P * P, device=DEV)
    L.check(lib.lshm_conv_fwd_pair(3, Pt(ad[0]), Pt(wd[0]), Pt(bd[0]), Pt(x2), Pt(ad[1]), Pt(wd[1]), Pt(bd[1]), Pt(x3c), nb, 8, C, 1,
                                   P * P // 4, 0, 0, 0, None, 0, st), "tconv5 pair")
    s_ref = torch.zeros(8, device=DEV, dtype=torch.float64)
    g_ref = [torch.empty(nb, C, P, P, device=DEV) for _ in range(3)]
    L.check(lib.lshm_recon_losses_fwd_bwd(Pt(xd), Pt(x1d), Pt(x2), Pt(x3c), Pt(yd[0]), Pt(yd[1]), Pt(yd[2]), 0.7, planes, P, Pt(s_ref),
                                          *[Pt(t) for t in g_ref], Pt(ws), st), "recon")
    s_new = torch.zeros(8, device=DEV, dtype=torch.float64)
    g_new = [torch.full((nb, C, P, P), float("nan"), device=DEV) for _ in range(3)]
    L.check(lib.lshm_recon_losses_from_a(Pt(xd), Pt(x1d), Pt(ad[0]), Pt(ad[1]), Pt(wd[0]), Pt(bd[0]), Pt(wd[1]), Pt(bd[1]), Pt(yd[0]),
                                         Pt(yd[1]), Pt(yd[2]), 0.7, planes, P, C, Pt(s_new), *[Pt(t) for t in g_new], Pt(ws), st),
            "recon from a")
    torch.cuda.synchronize()
    assert torch.equal(s_ref[:7], s_new[:7])
    for p, q in zip(g_ref, g_new):
        assert torch.equal(p, q)


@pytest.mark.parametrize("nb", [256, 3, 8])
def test_reconstruction_pass_with_the_last_1d_layers_backward_inside(nb):
    """lshm_recon_bwd5 (multiplier update, reconstruction terms and -- instead of two of the three gradient images -- the weight,
    bias and data gradients of netT / netF's last layer, src/lofar_models.py:142, src/kharmonic_lofar.py:150-158,200-202) against
    the launches it replaces: lshm_conv_fwd_pair of that layer, lshm_multiplier_update, lshm_recon_losses_fwd_bwd, then
    lshm_conv_bwd_fused of the layer on each network.  Multipliers, the seven sums and gx1_partial bit for bit; the layer's
    gradients to fp32 summation order; and against fp64."""
    from lshm_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(900 + nb)
    C, P, rho = 4, 128, 0.7
    planes = nb * C
    x, x1 = torch.randn(nb, C, P, P, generator=g), torch.randn(nb, C, P, P, generator=g)
    a = [TF.elu(torch.randn(nb, 8, P * P // 4, generator=g)) for _ in range(2)]
    w = [torch.randn(8, C, 4, generator=g) * 0.3 for _ in range(2)]
    b = [torch.randn(C, generator=g) * 0.1 for _ in range(2)]
    ys = [torch.randn(nb, C, P, P, generator=g) * 0.1 for _ in range(3)]
    dv = lambda t: t.to(DEV).contiguous()
    xd, x1d, ad, wd, bd = dv(x), dv(x1), [dv(t) for t in a], [dv(t) for t in w], [dv(t) for t in b]
    st, Pt = L.stream(), L.ptr
    # ---- the replaced launches
    y_ref = [dv(t) for t in ys]
    nws = max(lib.lshm_recon_workspace_floats(planes, P), lib.lshm_conv_workspace_floats(3, nb, 8, C, 1, P * P // 4))
    ws = torch.empty(nws, device=DEV)
    x2, x3c = torch.empty(nb, C, P * P, device=DEV), torch.empty(nb, C, P * P, device=DEV)
    L.check(lib.lshm_conv_fwd_pair(3, Pt(ad[0]), Pt(wd[0]), Pt(bd[0]), Pt(x2), Pt(ad[1]), Pt(wd[1]), Pt(bd[1]), Pt(x3c), nb, 8, C, 1,
                                   P * P // 4, 0, 0, 0, None, 0, st), "tconv5 pair")
    L.check(lib.lshm_multiplier_update(Pt(xd), Pt(x1d), Pt(x2), Pt(x3c), *[Pt(t) for t in y_ref], rho, planes, P, st), "multiplier update")
    s_ref = torch.zeros(8, device=DEV, dtype=torch.float64)
    g_ref = [torch.empty(nb, C, P * P, device=DEV) for _ in range(3)]
    L.check(lib.lshm_recon_losses_fwd_bwd(Pt(xd), Pt(x1d), Pt(x2), Pt(x3c), *[Pt(t) for t in y_ref], rho, planes, P, Pt(s_ref),
                                          *[Pt(t) for t in g_ref], Pt(ws), st), "recon")
    dw_ref = [torch.empty(8, C, 4, device=DEV) for _ in range(2)]
    db_ref = [torch.empty(C, device=DEV) for _ in range(2)]
    da_ref = [torch.empty(nb, 8, P * P // 4, device=DEV) for _ in range(2)]
    for k in range(2):
        L.check(lib.lshm_conv_bwd_fused(3, Pt(ad[k]), Pt(g_ref[1 + k]), Pt(wd[k]), Pt(dw_ref[k]), Pt(db_ref[k]), Pt(da_ref[k]), 1, nb, 8, C, 1,
                                        P * P // 4, Pt(ws), nws, st), "tconv5 backward")
    # ---- the one pass
    y_new = [dv(t) for t in ys]
    nws5 = lib.lshm_recon_bwd5_workspace_floats(nb)
    ws5 = torch.empty(nws5, device=DEV)
    s_new = torch.zeros(8, device=DEV, dtype=torch.float64)
    gx1p = torch.full((nb, C, P * P), float("nan"), device=DEV)
    da = [torch.full((nb, 8, P * P // 4), float("nan"), device=DEV) for _ in range(2)]
    dw = [torch.full((8, C, 4), float("nan"), device=DEV) for _ in range(2)]
    db = [torch.full((C,), float("nan"), device=DEV) for _ in range(2)]
    L.check(lib.lshm_recon_bwd5(Pt(xd), Pt(x1d), Pt(ad[0]), Pt(ad[1]), Pt(wd[0]), Pt(bd[0]), Pt(wd[1]), Pt(bd[1]), *[Pt(t) for t in y_new], rho,
                                nb, Pt(s_new), Pt(gx1p), Pt(da[0]), Pt(da[1]), Pt(dw[0]), Pt(db[0]), Pt(dw[1]), Pt(db[1]), Pt(ws5), nws5, 0, st),
            "recon_bwd5")
    torch.cuda.synchronize()
    for p, q in zip(y_ref, y_new):
        assert torch.equal(p, q)
    assert torch.equal(s_ref[:7], s_new[:7])
    assert torch.equal(g_ref[0], gx1p)
    for k in range(2):
        assert torch.isfinite(da[k]).all()
        assert rel_err(da[k], da_ref[k]) < 2e-6 and rel_err(dw[k], dw_ref[k]) < 2e-6 and rel_err(db[k], db_ref[k]) < 2e-6
    # ---- the layer's backward alone from the gradient images (what a schedule without the fused pass runs): the same bits
    da2 = [torch.full((nb, 8, P * P // 4), float("nan"), device=DEV) for _ in range(2)]
    dw2 = [torch.full((8, C, 4), float("nan"), device=DEV) for _ in range(2)]
    db2 = [torch.full((C,), float("nan"), device=DEV) for _ in range(2)]
    L.check(lib.lshm_tconv5_pair_bwd(Pt(g_ref[1]), Pt(g_ref[2]), Pt(ad[0]), Pt(ad[1]), Pt(wd[0]), Pt(wd[1]), Pt(da2[0]), Pt(da2[1]), Pt(dw2[0]),
                                     Pt(db2[0]), Pt(dw2[1]), Pt(db2[1]), nb, Pt(ws5), nws5, 0, st), "tconv5_pair_bwd")
    torch.cuda.synchronize()
    for k in range(2):
        assert torch.equal(da[k], da2[k]) and torch.equal(dw[k], dw2[k]) and torch.equal(db[k], db2[k])
    # ---- fp64 (the layer's backward from the reference pass's gradient images)
    for k in range(2):
        av = a[k].double().requires_grad_(True)
        wr = w[k].double().requires_grad_(True)
        br = b[k].double().requires_grad_(True)
        TF.conv_transpose1d(av, wr, br, stride=4).backward(g_ref[1 + k].cpu().double())
        assert rel_err(dw[k], wr.grad) < 1e-4 and rel_err(db[k], br.grad) < 1e-4
        eg = torch.where(a[k].double() > 0, torch.ones_like(av), a[k].double() + 1.0)
        assert rel_err(da[k], (av.grad * eg)) < 1e-5


@pytest.mark.parametrize("nb", [64, 3])
def test_tile_kernels_of_the_1d_pair_with_bf16_storage(nb):
    """lshm_conv0_bwd_tile, lshm_recon_bwd5 and lshm_tconv5_pair_bwd with storage_bf16 = 1 (BASELINE configs[2]: image-sized
    tensors are bf16 in HBM, arithmetic and the small results fp32): against fp64 on the bf16-rounded inputs, to bf16 rounding of
    the image-sized outputs."""
    from lshm_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(1200 + nb)
    C, P, rho = 4, 128, 0.7
    bf = lambda t: t.to(torch.bfloat16)
    dv = lambda t: t.to(DEV).contiguous()
    st, Pt = L.stream(), L.ptr
    # ---- backward of conv0 of both networks + the combination
    r = bf(torch.randn(nb, C, P, P, generator=g))
    dz = [bf(torch.randn(nb, 8, 4096, generator=g)) for _ in range(2)]
    w = [torch.randn(8, C, 4, generator=g) * 0.3 for _ in range(2)]
    gp = bf(torch.randn(nb, C, P, P, generator=g))
    rd, gpd, dzd, wd = dv(r), dv(gp), [dv(t) for t in dz], [dv(t) for t in w]
    nws = lib.lshm_conv0_bwd_tile_workspace_floats()
    ws = torch.empty(nws, device=DEV)
    dw = [torch.full((8, C, 4), float("nan"), device=DEV) for _ in range(2)]
    db = [torch.full((8,), float("nan"), device=DEV) for _ in range(2)]
    gx1 = torch.full((nb, C, P, P), float("nan"), device=DEV, dtype=torch.bfloat16)
    L.check(lib.lshm_conv0_bwd_tile(Pt(rd), Pt(dzd[0]), Pt(dzd[1]), Pt(wd[0]), Pt(wd[1]), Pt(gpd), Pt(gx1), Pt(dw[0]), Pt(db[0]), Pt(dw[1]), Pt(db[1]),
                                    nb, Pt(ws), nws, 0, 1, st), "conv0_bwd_tile bf16")
    torch.cuda.synchronize()
    seqs = (r.double().reshape(nb, C, -1), r.double().transpose(2, 3).reshape(nb, C, -1))
    gref = gp.double().clone()
    for k in range(2):
        xs = seqs[k].clone().requires_grad_(True)
        wr = w[k].double().requires_grad_(True)
        br = torch.zeros(8, dtype=torch.float64, requires_grad=True)
        TF.conv1d(xs, wr, br, stride=4, padding=1).backward(dz[k].double())
        assert rel_err(dw[k], wr.grad) < 1e-4 and rel_err(db[k], br.grad) < 1e-4
        d = xs.grad.reshape(nb, C, P, P)
        gref -= 0.5 * (d if k == 0 else d.transpose(2, 3))
    assert torch.isfinite(gx1.float()).all() and rel_err(gx1.float(), gref) < 6e-3
    # ---- reconstruction pass with the last 1-D layer's backward inside
    x = torch.randn(nb, C, P, P, generator=g)
    x1 = bf(torch.randn(nb, C, P, P, generator=g))
    a = [bf(TF.elu(torch.randn(nb, 8, P * P // 4, generator=g))) for _ in range(2)]
    w5 = [torch.randn(8, C, 4, generator=g) * 0.3 for _ in range(2)]
    b5 = [torch.randn(C, generator=g) * 0.1 for _ in range(2)]
    ys = [torch.randn(nb, C, P, P, generator=g) * 0.1 for _ in range(3)]
    xd, x1d, ad, w5d, b5d, yd = dv(x), dv(x1), [dv(t) for t in a], [dv(t) for t in w5], [dv(t) for t in b5], [dv(t) for t in ys]
    nws5 = lib.lshm_recon_bwd5_workspace_floats(nb)
    ws5 = torch.empty(nws5, device=DEV)
    s7 = torch.zeros(8, device=DEV, dtype=torch.float64)
    gx1p = torch.full((nb, C, P, P), float("nan"), device=DEV, dtype=torch.bfloat16)
    da = [torch.full((nb, 8, P * P // 4), float("nan"), device=DEV, dtype=torch.bfloat16) for _ in range(2)]
    dw5 = [torch.full((8, C, 4), float("nan"), device=DEV) for _ in range(2)]
    db5 = [torch.full((C,), float("nan"), device=DEV) for _ in range(2)]
    L.check(lib.lshm_recon_bwd5(Pt(xd), Pt(x1d), Pt(ad[0]), Pt(ad[1]), Pt(w5d[0]), Pt(b5d[0]), Pt(w5d[1]), Pt(b5d[1]), *[Pt(t) for t in yd], rho, nb,
                                Pt(s7), Pt(gx1p), Pt(da[0]), Pt(da[1]), Pt(dw5[0]), Pt(db5[0]), Pt(dw5[1]), Pt(db5[1]), Pt(ws5), nws5, 1, st),
            "recon_bwd5 bf16")
    torch.cuda.synchronize()
    # fp64 restatement of the pass (src/kharmonic_lofar.py:150-158,200-202) with the storage roundings of the separate launches
    rb = lambda t: t.to(torch.bfloat16).double()
    X, X1 = x.double(), x1.double()
    x2 = rb(TF.conv_transpose1d(a[0].double(), w5[0].double(), b5[0].double(), stride=4)).reshape(nb, C, P, P)
    x3 = rb(TF.conv_transpose1d(a[1].double(), w5[1].double(), b5[1].double(), stride=4)).reshape(nb, C, P, P).transpose(2, 3)
    r1 = X - X1; h = 0.5 * r1; r2 = h - x2; r3 = h - x3; e = X1 + x2 + x3 - X
    m = [ys[0].double() + rho * r1, ys[1].double() + rho * r2, ys[2].double() + rho * r3]
    for k in range(3):  # (a reconstruction within 1e-7 of a bf16 rounding boundary may round the other way in fp64: ~1e-5 of them)
        assert rel_err(yd[k], m[k]) < 3e-4
    n = float(nb * C * P * P)
    sums = torch.stack([(e * e).sum(), (m[0] * r1).sum(), (r1 * r1).sum(), (m[1] * r2).sum(), (r2 * r2).sum(), (m[2] * r3).sum(), (r3 * r3).sum()])
    assert rel_err(s7[:7].cpu(), sums) < 1e-4
    t2, t3 = m[1] + rho * r2, m[2] + rho * r3
    g1p = (2 * e - m[0] - rho * r1) / n - 0.5 * (t2 + t3) / n
    assert rel_err(gx1p.float(), g1p) < 6e-3
    g2, g3 = rb((2 * e - t2) / n), rb((2 * e - t3) / n)
    gim = (g2.reshape(nb, C, -1), g3.transpose(2, 3).reshape(nb, C, -1))
    for k in range(2):
        av = a[k].double().requires_grad_(True)
        wr = w5[k].double().requires_grad_(True)
        br = b5[k].double().requires_grad_(True)
        TF.conv_transpose1d(av, wr, br, stride=4).backward(gim[k])
        assert rel_err(dw5[k], wr.grad) < 2e-3 and rel_err(db5[k], br.grad) < 2e-3   # (the images' bf16 rounding is modelled, not reproduced bit for bit)
        eg = torch.where(a[k].double() > 0, torch.ones_like(av), a[k].double() + 1.0)
        assert rel_err(da[k].float(), av.grad * eg) < 6e-3
    # ---- the layer's backward alone from (modelled) bf16 gradient images agrees with the fused pass
    gx2d = dv(g2.reshape(nb, C, P * P).to(torch.bfloat16))
    gx3d = dv(g3.transpose(2, 3).reshape(nb, C, P * P).to(torch.bfloat16))
    da2 = [torch.full((nb, 8, P * P // 4), float("nan"), device=DEV, dtype=torch.bfloat16) for _ in range(2)]
    dw2 = [torch.full((8, C, 4), float("nan"), device=DEV) for _ in range(2)]
    db2 = [torch.full((C,), float("nan"), device=DEV) for _ in range(2)]
    L.check(lib.lshm_tconv5_pair_bwd(Pt(gx2d), Pt(gx3d), Pt(ad[0]), Pt(ad[1]), Pt(w5d[0]), Pt(w5d[1]), Pt(da2[0]), Pt(da2[1]), Pt(dw2[0]), Pt(db2[0]),
                                     Pt(dw2[1]), Pt(db2[1]), nb, Pt(ws5), nws5, 1, st), "tconv5_pair_bwd bf16")
    torch.cuda.synchronize()
    for k in range(2):
        assert rel_err(dw2[k], dw5[k]) < 2e-3 and rel_err(da2[k].float(), da[k].float()) < 6e-3


def _deep2d_problem(nb, seed):
    """Random weights (torch layouts) and inputs of the deep section of AutoEncoderCNN2(latent_dim=224, rica=True)."""
    g = torch.Generator().manual_seed(seed)
    L, hd = 224, 16
    rn = lambda *s, scale=1.0: torch.randn(*s, generator=g) * scale
    w = [rn(48, 24, 4, 4, scale=(3.0 / (24 * 16)) ** 0.5), rn(96, 48, 4, 4, scale=(3.0 / (48 * 16)) ** 0.5),
         rn(192, 96, 4, 4, scale=(3.0 / (96 * 16)) ** 0.5), rn(L, 768 + hd, scale=(3.0 / 784) ** 0.5),
         rn(L, L, scale=(3.0 / L) ** 0.5), rn(L, L, scale=(3.0 / L) ** 0.5), rn(768, L + hd, scale=(3.0 / 240) ** 0.5),
         rn(192, 96, 4, 4, scale=(3.0 / (192 * 4)) ** 0.5), rn(96, 48, 4, 4, scale=(3.0 / (96 * 4)) ** 0.5),
         rn(48, 24, 4, 4, scale=(3.0 / (48 * 4)) ** 0.5), rn(24, 12, 4, 4, scale=(3.0 / (24 * 4)) ** 0.5)]
    b = [rn(n, scale=0.1) for n in (48, 96, 192, L, L, L, 768, 96, 48, 24, 12)]
    x2 = TF.elu(rn(nb, 24, 16, 16))
    uv1 = TF.elu(rn(nb, hd))  # elu(fcuv1(uvh)), elu(fcuv3(uvh)): produced by another launch in the engine
    uv3 = TF.elu(rn(nb, hd))
    return w, b, x2, uv1, uv3


def _deep2d_reference(w, b, x2, uv1, uv3, dt, bf16_operands=False):
    """bf16_operands: the convolution stages multiply on the bf16 matrix cores -- their input activations are rounded to bf16
    (the dense stages multiply in fp32)."""
    c = lambda t: t.to(dt)
    q = (lambda t: t.to(torch.bfloat16).to(dt)) if bf16_operands else (lambda t: t)
    a3 = TF.elu(TF.conv2d(q(c(x2)), c(w[0]), c(b[0]), stride=2, padding=1))
    a4 = TF.elu(TF.conv2d(q(a3), c(w[1]), c(b[1]), stride=2, padding=1))
    a5 = TF.elu(TF.conv2d(q(a4), c(w[2]), c(b[2]), stride=2, padding=1))
    cat1 = torch.cat([a5.flatten(1), c(uv1)], 1)
    z1 = TF.elu(TF.linear(cat1, c(w[3]), c(b[3])))
    mu = TF.elu(TF.linear(z1, c(w[4]), c(b[4])))
    h = TF.elu(TF.linear(mu, c(w[5]), c(b[5])))
    cat3 = torch.cat([h, c(uv3)], 1)
    d0 = TF.linear(cat3, c(w[6]), c(b[6]))
    t0 = TF.elu(TF.conv_transpose2d(q(d0).view(-1, 192, 2, 2), c(w[7]), c(b[7]), stride=2, padding=1))
    t1 = TF.elu(TF.conv_transpose2d(q(t0), c(w[8]), c(b[8]), stride=2, padding=1))
    t2 = TF.elu(TF.conv_transpose2d(q(t1), c(w[9]), c(b[9]), stride=2, padding=1))
    t3 = TF.elu(TF.conv_transpose2d(q(t2), c(w[10]), c(b[10]), stride=2, padding=1))
    return [a3, a4, cat1, z1, mu, cat3, d0, t0, t1, t2, t3]


@pytest.mark.parametrize("variant,nb", [(0, 256), (1, 256), (2, 256), (0, 5), (1, 5), (4, 256), (5, 37)],
                         ids=["one-patch", "two-patches", "512-threads", "one-patch-ragged", "two-patches-ragged", "bf16-weights",
                              "bf16-weights-two-patches-ragged"])
def test_deep_section_of_the_2d_autoencoder_as_one_launch(variant, nb):
    """lshm_deep2d_fwd (conv3 -> conv4 -> conv5 -> fc1 -> fc2in -> fc2out -> fc3 -> tconv0 -> tconv1 -> tconv2 -> tconv3 of
    AutoEncoderCNN2, src/lofar_models.py:36-55,66-98, one workgroup per patch (or two), activations resident in LDS,
    weights streamed in fragment order) against the eleven separate launches it replaces and against fp64: every layer's
    output, B = 256 and a batch that does not fill the last workgroup."""
    import ctypes as C
    from lshm_amd import _lib as L
    lib = L.load()
    Ld, hd = 224, 16
    w, b, x2, uv1, uv3 = _deep2d_problem(nb, 7 + variant)
    wd, bd = [t.to(DEV) for t in w], [t.to(DEV) for t in b]
    x2d = x2.to(DEV)
    shapes = [(nb, 48, 8, 8), (nb, 96, 4, 4), (nb, 768 + hd), (nb, Ld), (nb, 256), (nb, Ld + hd), (nb, 768), (nb, 96, 4, 4),
              (nb, 48, 8, 8), (nb, 24, 16, 16), (nb, 12, 32, 32)]
    outs = [torch.full(s, float("nan"), device=DEV) for s in shapes]
    outs[2][:, 768:] = uv1.to(DEV)
    outs[5][:, Ld:] = uv3.to(DEV)
    packed = torch.empty(lib.lshm_deep2d_packed_floats(), device=DEV)
    arr = lambda ts: (C.c_void_p * 11)(*[t.data_ptr() for t in ts])
    L.check(lib.lshm_deep2d_fwd(L.ptr(x2d), arr(wd), arr(bd), arr(outs), 256, L.ptr(packed), nb, variant, None, L.stream()), "deep2d_fwd")
    torch.cuda.synchronize()
    got = list(outs)
    got[4] = outs[4][:, :Ld]
    for k, o in enumerate(got):
        assert torch.isfinite(o).all(), k
    # fp64 on whole samples (first / last patch, both patches of a two-patch workgroup)
    idx = [i for i in SAMPLES if i < nb] if nb > 5 else list(range(nb))
    # variants 4, 5 stream bf16 copies of the weights into the bf16 matrix instructions (v_mfma_f32_16x16x16_bf16 / 4x4x4_bf16: the
    # activations a convolution stage multiplies are rounded to bf16 too; fp32 sums, fp32 dense stages): fp64 with the same roundings.
    # An activation within fp32 rounding of a bf16 rounding boundary rounds the other way in fp64 (4e-3 of that element), and every
    # stage inherits the flips of the stages before it: 1e-5 at the first stage, 6e-4 at the ninth (a wrong index would give O(1))
    wref = [t.bfloat16().float() for t in w] if variant & 4 else w
    ref = _deep2d_reference(wref, b, x2[idx], uv1[idx], uv3[idx], torch.float64, bf16_operands=bool(variant & 4))
    for k, (o, r) in enumerate(zip(got, ref)):
        assert rel_err(o[idx].cpu().double().reshape(r.shape), r) < (2e-3 if variant & 4 else 2e-5), (variant, k)
    if variant & 4:
        return
    # the separate launches of the library (implicit GEMMs with their own summation order)
    st = L.stream()
    cur = x2d
    sep = []
    for li, (kind, cin, cout, hin) in enumerate([(0, 24, 48, 16), (0, 48, 96, 8), (0, 96, 192, 4)]):
        y = torch.empty(nb, cout, hin // 2, hin // 2, device=DEV)
        nws = lib.lshm_conv_workspace_floats(kind, nb, cin, cout, hin, hin)
        ws = torch.empty(max(nws, 1), device=DEV)
        L.check(lib.lshm_conv_fwd(kind, L.ptr(cur), L.ptr(wd[li]), L.ptr(bd[li]), L.ptr(y), nb, cin, cout, hin, hin, 0, 0, 1,
                                  L.ptr(ws), nws, st), "conv_fwd")
        sep.append(y)
        cur = y
    cat1 = torch.cat([cur.flatten(1), uv1.to(DEV)], 1).contiguous()
    sep[2] = cat1

    def lin(x, wi, act):
        K, N = x.shape[1], wd[wi].shape[0]
        y = torch.empty(nb, N, device=DEV)
        nws = lib.lshm_linear_workspace_floats(nb, K, N)
        ws = torch.empty(max(nws, 1), device=DEV)
        L.check(lib.lshm_linear_fwd(L.ptr(x), K, L.ptr(wd[wi]), L.ptr(bd[wi]), L.ptr(y), N, nb, K, N, act, L.ptr(ws), nws, st), "linear_fwd")
        return y

    z1 = lin(cat1, 3, 1)
    mu = lin(z1, 4, 1)
    cat3 = torch.cat([lin(mu, 5, 1), uv3.to(DEV)], 1).contiguous()
    d0 = lin(cat3, 6, 0)
    sep += [z1, mu, cat3, d0]
    cur = d0.view(nb, 192, 2, 2)
    for li, (cin, cout, hin) in zip((7, 8, 9, 10), [(192, 96, 2), (96, 48, 4), (48, 24, 8), (24, 12, 16)]):
        y = torch.empty(nb, cout, 2 * hin, 2 * hin, device=DEV)
        nws = lib.lshm_conv_workspace_floats(1, nb, cin, cout, hin, hin)
        ws = torch.empty(max(nws, 1), device=DEV)
        L.check(lib.lshm_conv_fwd(1, L.ptr(cur), L.ptr(wd[li]), L.ptr(bd[li]), L.ptr(y), nb, cin, cout, hin, hin, 0, 0, 1,
                                  L.ptr(ws), nws, st), "tconv_fwd")
        sep.append(y)
        cur = y
    torch.cuda.synchronize()
    for k, (o, r) in enumerate(zip(got, sep)):
        assert rel_err(o, r.reshape(o.shape)) < 1e-5, (variant, k)


def test_deep_section_variants_agree_bit_for_bit():
    """One, two patches per workgroup and the 512-thread form of lshm_deep2d_fwd take the same products in the same order."""
    import ctypes as C
    from lshm_amd import _lib as L
    lib = L.load()
    nb, Ld, hd = 37, 224, 16
    w, b, x2, uv1, uv3 = _deep2d_problem(nb, 99)
    wd, bd = [t.to(DEV) for t in w], [t.to(DEV) for t in b]
    x2d = x2.to(DEV)
    shapes = [(nb, 48, 8, 8), (nb, 96, 4, 4), (nb, 768 + hd), (nb, Ld), (nb, 256), (nb, Ld + hd), (nb, 768), (nb, 96, 4, 4),
              (nb, 48, 8, 8), (nb, 24, 16, 16), (nb, 12, 32, 32)]
    arr = lambda ts: (C.c_void_p * 11)(*[t.data_ptr() for t in ts])
    res = []
    for variant in (0, 1, 2):
        outs = [torch.zeros(s, device=DEV) for s in shapes]
        outs[2][:, 768:] = uv1.to(DEV)
        outs[5][:, Ld:] = uv3.to(DEV)
        packed = torch.empty(lib.lshm_deep2d_packed_floats(), device=DEV)
        L.check(lib.lshm_deep2d_fwd(L.ptr(x2d), arr(wd), arr(bd), arr(outs), 256, L.ptr(packed), nb, variant, None, L.stream()), "deep2d_fwd")
        torch.cuda.synchronize()
        res.append([o.clone() for o in outs])
    for k in range(11):
        assert torch.equal(res[0][k], res[1][k]) and torch.equal(res[0][k], res[2][k]), k


def _eg(y):
    """ELU'(x) from the saved output y = ELU(x)."""
    return torch.where(y > 0, torch.ones_like(y), y + 1.0)


def _deep2d_bwd_reference(w, saved, g_t2, gmu, dt, bf16_operands=False):
    """The data-gradient pass through tconv2 .. conv2 of AutoEncoderCNN2 written out: the data gradient of a k4 s2 p1
    transposed conv is the conv with the same tensor and vice versa (autograd of src/lofar_models.py:73-98).
    bf16_operands: the convolution stages' input gradients are rounded to bf16 (bf16 matrix instructions)."""
    c = lambda t: t.to(dt)
    q = (lambda t: t.to(torch.bfloat16).to(dt)) if bf16_operands else (lambda t: t)
    c2, c3, c4, c5, fc1, fc2in, fc2out, fc3, t0, t1, t2, _t3 = [c(t) for t in w]
    s_t1, s_t0, s_cat3, s_mu, s_z1, s_cat1, s_c4, s_c3, s_c2, s_c1 = [c(t) for t in saved]
    g_t1 = TF.conv2d(q(c(g_t2)), t2, None, stride=2, padding=1) * _eg(s_t1)
    g_t0 = TF.conv2d(q(g_t1), t1, None, stride=2, padding=1) * _eg(s_t0)
    g_d0 = TF.conv2d(q(g_t0), t0, None, stride=2, padding=1).flatten(1)
    g_cat3 = (g_d0 @ fc3) * _eg(s_cat3)
    g_mu = g_cat3[:, :224] @ fc2out
    if gmu is not None:
        g_mu = g_mu + c(gmu)
    g_mu = g_mu * _eg(s_mu)
    g_z1 = (g_mu @ fc2in) * _eg(s_z1)
    g_cat1 = (g_z1 @ fc1) * _eg(s_cat1)
    g_c4 = TF.conv_transpose2d(q(g_cat1[:, :768]).reshape(-1, 192, 2, 2), c5, None, stride=2, padding=1) * _eg(s_c4)
    g_c3 = TF.conv_transpose2d(q(g_c4), c4, None, stride=2, padding=1) * _eg(s_c3)
    g_c2 = TF.conv_transpose2d(q(g_c3), c3, None, stride=2, padding=1) * _eg(s_c2)
    g_c1 = TF.conv_transpose2d(q(g_c2), c2, None, stride=2, padding=1) * _eg(s_c1)
    return [g_t1, g_t0, g_d0, g_cat3, g_mu, g_z1, g_cat1, g_c4, g_c3, g_c2, g_c1]


def _deep2d_bwd_problem(nb, seed):
    g = torch.Generator().manual_seed(seed)
    L, hd = 224, 16
    rn = lambda *s, scale=1.0: torch.randn(*s, generator=g) * scale
    w = [rn(24, 12, 4, 4, scale=(3.0 / (24 * 4)) ** 0.5), rn(48, 24, 4, 4, scale=(3.0 / (48 * 4)) ** 0.5),
         rn(96, 48, 4, 4, scale=(3.0 / (96 * 4)) ** 0.5), rn(192, 96, 4, 4, scale=(3.0 / (192 * 4)) ** 0.5),
         rn(L, 768 + hd, scale=(3.0 / L) ** 0.5), rn(L, L, scale=(3.0 / L) ** 0.5), rn(L, L, scale=(3.0 / L) ** 0.5),
         rn(768, L + hd, scale=(3.0 / 768) ** 0.5), rn(192, 96, 4, 4, scale=(3.0 / (96 * 16)) ** 0.5),
         rn(96, 48, 4, 4, scale=(3.0 / (48 * 16)) ** 0.5), rn(48, 24, 4, 4, scale=(3.0 / (24 * 16)) ** 0.5), rn(24, 12, 4, 4)]
    shapes = [(nb, 48, 8, 8), (nb, 96, 4, 4), (nb, L + hd), (nb, L), (nb, L), (nb, 768 + hd), (nb, 96, 4, 4), (nb, 48, 8, 8),
              (nb, 24, 16, 16), (nb, 12, 32, 32)]
    saved = [TF.elu(rn(*s)) for s in shapes]
    g_t2 = rn(nb, 24, 16, 16)
    gmu = rn(nb, L)
    return w, saved, g_t2, gmu


@pytest.mark.parametrize("variant,nb", [(0, 256), (1, 256), (0, 5), (1, 5), (4, 37)],
                         ids=["one-patch", "two-patches", "one-patch-ragged", "two-patches-ragged", "bf16-weights"])
def test_deep_section_backward_as_one_launch(variant, nb):
    """lshm_deep2d_bwd: the data gradients of tconv2, tconv1, tconv0, fc3, fc2out (+ the latent-term gradient), fc2in, fc1,
    conv5, conv4, conv3, conv2 of AutoEncoderCNN2 (autograd of src/lofar_models.py:73-98) from ONE launch -- the forward's
    eleven-stage pipeline on the layers' own weight tensors -- against fp64 and against the separate data-gradient launches
    of the library, every stage's output."""
    import ctypes as C
    from lshm_amd import _lib as L
    lib = L.load()
    Ld, hd = 224, 16
    w, saved, g_t2, gmu = _deep2d_bwd_problem(nb, 31 + variant)
    wd, sd = [t.to(DEV) for t in w], [t.to(DEV) for t in saved]
    mu_buf = torch.zeros(nb, 256, device=DEV)  # the code lives inside the shared latent matrix (row pitch 256)
    mu_buf[:, :Ld] = sd[3]
    sd_call = list(sd)
    sd_call[3] = mu_buf
    gd, gmud = g_t2.to(DEV), gmu.to(DEV)
    oshapes = [(nb, 48, 8, 8), (nb, 96, 4, 4), (nb, 768), (nb, Ld + hd), (nb, Ld), (nb, Ld), (nb, 768 + hd), (nb, 96, 4, 4), (nb, 48, 8, 8),
               (nb, 24, 16, 16), (nb, 12, 32, 32)]
    packed = torch.empty(lib.lshm_deep2d_packed_floats(), device=DEV)
    arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])

    def run(with_gmu):
        outs = [torch.full(s, float("nan"), device=DEV) for s in oshapes]
        L.check(lib.lshm_deep2d_bwd(L.ptr(gd), arr(wd), arr(sd_call), 256, L.ptr(gmud) if with_gmu else None, Ld, arr(outs), L.ptr(packed),
                                    nb, variant, L.stream()), "deep2d_bwd")
        torch.cuda.synchronize()
        return outs

    got = run(True)
    for k, o in enumerate(got):
        assert torch.isfinite(o).all(), k
    idx = [i for i in SAMPLES if i < nb] if nb > 5 else list(range(nb))
    wref = [t.bfloat16().float() for t in w] if variant & 4 else w  # (variant 4: bf16 copies of the weights are streamed)
    ref = _deep2d_bwd_reference(wref, [t[idx] for t in saved], g_t2[idx], gmu[idx], torch.float64, bf16_operands=bool(variant & 4))
    for k, (o, r) in enumerate(zip(got, ref)):
        assert rel_err(o[idx].cpu().double().reshape(r.shape), r) < (2e-3 if variant & 4 else 3e-5), (variant, k)
    if variant & 4:
        return
    # the separate launches of the library (no latent-term gradient: the C entry of the dense data gradient has no addend)
    got0 = run(False)
    st = L.stream()
    sep = []
    cur = gd
    for wi, si, (cin, cout, hin) in [(10, 0, (48, 24, 8)), (9, 1, (96, 48, 4)), (8, None, (192, 96, 2))]:  # data gradients of tconv2, tconv1, tconv0
        dx = torch.empty(nb, cin, hin, hin, device=DEV)
        nws = lib.lshm_conv_workspace_floats(1, nb, cin, cout, hin, hin)
        ws = torch.empty(max(nws, 1), device=DEV)
        L.check(lib.lshm_conv_dgrad(1, L.ptr(cur), L.ptr(wd[wi]), L.ptr(dx), L.ptr(sd[si]) if si is not None else None, nb, cin, cout, hin, hin,
                                    0, 0, L.ptr(ws), nws, st), "tconv_dgrad")
        sep.append(dx)
        cur = dx
    sep[2] = sep[2].reshape(nb, 768)

    def lin_dgrad(dz, lddz, wi, K, N, xs, ldxs):
        dx = torch.empty(nb, K, device=DEV)
        nws = lib.lshm_linear_workspace_floats(nb, K, N)
        ws = torch.empty(max(nws, 1), device=DEV)
        L.check(lib.lshm_linear_dgrad(L.ptr(dz), lddz, L.ptr(wd[wi]), L.ptr(dx), K, L.ptr(xs), ldxs, nb, K, N, L.ptr(ws), nws, st), "linear_dgrad")
        return dx

    g_cat3 = lin_dgrad(sep[2], 768, 7, Ld + hd, 768, sd[2], Ld + hd)
    g_mu = lin_dgrad(g_cat3, Ld + hd, 6, Ld, Ld, mu_buf, 256)
    g_z1 = lin_dgrad(g_mu, Ld, 5, Ld, Ld, sd[4], Ld)
    g_cat1 = lin_dgrad(g_z1, Ld, 4, 768 + hd, Ld, sd[5], 768 + hd)
    sep += [g_cat3, g_mu, g_z1, g_cat1]
    cur = g_cat1  # (B, 784): the conv5 gradient is its first 768 columns, batch stride 784
    in_bs = 768 + hd
    for wi, si, (cin, cout, hin) in [(3, 6, (96, 192, 4)), (2, 7, (48, 96, 8)), (1, 8, (24, 48, 16)), (0, 9, (12, 24, 32))]:  # conv5 .. conv2
        dx = torch.empty(nb, cin, hin, hin, device=DEV)
        nws = lib.lshm_conv_workspace_floats(0, nb, cin, cout, hin, hin)
        ws = torch.empty(max(nws, 1), device=DEV)
        L.check(lib.lshm_conv_dgrad(0, L.ptr(cur), L.ptr(wd[wi]), L.ptr(dx), L.ptr(sd[si]), nb, cin, cout, hin, hin, 0, in_bs, L.ptr(ws),
                                    nws, st), "conv_dgrad")
        sep.append(dx)
        cur = dx
        in_bs = 0
    torch.cuda.synchronize()
    for k, (o, r) in enumerate(zip(got0, sep)):
        assert rel_err(o, r.reshape(o.shape)) < 2e-5, (variant, k)


def test_whole_1d_mid_and_deep_section_as_one_launch():
    """lshm_chain1d_full_fwd (conv2 .. tconv3 of AutoEncoder1DCNN, src/lofar_models.py:119-140,158-183, one workgroup per
    patch; an opt-in experiment of the engine, LSHM_SCHED_TRY_FULL1D) against fp64, every layer's output, B = 37."""
    import ctypes as C
    from lshm_amd import _lib as L
    lib = L.load()
    nb = 37
    g = torch.Generator().manual_seed(5)
    ch = [12, 24, 48, 96, 192]
    w = [torch.randn(ch[i + 1], ch[i], 4, generator=g) * (3.0 / (4 * ch[i])) ** 0.5 for i in range(4)]
    w += [torch.randn(16, 784, generator=g) * 0.05, torch.randn(16, 16, generator=g) * 0.3, torch.randn(16, 16, generator=g) * 0.3,
          torch.randn(768, 32, generator=g) * 0.2]
    w += [torch.randn(ch[4 - i], ch[3 - i], 4, generator=g) * (3.0 / ch[4 - i]) ** 0.5 for i in range(4)]
    bs = [torch.randn(n, generator=g) * 0.1 for n in (24, 48, 96, 192, 16, 16, 16, 768, 96, 48, 24, 12)]
    x1 = TF.elu(torch.randn(nb, 12, 1024, generator=g))
    uv1, uv3 = TF.elu(torch.randn(nb, 16, generator=g)), TF.elu(torch.randn(nb, 16, generator=g))
    shapes = [(nb, 24, 256), (nb, 48, 64), (nb, 96, 16), (nb, 784), (nb, 16), (nb, 256), (nb, 32), (nb, 768), (nb, 96, 16), (nb, 48, 64),
              (nb, 24, 256), (nb, 12, 1024)]
    outs = [torch.full(s, float("nan"), device=DEV) for s in shapes]
    outs[3][:, 768:] = uv1.to(DEV)
    outs[6][:, 16:] = uv3.to(DEV)
    wd, bd = [t.to(DEV) for t in w], [t.to(DEV) for t in bs]
    arr = lambda ts: (C.c_void_p * 12)(*[t.data_ptr() for t in ts])
    L.check(lib.lshm_chain1d_full_fwd(L.ptr(x1.to(DEV)), arr(wd), arr(bd), arr(outs), 256, nb, None, L.stream()), "chain1d_full_fwd")
    torch.cuda.synchronize()
    d = lambda t: t.double()
    cur = d(x1)
    ref = []
    for i in range(3):
        cur = TF.elu(TF.conv1d(cur, d(w[i]), d(bs[i]), stride=4, padding=1))
        ref.append(cur)
    a5 = TF.elu(TF.conv1d(cur, d(w[3]), d(bs[3]), stride=4, padding=1))
    cat1 = torch.cat([a5.flatten(1), d(uv1)], 1)
    z1 = TF.elu(TF.linear(cat1, d(w[4]), d(bs[4])))
    mu = TF.elu(TF.linear(z1, d(w[5]), d(bs[5])))
    cat3 = torch.cat([TF.elu(TF.linear(mu, d(w[6]), d(bs[6]))), d(uv3)], 1)
    d0 = TF.linear(cat3, d(w[7]), d(bs[7]))
    ref += [cat1, z1, mu, cat3, d0]
    cur = d0.view(nb, 192, 4)
    for i in range(4):
        cur = TF.elu(TF.conv_transpose1d(cur, d(w[8 + i]), d(bs[8 + i]), stride=4))
        ref.append(cur)
    got = list(outs)
    got[5] = outs[5][:, :16]
    for k, (o, r) in enumerate(zip(got, ref)):
        assert torch.isfinite(o).all(), k
        assert rel_err(o.cpu().double().reshape(r.shape), r) < 2e-5, k
