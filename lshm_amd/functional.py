"""torch.autograd wrappers around the HIP kernels (one Function per fused layer).

These give the drop-in modules of ``lofar_models.py`` ordinary autograd
semantics, so a user closure written against the reference
(src/kharmonic_lofar.py:132-182) back-propagates through them unchanged.
Every op calls the C ABI of liblshm_hip.so on torch's current stream; there is
no eager/PyTorch fallback.
"""
from __future__ import annotations

import functools

import torch

from . import _lib as L

CONV2D, TCONV2D, CONV1D, TCONV1D = 0, 1, 2, 3
EPS_KHM = 1e-9


def _on_tensor_device(f):
    """Run `f` with the device of its first tensor argument current: HIP launches go to the current
    device, whatever device the pointers and the stream belong to."""
    @functools.wraps(f)
    def wrapped(*args, **kw):
        for t in args:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                with torch.cuda.device(t.device):
                    return f(*args, **kw)
        return f(*args, **kw)
    return wrapped


def _geom(kind: int, x: torch.Tensor, w: torch.Tensor):
    """(B, Cin, Cout, Hin, Win, out_shape) for a layer of the given kind."""
    B, Cin = x.shape[0], x.shape[1]
    if kind in (CONV2D, TCONV2D):
        if x.dim() != 4 or w.dim() != 4 or tuple(w.shape[2:]) != (4, 4):
            raise RuntimeError(f"expected NCHW input and a 4x4 kernel, got {tuple(x.shape)} / {tuple(w.shape)}")
        Hin, Win = x.shape[2], x.shape[3]
    else:
        if x.dim() != 3 or w.dim() != 3 or w.shape[2] != 4:
            raise RuntimeError(f"expected NCL input and a width-4 kernel, got {tuple(x.shape)} / {tuple(w.shape)}")
        Hin, Win = 1, x.shape[2]
    if kind in (CONV2D, CONV1D):
        Cout, wcin = w.shape[0], w.shape[1]
    else:
        Cout, wcin = w.shape[1], w.shape[0]
    if wcin != Cin:
        # same failure class as torch: a shape mismatch is a RuntimeError
        raise RuntimeError(f"weight expects {wcin} input channels, input has {Cin}")
    if kind == CONV2D:
        if Hin % 4 or Win % 4:
            raise RuntimeError("conv2d k4s2p1 kernel needs H and W to be multiples of 4")
        out = (B, Cout, Hin // 2, Win // 2)
    elif kind == TCONV2D:
        out = (B, Cout, Hin * 2, Win * 2)
    elif kind == CONV1D:
        if Win % 16:
            raise RuntimeError("conv1d k4s4p1 kernel needs L to be a multiple of 16")
        out = (B, Cout, (Win - 2) // 4 + 1)
    else:
        out = (B, Cout, Win * 4)
    return B, Cin, Cout, Hin, Win, out


class _ConvAct(torch.autograd.Function):
    """y = act(conv(x, w) + b) for the four conv flavours (src/lofar_models.py:73-78,93-98,158-163,178-183)."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, x, w, b, kind, act, bf16=False):
        L.require_device(x, w, b)
        x = x.contiguous()
        w = w.contiguous()
        lib = L.load()
        B, Cin, Cout, Hin, Win, oshape = _geom(kind, x, w)
        y = torch.empty(oshape, device=x.device, dtype=torch.float32)
        if B > 0:
            nws = lib.lshm_conv_workspace_floats(kind, B, Cin, Cout, Hin, Win)
            ws = L.scratch(x.device, nws)
            L.check(L.fn("lshm_conv_fwd", bf16)(kind, L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), B, Cin, Cout, Hin, Win,
                                                0, 0, int(act), L.ptr(ws), ws.numel(), L.stream()), "conv_fwd")
        ctx.save_for_backward(x, w, y)
        ctx.kind, ctx.act, ctx.has_bias, ctx.bf16 = kind, act, b is not None, bool(bf16)
        return y

    @staticmethod
    @_on_tensor_device
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        lib = L.load()
        kind = ctx.kind
        B, Cin, Cout, Hin, Win, _ = _geom(kind, x, w)
        gy = gy.contiguous()
        st = L.stream()
        if ctx.act:
            dz = torch.empty_like(gy)
            L.check(lib.lshm_elu_bwd(L.ptr(gy), L.ptr(y), L.ptr(dz), dz.numel(), st), "elu_bwd")
        else:
            dz = gy
        dx = dw = db = None
        if B == 0:
            return (torch.zeros_like(x), torch.zeros_like(w),
                    torch.zeros(Cout, device=x.device) if ctx.has_bias else None, None, None, None)
        nws = lib.lshm_conv_workspace_floats(kind, B, Cin, Cout, Hin, Win)
        ws = L.scratch(x.device, nws)
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            L.check(L.fn("lshm_conv_dgrad", ctx.bf16)(kind, L.ptr(dz), L.ptr(w), L.ptr(dx), None, B, Cin, Cout, Hin, Win,
                                                      0, 0, L.ptr(ws), ws.numel(), st), "conv_dgrad")
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty_like(w)
            db = torch.empty(Cout, device=x.device, dtype=torch.float32) if ctx.has_bias else None
            L.check(L.fn("lshm_conv_wgrad", ctx.bf16)(kind, L.ptr(x), L.ptr(dz), L.ptr(dw), L.ptr(db), B, Cin, Cout,
                                                      Hin, Win, 0, 0, L.ptr(ws), ws.numel(), 0, st), "conv_wgrad")
        return dx, dw, db, None, None, None


def conv_act(x, w, b, kind: int, act: bool, bf16: bool = False):
    """bf16: operands of the GEMM-shaped layers rounded to bf16 for this call (fp32 accumulation / storage)."""
    return _ConvAct.apply(x, w, b, kind, act, bf16)


class _LinearAct(torch.autograd.Function):
    """y = act(x @ w.T + b) (src/lofar_models.py:80-83,89-91,67-68)."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, x, w, b, act, bf16=False):
        L.require_device(x, w, b)
        if x.dim() != 2:
            raise RuntimeError("linear expects a 2-D input")
        x = x.contiguous()
        w = w.contiguous()
        B, K = x.shape
        N = w.shape[0]
        if w.shape[1] != K:
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({B}x{K} and {w.shape[1]}x{N})")
        y = torch.empty((B, N), device=x.device, dtype=torch.float32)
        if B > 0:
            lib = L.load()
            ws = L.scratch(x.device, lib.lshm_linear_workspace_floats(B, K, N))
            L.check(L.fn("lshm_linear_fwd", bf16)(L.ptr(x), K, L.ptr(w), L.ptr(b), L.ptr(y), N, B, K, N, int(act),
                                                  L.ptr(ws), ws.numel(), L.stream()), "linear_fwd")
        ctx.save_for_backward(x, w, y)
        ctx.act, ctx.has_bias, ctx.bf16 = act, b is not None, bool(bf16)
        return y

    @staticmethod
    @_on_tensor_device
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        lib = L.load()
        B, K = x.shape
        N = w.shape[0]
        st = L.stream()
        gy = gy.contiguous()
        if B == 0:
            return (torch.zeros_like(x), torch.zeros_like(w),
                    (torch.zeros(N, device=x.device) if ctx.has_bias else None), None, None)
        if ctx.act:
            dz = torch.empty_like(gy)
            L.check(lib.lshm_elu_bwd(L.ptr(gy), L.ptr(y), L.ptr(dz), dz.numel(), st), "elu_bwd")
        else:
            dz = gy
        dx = dw = db = None
        ws = L.scratch(x.device, lib.lshm_linear_workspace_floats(B, K, N))
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            L.check(L.fn("lshm_linear_dgrad", ctx.bf16)(L.ptr(dz), N, L.ptr(w), L.ptr(dx), K, None, 0, B, K, N,
                                                        L.ptr(ws), ws.numel(), st), "linear_dgrad")
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty_like(w)
            db = torch.empty(N, device=x.device, dtype=torch.float32) if ctx.has_bias else None
            L.check(L.fn("lshm_linear_wgrad", ctx.bf16)(L.ptr(x), K, L.ptr(dz), N, L.ptr(dw), L.ptr(db), B, K, N,
                                                        L.ptr(ws), ws.numel(), st), "linear_wgrad")
        return dx, dw, db, None, None


def linear_act(x, w, b, act: bool, bf16: bool = False):
    return _LinearAct.apply(x, w, b, act, bf16)


@_on_tensor_device
def uv_harmonics(scales: torch.Tensor, uv: torch.Tensor) -> torch.Tensor:
    """kron(scales, uv) -> cat(sin, cos) (src/lofar_models.py:60-62).  Not differentiated:
    uv are data and the scales are a plain attribute upstream."""
    L.require_device(scales, uv)
    uv = uv.detach().contiguous()
    scales = scales.detach().contiguous()
    if uv.dim() != 2 or uv.shape[1] != 2:
        raise RuntimeError("uv must have shape (B, 2)")
    H, B = scales.numel(), uv.shape[0]
    out = torch.empty((B, 4 * H), device=uv.device, dtype=torch.float32)
    L.check(L.load().lshm_uv_harmonics(L.ptr(uv), L.ptr(scales), H, B, L.ptr(out), L.stream()), "uv_harmonics")
    return out


class _KHMLoss(torch.autograd.Function):
    """Kmeans.forward (src/lofar_models.py:199-209): loss and both gradients in one fused pass."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, X, M, p, eps):
        L.require_device(X, M)
        X = X.contiguous()
        M = M.contiguous()
        N, D = X.shape
        K = M.shape[0]
        if M.shape[1] != D:
            raise RuntimeError(f"The size of tensor a ({M.shape[1]}) must match the size of tensor b ({D})")
        lib = L.load()
        if N == 0:
            raise ZeroDivisionError("division by zero")  # upstream divides by nbatch*K*latent_dim
        nws = lib.lshm_khm_workspace_floats(N, D, K)
        ws = torch.empty(nws, device=X.device, dtype=torch.float32)
        loss = torch.empty(1, device=X.device, dtype=torch.float64)
        dX = torch.empty_like(X)
        dM = torch.empty_like(M)
        inv = 1.0 / (float(N) * K * D)
        L.check(lib.lshm_khm_fwd_bwd(L.ptr(X), D, L.ptr(M), N, D, K, float(p), float(eps), inv, 1.0,
                                     L.ptr(loss), L.ptr(dX), D, L.ptr(dM), 0, L.ptr(ws), nws, L.stream()),
                "khm_fwd_bwd")
        ctx.save_for_backward(dX, dM)
        return (loss[0] * inv).to(torch.float32)

    @staticmethod
    @_on_tensor_device
    def backward(ctx, g):
        dX, dM = ctx.saved_tensors
        return (g * dX if ctx.needs_input_grad[0] else None,
                g * dM if ctx.needs_input_grad[1] else None, None, None)


def khm_loss(X, M, p, eps=EPS_KHM):
    return _KHMLoss.apply(X, M, p, eps)


class _ClusterSim(torch.autograd.Function):
    """Kmeans.cluster_similarity (src/lofar_models.py:214-229)."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, M, eps):
        L.require_device(M)
        M = M.contiguous()
        K, D = M.shape
        loss = torch.empty(1, device=M.device, dtype=torch.float64)
        dM = torch.empty_like(M)
        L.check(L.load().lshm_cluster_sim_fwd_bwd(L.ptr(M), K, D, float(eps), 1.0, L.ptr(loss), L.ptr(dM), 0,
                                                  L.stream()), "cluster_sim")
        ctx.save_for_backward(dM)
        return loss[0].to(torch.float32)

    @staticmethod
    @_on_tensor_device
    def backward(ctx, g):
        (dM,) = ctx.saved_tensors
        return g * dM, None


def cluster_similarity(M, eps=EPS_KHM):
    return _ClusterSim.apply(M, eps)


class _AugLoss(torch.autograd.Function):
    """augmented_loss(mu, batch_per_bline, batch_size) (src/kharmonic_lofar.py:97-110); shape (1,)."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, Z, bpb, batch_size):
        L.require_device(Z)
        Z = Z.contiguous()
        rows, D = Z.shape
        groups = (rows + bpb - 1) // bpb
        loss = torch.zeros(groups + 2, device=Z.device, dtype=torch.float64)
        dZ = torch.empty_like(Z)
        L.check(L.load().lshm_aug_loss_fwd_bwd(L.ptr(Z), D, rows, D, int(bpb), int(batch_size), 1.0,
                                               L.ptr(loss), L.ptr(dZ), D, 0, L.stream()), "aug_loss")
        ctx.save_for_backward(dZ)
        return loss[:1].to(torch.float32)

    @staticmethod
    @_on_tensor_device
    def backward(ctx, g):
        (dZ,) = ctx.saved_tensors
        return g.reshape(()) * dZ, None, None


def augmented_loss(mu, batch_per_bline, batch_size):
    return _AugLoss.apply(mu, batch_per_bline, batch_size)


@_on_tensor_device
def khm_offline_partials(X, M, p, eps=EPS_KHM):
    """Numerator (K,D) and denominator (K) of Zhang's recursion (intent of
    Kmeans.offline_update, src/lofar_models.py:231-261)."""
    L.require_device(X, M)
    X = X.detach().contiguous()
    M = M.detach().contiguous()
    N, D = X.shape
    K = M.shape[0]
    lib = L.load()
    nws = lib.lshm_khm_workspace_floats(N, D, K)
    ws = torch.empty(nws, device=X.device, dtype=torch.float32)
    num = torch.empty((K, D), device=X.device, dtype=torch.float32)
    den = torch.empty(K, device=X.device, dtype=torch.float32)
    L.check(lib.lshm_khm_offline_partials(L.ptr(X), D, L.ptr(M), N, D, K, float(p), float(eps), L.ptr(num),
                                          L.ptr(den), L.ptr(ws), nws, L.stream()), "khm_offline_partials")
    return num, den


@_on_tensor_device
def khm_mean_distances(X, M, p):
    """dist[k] = mean_n ||X_n - M_k||^p (src/evaluate_clustering.py:111-115)."""
    L.require_device(X, M)
    X = X.detach().contiguous()
    M = M.detach().contiguous()
    N, D = X.shape
    K = M.shape[0]
    lib = L.load()
    nws = lib.lshm_khm_workspace_floats(N, D, K)
    ws = torch.empty(nws, device=X.device, dtype=torch.float32)
    dist = torch.empty(K, device=X.device, dtype=torch.float32)
    L.check(lib.lshm_khm_mean_distances(L.ptr(X), D, L.ptr(M), N, D, K, float(p), L.ptr(dist), L.ptr(ws), nws,
                                        L.stream()), "khm_mean_distances")
    return dist


@_on_tensor_device
def khm_assign(dist: torch.Tensor):
    """(argmin index (0-dim int64 tensor), softmax(-dist / dist.mean())) of a distance vector (K <= 64):
    the cluster id of src/evaluate_clustering.py:116-119 and the soft labels of src/train_graph_stat.py:206-210."""
    L.require_device(dist)
    dist = dist.detach().contiguous()
    K = dist.numel()
    idx = torch.empty(1, device=dist.device, dtype=torch.int32)
    prob = torch.empty(K, device=dist.device, dtype=torch.float32)
    L.check(L.load().lshm_khm_assign(L.ptr(dist), K, L.ptr(idx), L.ptr(prob), L.stream()), "khm_assign")
    return idx[0].to(torch.int64), prob


class _FFTFeatures(torch.autograd.Function):
    """fftn(dim=(2,3), ortho) -> fftshift -> cat(real, imag) -> clamp (Demo.ipynb:169-175) with its backward."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, r, clamp):
        L.require_device(r)
        r = r.contiguous()
        if r.dim() != 4 or r.shape[2] != 128 or r.shape[3] != 128:
            raise RuntimeError("fft_features expects (B, C, 128, 128)")
        B, Cc = r.shape[0], r.shape[1]
        out = torch.empty((B, 2 * Cc, 128, 128), device=r.device, dtype=torch.float32)
        L.check(L.load().lshm_fft2_ortho_shift_cat_clamp(L.ptr(r), L.ptr(out), B, Cc, float(clamp), L.stream()), "fft2")
        ctx.save_for_backward(out)
        ctx.clamp = float(clamp)
        return out

    @staticmethod
    @_on_tensor_device
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        B, Cc = out.shape[0], out.shape[1] // 2
        g = g.contiguous()
        lib = L.load()
        nws = lib.lshm_fft2_backward_workspace_floats(B, Cc)
        ws = torch.empty(nws, device=g.device, dtype=torch.float32)
        dx = torch.empty((B, Cc, 128, 128), device=g.device, dtype=torch.float32)
        L.check(lib.lshm_fft2_backward(L.ptr(g), L.ptr(out), L.ptr(dx), B, Cc, ctx.clamp, L.ptr(ws), nws, L.stream()),
                "fft2_backward")
        return dx, None


def fft_features(r: torch.Tensor, clamp: float = 10.0) -> torch.Tensor:
    """fftn(dim=(2,3), ortho) -> fftshift -> cat(real, imag) -> clamp (Demo.ipynb:169-175); differentiable."""
    return _FFTFeatures.apply(r, clamp)
