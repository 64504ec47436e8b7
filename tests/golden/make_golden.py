#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF on CPU.

Only runs in the build container (needs /root/reference, which never travels to
the GPU box).  It imports upstream ``src/lofar_models.py`` unmodified (the two
imported-but-unused ``torchvision`` names are satisfied with empty stub
modules), loads deterministic closed-form parameters / inputs (the same
``closed_form_*`` rules the tests use, so only *outputs* are stored) and records
outputs + gradients.  The training-step fixtures run a fresh restatement of the
closure of ``src/kharmonic_lofar.py:132-202`` on top of the *imported reference
modules* and ``torch.optim.Adam`` (the upstream script itself cannot be
imported: it needs h5py and LOFAR files at import time).

Usage:  python tests/golden/make_golden.py      (writes next to this file)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

REF = os.environ.get("LSHM_REFERENCE", "/root/reference/src")


def import_reference():
    for n in ("torchvision", "torchvision.transforms"):
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.path.insert(0, REF)
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))  # imported, used only by the H5 readers
    import lofar_models  # noqa
    import lofar_tools  # noqa
    return lofar_models, lofar_tools


def reference_augmented_loss():
    """The upstream script cannot be imported (it trains at import time), so the
    one function this path needs from it is compiled from the script's own AST,
    in memory, and run as-is: src/kharmonic_lofar.py:97-110."""
    import ast
    src = open(os.path.join(REF, "kharmonic_lofar.py")).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "augmented_loss"]
    mod = ast.Module(body=fn, type_ignores=[])
    env = {"torch": torch, "mydevice": torch.device("cpu")}
    exec(compile(mod, "kharmonic_lofar.py", "exec"), env)
    return env["augmented_loss"]


def probe(t: torch.Tensor, nsample: int = 512) -> dict:
    """Compact, discriminating summary of a tensor."""
    v = t.detach().double().reshape(-1).numpy()
    n = v.size
    idx = np.unique(np.linspace(0, n - 1, min(n, nsample)).astype(np.int64))
    w = np.sin(0.37 * np.arange(n) + 0.11)
    return {"n": np.int64(n), "sum": v.sum(), "asum": np.abs(v).sum(),
            "wsum": (v * w).sum(), "idx": idx, "sample": v[idx].astype(np.float32)}


def put(store: dict, name: str, t: torch.Tensor, full: bool = False):
    if full:
        store[name] = t.detach().float().numpy()
    else:
        for k, val in probe(t).items():
            store[f"{name}/{k}"] = val


def main():
    torch.manual_seed(0)
    torch.set_num_threads(max(1, (os.cpu_count() or 2)))
    lm, lt = import_reference()
    ref_aug = reference_augmented_loss()
    from oracle import lshm_oracle as O

    scales = torch.tensor(O.DEFAULT_SCALES)

    # ---------------- (7) harmonic features -------------------------------
    g = {}
    uv3 = torch.tensor([[1234.5, -987.25], [0.0, 3.5], [-20000.0, 15.125]])
    k = torch.kron(scales, uv3)
    put(g, "harmonics/uv", uv3, True)
    put(g, "harmonics/out", torch.cat((torch.sin(k), torch.cos(k)), 1), True)
    np.savez_compressed(os.path.join(HERE, "harmonics.npz"), **g)

    # ---------------- (1) per-op conv flavours ----------------------------
    g = {}
    ch = O.CH
    B = 2
    for i in range(6):
        for kind in ("conv2d", "tconv2d", "conv1d", "tconv1d"):
            nd = 2 if kind.endswith("2d") else 1
            tr = kind.startswith("t")
            if nd == 2:
                hw_in = 128 >> i if not tr else 2 << i
                ishape = (B, ch[i] if not tr else ch[6 - i], hw_in, hw_in)
            else:
                Ls = [16384, 4096, 1024, 256, 64, 16, 4]
                ishape = (B, ch[i] if not tr else ch[6 - i], Ls[i] if not tr else Ls[6 - i])
            cin = ishape[1]
            cout = ch[i + 1] if not tr else ch[5 - i]
            kk = (4, 4) if nd == 2 else (4,)
            wshape = ((cout, cin) if not tr else (cin, cout)) + kk
            key = f"{kind}{i}"
            x = O.closed_form(ishape, key + ":x", 1.0, 0.2113).requires_grad_(True)
            fan = (cin if not tr else cout) * int(np.prod(kk))
            w = O.closed_form(wshape, key + ":w", (3.0 / fan) ** 0.5).requires_grad_(True)
            b = O.closed_form((cout,), key + ":b", fan ** -0.5).requires_grad_(True)
            if kind == "conv2d":
                y = torch.nn.functional.conv2d(x, w, b, stride=2, padding=1)
            elif kind == "tconv2d":
                y = torch.nn.functional.conv_transpose2d(x, w, b, stride=2, padding=1)
            elif kind == "conv1d":
                y = torch.nn.functional.conv1d(x, w, b, stride=4, padding=1)
            else:
                y = torch.nn.functional.conv_transpose1d(x, w, b, stride=4, padding=0)
            act = torch.nn.functional.elu(y)
            gy = O.closed_form(tuple(y.shape), key + ":gy", 1.0, 0.3331)
            (act * gy).sum().backward()
            put(g, key + "/y", act)
            put(g, key + "/dx", x.grad)
            put(g, key + "/dw", w.grad)
            put(g, key + "/db", b.grad, True)
    np.savez_compressed(os.path.join(HERE, "convops.npz"), **g)

    # ---------------- (2) per-AE ------------------------------------------
    g = {}
    x2, uv2 = O.closed_form_inputs(2, 4)
    cases = [("ae2d_rica", lm.AutoEncoderCNN2, 224, 4, 2, True),
             ("ae2d_norica", lm.AutoEncoderCNN2, 224, 4, 2, False),
             ("ae1d_rica", lm.AutoEncoder1DCNN, 16, 4, 1, True),
             ("fnet8", lm.AutoEncoderCNN2, 64, 8, 2, False)]
    for name, cls, L, C, nd, rica in cases:
        net = cls(latent_dim=L, channels=C, harmonic_scales=scales, rica=rica)
        sd = O.closed_form_state_dict(L, C, nd, rica, name)
        net.load_state_dict(sd)
        xin, uvin = O.closed_form_inputs(2, C)
        if nd == 1:
            xin = xin.flatten(2, 3)
        xin = xin.clone().requires_grad_(True)
        xhat, mu = net(xin, uvin)
        gy = O.closed_form(tuple(xhat.shape), name + ":gy", 1.0, 0.3331)
        gm = O.closed_form(tuple(mu.shape), name + ":gm", 1.0, 0.7717)
        ((xhat * gy).sum() + (mu * gm).sum()).backward()
        put(g, name + "/xhat", xhat)
        put(g, name + "/mu", mu, True)
        put(g, name + "/dx", xin.grad)
        for pn, p in net.named_parameters():
            g[f"{name}/gnorm/{pn}"] = p.grad.double().norm().item()
            g[f"{name}/ghead/{pn}"] = p.grad.reshape(-1)[:8].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "autoencoders.npz"), **g)

    # ---------------- (3) KHM / similarity / augmented loss ---------------
    g = {}
    khm_cases = [(8, 4, 256, 4, False), (32, 10, 256, 4, False), (16, 10, 256, 2, False),
                 (16, 10, 256, 3, False), (64, 64, 256, 4, False), (8, 4, 256, 4, True)]
    for ci, (Bk, K, D, p, degenerate) in enumerate(khm_cases):
        mod = lm.Kmeans(latent_dim=D, K=K, p=p)
        M0 = 0.5 + 0.5 * O.closed_form((K, D), f"khm{ci}:M", 1.0, 0.618)
        X = (0.8 * O.closed_form((Bk, D), f"khm{ci}:X", 1.0, 0.4142) + 0.3)
        if degenerate:
            X[0] = M0[0]
        with torch.no_grad():
            mod.M.copy_(M0)
        X = X.clone().requires_grad_(True)
        loss = mod.clustering_error(X)
        g[f"khm{ci}/loss"] = loss.item()
        if not degenerate:
            # upstream's norm() has a NaN gradient at zero distance; the build's
            # (d^2)^(p/2) form is finite there, so only the loss is pinned
            loss.backward()
            put(g, f"khm{ci}/dX", X.grad, True)
            put(g, f"khm{ci}/dM", mod.M.grad, True)
        g[f"khm{ci}/cfg"] = np.array([Bk, K, D, p, int(degenerate)])
    for K in (4, 10, 64):
        mod = lm.Kmeans(latent_dim=256, K=K, p=4)
        with torch.no_grad():
            mod.M.copy_(0.5 + 0.5 * O.closed_form((K, 256), f"sim{K}:M", 1.0, 0.618))
        s = mod.cluster_similarity()
        s.backward()
        g[f"sim{K}/loss"] = s.item()
        put(g, f"sim{K}/dM", mod.M.grad, True)

    for (bpb, bs) in ((4, 2), (8, 8), (9, 12)):
        Z = (0.8 * O.closed_form((bpb * bs, 256), f"aug{bpb}_{bs}:Z", 1.0, 0.4142) + 0.1)
        Z = Z.clone().requires_grad_(True)
        a = ref_aug(Z, bpb, bs)
        a.sum().backward()
        g[f"aug{bpb}_{bs}/loss"] = a.item()
        put(g, f"aug{bpb}_{bs}/dZ", Z.grad, True)
    np.savez_compressed(os.path.join(HERE, "latent_losses.npz"), **g)

    # ---------------- (6) FFT feature step --------------------------------
    g = {}
    r, _ = O.closed_form_inputs(2, 4)
    r = r * 3.0  # make the clamp bite on some bins
    Fc = torch.fft.fftn(r, dim=(2, 3), norm="ortho")
    re, im = lt.torch_fftshift(Fc.real, Fc.imag)  # src/lofar_tools.py:24-30
    yy = torch.cat((re, im), dim=1)
    yy.clamp_(-10, 10)
    put(g, "fft/out", yy)
    g["fft/crop"] = yy[:, :, 60:68, 60:68].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "fft.npz"), **g)

    # ---------------- (8) minibatch loader: the reference's get_data_minibatch on a synthetic SAP ----
    # h5py is absent in the image; the loader only indexes nested groups/datasets, so a stand-in
    # module whose File() returns dicts of numpy arrays lets the UNMODIFIED reference function run.
    from tests.h5_fixture import make_sap
    sap, info = make_sap()
    fake = types.ModuleType("h5py")
    fake.File = lambda filename, mode="r": {"measurement": {"saps": {"0": sap}, "info": info}}
    lt.h5py = fake
    g8 = {}
    for nchan, suffix in ((4, ""), (8, "8")):  # 8: the reference's default, all four polarisations (:101-111)
        for normalize in (False, True):
            np.random.seed(123)
            px, py, yy8, uv8 = lt.get_data_minibatch(["synthetic.h5"], ["0"], batch_size=3, patch_size=128,
                                                      normalize_data=normalize, num_channels=nchan, uvdist=True)
            tag = ("norm" if normalize else "raw") + suffix
            g8[f"{tag}/patchxy"] = np.array([px, py])
            put(g8, f"{tag}/y", yy8)
            g8[f"{tag}/y_first"] = yy8[0, :, 60:64, 60:64].numpy().copy()
            g8[f"{tag}/uv"] = uv8.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "minibatch.npz"), **g8)

    # ---------------- (4) full closure + 3 Adam steps ---------------------
    g = {}
    cfg = O.StepConfig(K=4, bpb=4, batch_size=2)
    params, M0 = O.make_params(cfg)
    net = lm.AutoEncoderCNN2(latent_dim=cfg.L, channels=4, harmonic_scales=scales, rica=True)
    netT = lm.AutoEncoder1DCNN(latent_dim=cfg.Lt, channels=4, harmonic_scales=scales, rica=True)
    netF = lm.AutoEncoder1DCNN(latent_dim=cfg.Lt, channels=4, harmonic_scales=scales, rica=True)
    mod = lm.Kmeans(latent_dim=cfg.L + 2 * cfg.Lt, K=cfg.K, p=cfg.p)
    net.load_state_dict(params["net"])
    netT.load_state_dict(params["netT"])
    netF.load_state_dict(params["netF"])
    with torch.no_grad():
        mod.M.copy_(M0)
    x, uv = O.closed_form_inputs(8, 4)
    n = x.numel()
    plist = list(net.parameters()) + list(netT.parameters()) + list(netF.parameters()) \
        + list(mod.parameters())
    names = [f"net.{k}" for k, _ in net.named_parameters()] \
        + [f"netT.{k}" for k, _ in netT.named_parameters()] \
        + [f"netF.{k}" for k, _ in netF.named_parameters()] + ["mod.M"]
    opt = torch.optim.Adam(plist, lr=cfg.lr)
    ys = [torch.zeros(n) for _ in range(3)]
    sse = torch.nn.MSELoss(reduction="sum")

    def three(xin):
        a1, z = net(xin, uv)
        h = (xin - a1) / 2
        a2, zT = netT(torch.flatten(h, 2, 3), uv)
        a2 = a2.view_as(h)
        a3, zF = netF(torch.flatten(torch.transpose(h, 2, 3), 2, 3), uv)
        a3 = torch.transpose(a3.view_as(h), 2, 3)
        return a1, h, a2, a3, z, zT, zF

    log = []
    for it in range(4):
        opt.zero_grad()
        a1, h, a2, a3, z, zT, zF = three(x)
        t0 = sse(a1 + a2 + a3, x) / n
        t1 = (torch.dot(ys[0], (x - a1).view(-1)) + cfg.rho / 2 * sse(x, a1)) / n
        t2 = (torch.dot(ys[1], (h - a2).view(-1)) + cfg.rho / 2 * sse(h, a2)) / n
        t3 = (torch.dot(ys[2], (h - a3).reshape(-1)) + cfg.rho / 2 * sse(h, a3)) / n
        Z = torch.cat((z, zT, zF), 1)
        kd = cfg.alpha * mod.clustering_error(Z)
        cs = cfg.beta * mod.cluster_similarity()
        au = cfg.gamma * ref_aug(Z, cfg.bpb, cfg.batch_size)
        lc = lambda t: torch.sum(torch.log(torch.cosh(t))) / t.numel()
        rc = cfg.rica_lambda * (lc(z) + lc(zT) + lc(zF))
        total = t0 + t1 + t2 + t3 + kd + au + cs + rc
        total.backward()
        terms = [t0.item(), t1.item(), t2.item(), t3.item(), kd.item(), au.item(), cs.item(),
                 rc.item(), total.item()]
        log.append(terms)
        if it in (0, 3):
            for nm, p_ in zip(names, plist):
                g[f"it{it}/gnorm/{nm}"] = p_.grad.double().norm().item()
                g[f"it{it}/ghead/{nm}"] = p_.grad.reshape(-1)[:8].numpy().copy()
            g[f"it{it}/ynorm"] = np.array([t.double().norm().item() for t in ys])
        if it == 3:
            break
        opt.step()
        with torch.no_grad():
            a1, h, a2, a3, *_ = three(x)
            ys = [ys[0] + cfg.rho * (x - a1).view(-1), ys[1] + cfg.rho * (h - a2).view(-1),
                  ys[2] + cfg.rho * (h - a3).reshape(-1)]
    # ---------------- (5) LBFGSNew trajectories on a small deterministic regression problem ---
    import lbfgsnew as ref_lbfgs
    from tests.lbfgs_problem import make_problem
    gl = {}
    for mode, kw in (("batch", dict(history_size=7, max_iter=4, line_search_fn=True, batch_mode=True)),
                     ("full", dict(history_size=7, max_iter=6, line_search_fn=True, batch_mode=False)),
                     ("fixed", dict(history_size=5, max_iter=3, line_search_fn=False, batch_mode=False, lr=0.05))):
        params, closure_for = make_problem()
        opt = ref_lbfgs.LBFGSNew(params, **kw)
        losses, ts = [], []
        for it in range(4):
            closure = closure_for(opt, it if mode == "batch" else 0)
            l0 = opt.step(closure)
            losses.append(float(l0))
            ts.append(float(opt.state[opt._params[0]]["t"]))
        gl[f"{mode}/loss"] = np.array(losses)
        gl[f"{mode}/t"] = np.array(ts)
        gl[f"{mode}/final"] = torch.cat([p.detach().reshape(-1) for p in params]).numpy()
        gl[f"{mode}/func_evals"] = np.int64(opt.state[opt._params[0]]["func_evals"])
    np.savez_compressed(os.path.join(HERE, "lbfgs.npz"), **gl)

    g["terms"] = np.array(log)
    for nm, p_ in zip(names, plist):
        g[f"final/pnorm/{nm}"] = p_.detach().double().norm().item()
    np.savez_compressed(os.path.join(HERE, "closure.npz"), **g)
    print("golden fixtures written to", HERE)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
