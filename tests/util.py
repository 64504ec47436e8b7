"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def probe(t, idx):
    v = t.detach().double().cpu().reshape(-1).numpy()
    w = np.sin(0.37 * np.arange(v.size) + 0.11)
    return {"n": v.size, "sum": v.sum(), "asum": np.abs(v).sum(), "wsum": (v * w).sum(),
            "sample": v[idx]}


def assert_probe(g, name, t, rtol=1e-5, atol=1e-6):
    """Compare tensor t against the compact summary stored under `name` in g."""
    idx = g[f"{name}/idx"]
    p = probe(t, idx)
    assert p["n"] == int(g[f"{name}/n"]), (name, p["n"], int(g[f"{name}/n"]))
    ref = g[f"{name}/sample"]
    # fp32 dot products of O(1) terms: the absolute error scales with the tensor's magnitude,
    # so near-zero elements are compared against rtol * max|ref| rather than their own size
    np.testing.assert_allclose(p["sample"], ref, rtol=rtol, atol=atol + rtol * float(np.abs(ref).max()),
                               err_msg=name + " sample")
    asum = float(g[f"{name}/asum"])
    tol = rtol * asum + atol
    assert abs(p["asum"] - asum) <= tol, (name, "asum", p["asum"], asum)
    # signed / weighted sums cancel heavily: bound by rtol * sum|v|
    assert abs(p["sum"] - float(g[f"{name}/sum"])) <= tol, (name, "sum")
    assert abs(p["wsum"] - float(g[f"{name}/wsum"])) <= tol, (name, "wsum")


def assert_close(a, b, rtol=1e-5, atol=1e-6, msg=""):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().double().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=msg)


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()
