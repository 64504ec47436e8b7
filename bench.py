#!/usr/bin/env python3
"""Headline benchmark: spectrogram-patches/sec of one LSHM training step (one ADMM iteration of
src/kharmonic_lofar.py:131-202: closure forward + backward over the 2D AE, the two 1D AEs and the
K-harmonic / similarity / augmentation / RICA terms, Adam update of all four parameter groups,
no-grad forward and multiplier update) on synthetic (B=256,4,128,128) patches per GPU, fp32.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant kernel,
timed live with HIP events on the launch stream) and, at N=1, `cpu_baseline` (the oracle port of the
same step on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

# the host driver of this pool only supports dmabuf IPC (RCCL, tensor sharing): must be in the environment before the
# HIP runtime initialises, i.e. before the first torch.cuda call of this process
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
# algorithmic bytes per patch per training step: layer-wise in+out, no recompute
STEP_BYTES_PER_PATCH = 15.04e6


def _elided_bytes_per_patch(cfg, use_graph):
    """Algorithmic bytes (SURVEY 8d accounting) of the layers the schedule does not run: with the shared reconstruction
    pass the closure forward leaves out ConvTranspose1d(8, 4, 4, stride=4) of netT and netF."""
    shared = (cfg.share_recon_pass or cfg.reuse_forward) and not use_graph
    return 2 * 4.0 * (8 * 4096 + 4 * 16384) if shared else 0.0
STEP_FLOP_PER_PATCH = 205.6e6


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="patches per GPU")
    ap.add_argument("--K", type=int, default=10)
    ap.add_argument("--bpb", type=int, default=8)
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a HIP graph (slower: the graph executor serialises the two-stream backward)")
    ap.add_argument("--no-graph", action="store_true", help="(default) launch eagerly")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=6)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--save-tuning", default=None, metavar="PATH",
                    help="write the implicit-GEMM tile configurations measured in this run (lshm_amd/tuned_gfx950.txt)")
    ap.add_argument("--no-reuse-mode", action="store_true", help="skip the extra reuse_forward timing")
    ap.add_argument("--no-lbfgs", action="store_true", help="skip the extra LBFGS-iteration timing")
    ap.add_argument("--no-rica", action="store_true", help="skip the dictionary-learning (rica_lofar) timing")
    ap.add_argument("--bf16", action="store_true",
                    help="BASELINE configs[2]: bf16 operands on the matrix cores for the GEMM-shaped layers and bf16 storage of "
                         "the image-sized activations / gradients of the outer layers and glue passes (fp32 accumulate)")
    ap.add_argument("--bf16-operands-only", action="store_true",
                    help="with --bf16: keep every tensor in HBM fp32 (round the GEMM operands only)")
    ap.add_argument("--only-khm", action="store_true", help="time only the K-harmonic kernel (dev aid)")
    ap.add_argument("--no-extra-modes", action="store_true",
                    help="skip the secondary objects (sequential_forwards_mode, bf16_mode, k64_mode, admm10_loop)")
    ap.add_argument("--sequential-forwards", action="store_true",
                    help="headline with TrainConfig.overlap_forwards=False (the round-2 schedule), for A/B")
    ap.add_argument("--schedule-off", default="", metavar="NAMES",
                    help="comma-separated schedule choices of the headline trainer to switch off (TrainConfig.schedule_off, e.g. "
                         "no_deep2d), for A/B")
    ap.add_argument("--tune", type=int, default=0, help="TrainConfig.tune of the headline trainer (experimental placement word)")
    return ap.parse_args()


def event_time_ms(fn, iters, warm=2):
    """Average duration of fn() (enqueues on torch's current stream, which is the stream the
    kernels are launched on) measured with HIP events."""
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def khm_roofline(dev, N=1 << 20, D=256, K=10, p=4.0):
    """Fused K-harmonic forward+backward at the streaming shape of SURVEY 8(d): algorithmic bytes
    = read X + M, write dX + dM = 8 (N D + K D)."""
    from lshm_amd import _lib as L
    lib = L.load()
    X = torch.rand(N, D, device=dev)
    M = torch.rand(K, D, device=dev)
    dX = torch.empty_like(X)
    dM = torch.empty_like(M)
    loss = torch.zeros(1, device=dev, dtype=torch.float64)
    nws = lib.lshm_khm_workspace_floats(N, D, K)
    ws = torch.empty(nws, device=dev)
    inv = 1.0 / (float(N) * K * D)

    def run():
        L.check(lib.lshm_khm_fwd_bwd(L.ptr(X), D, L.ptr(M), N, D, K, p, 1e-9, inv, 1.0, L.ptr(loss), L.ptr(dX),
                                     D, L.ptr(dM), 0, L.ptr(ws), nws, L.stream()))
    ms = event_time_ms(run, 10)
    nbytes = 8.0 * (N * D + K * D)
    ach = nbytes / (ms * 1e-3) / 1e9
    if K <= 16:
        return {"kernel": "khm256_kernel<0,...> + khm_reduce (fused fwd+bwd)", "shape": f"N={N},D={D},K={K}", "bound": "hbm",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "ms": round(ms, 4), "bytes_per_launch": nbytes,
                "traffic": _pmc_traffic("khm256_kernel<0", f"khm_N{N}_K{K}") if N == 1 << 20 else None}
    # 16 < K <= 64: three K x D x N products on the fp32 matrix cores (S = X M^T, W M, W^T X): 6 N K D flop for
    # 8 N D bytes -- arithmetic intensity 48 flop/B at K = 64 against a ridge of ~20: the matrix pipe is the bound
    flop = 6.0 * N * K * D
    tf = flop / (ms * 1e-3) / 1e12
    return {"kernel": "khm_mfma_kernel<0> + khm_reduce (fused fwd+bwd, v_mfma_f32_16x16x4_f32)", "shape": f"N={N},D={D},K={K}",
            "bound": "mfma", "achieved": round(tf, 1), "peak": 157.3, "unit": "TFLOP/s", "frac": round(tf / 157.3, 4),
            "ms": round(ms, 4), "flop_per_launch": flop, "bytes_per_launch": nbytes,
            "hbm": {"achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)},
            "traffic": _pmc_traffic("khm_mfma_kernel<0", f"khm_N{N}_K{K}") if N == 1 << 20 else None}


def khm_distance_roofline(dev, N=1 << 20, D=256, K=10, p=4.0):
    """The all-pairs latent<->centroid distance pass alone (evaluation mode, src/evaluate_clustering.py:111-115:
    dist[k] = mean_n ||X_n - M_k||^p): reads X and M once, algorithmic bytes 4 (N D + K D)."""
    from lshm_amd import _lib as L
    lib = L.load()
    X = torch.rand(N, D, device=dev)
    M = torch.rand(K, D, device=dev)
    dist = torch.empty(K, device=dev)
    nws = lib.lshm_khm_workspace_floats(N, D, K)
    ws = torch.empty(nws, device=dev)

    def run():
        L.check(lib.lshm_khm_mean_distances(L.ptr(X), D, L.ptr(M), N, D, K, p, L.ptr(dist), L.ptr(ws), nws, L.stream()))
    ms = event_time_ms(run, 10)
    nbytes = 4.0 * (N * D + K * D)
    ach = nbytes / (ms * 1e-3) / 1e9
    return {"kernel": "khm256_kernel<2, 12, 4> + khm_reduce (distances only)", "shape": f"N={N},D={D},K={K}",
            "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "ms": round(ms, 4)}


PMC_FILE = next((q for q in (os.path.join(ROOT, "profiles", r, "hbm_traffic.json") for r in ("r04", "r03", "r02"))
                 if os.path.exists(q)), os.path.join(ROOT, "profiles", "r03", "hbm_traffic.json"))


def _pmc_file_id():
    """Which committed PMC file `traffic` figures come from, and its git blob id (sha1 of "blob <size>\0" + content):
    traffic is NOT measured in this run, so a stale file must be visible in the line itself."""
    import hashlib
    try:
        with open(PMC_FILE, "rb") as f:
            data = f.read()
        return {"file": os.path.relpath(PMC_FILE, ROOT), "git_blob": hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()}
    except OSError:
        return None


def _pmc_traffic(kernel_key, grid=None):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in
    separate --pmc runs; gfx950 correction 2*FETCH_SIZE; profiles/pmc_traffic.py -> profiles/r02/hbm_traffic.json).
    Entries are keyed by kernel name AND grid size, so a kernel that runs at two shapes (the K-harmonic kernel:
    B=256 inside the step, N=2^20 in the roofline demonstration) is never averaged across them."""
    try:
        with open(PMC_FILE) as f:
            ks = json.load(f)["kernels"]
        for k, v in ks.items():  # kernel_key may be a prefix (template arguments left out)
            if k.startswith(kernel_key) and (grid is None or k.endswith(f"@{grid}")):
                return v["traffic_bytes_per_launch"]
        return None
    except Exception:
        return None


def _pmc_step_traffic(key="step"):
    """HBM bytes of one whole ADMM iteration (sum over its launches) from the same file, or None."""
    try:
        with open(PMC_FILE) as f:
            return json.load(f)[key]["traffic_bytes"]
    except Exception:
        return None


def stream_kernel_roofline(tr, dev):
    """The convolution launch that moves the most bytes (1-D tconv5 forward, netT + netF in one launch).  Headline
    figure: HBM-cold (rotating buffer sets, every byte from HBM); `warm` = one buffer set re-used, as inside the
    step where its 67 MB input was just written and still sits in the 256 MiB Infinity Cache."""
    cold = dominant_kernel_roofline(tr, dev, cold=True)
    warm = dominant_kernel_roofline(tr, dev, cold=False)
    cold["warm"] = {"achieved": warm["achieved"], "frac": warm["frac"], "ms": warm["ms"],
                    "note": "input resident in the Infinity Cache, as in the step; not an HBM figure"}
    cold["traffic"] = warm["traffic"]
    return cold


def dominant_kernel_roofline(tr, dev, cold=False):
    """The largest convolution launch of the 1-D autoencoders: the k4 s4 transposed conv1d of the
    outermost decoder layer, netT and netF sharing the launch -- tconv5 of AutoEncoder1DCNN ((B,8,4096) ->
    (B,4,16384), src/lofar_models.py:142,183), run in the closure forward and in the no-grad forward.  Its
    kernel name (tconv1d_stream_kernel<8,4,false>) maps to exactly this shape, so the rocprofv3 average in
    profiles/ is directly comparable.  Algorithmic bytes per launch = read inputs + write outputs of both
    problems (weights ignored) = 2*4*(B*8*4096 + B*4*16384)."""
    from lshm_amd import _lib as L
    lib = L.load()
    B = tr.B
    # In the step this kernel's input was written by the previous kernel and is still in the 256 MiB
    # Infinity Cache; re-launching on one buffer set reproduces that (and matches the rocprofv3 in-step
    # average).  --roofline-cold rotates three sets (0.6 GB) so that every byte comes from HBM.
    nsets = 3 if cold else 1
    xs = [torch.randn(B, 8, 4096, device=dev) for _ in range(2 * nsets)]
    ws_ = [torch.randn(8, 4, 4, device=dev) * 0.1 for _ in range(2)]
    bs = [torch.zeros(4, device=dev) for _ in range(2)]
    ys = [torch.empty(B, 4, 16384, device=dev) for _ in range(2 * nsets)]
    turn = [0]

    def run():
        i = 2 * (turn[0] % nsets)
        turn[0] += 1
        L.check(lib.lshm_conv_fwd_pair(3, L.ptr(xs[i]), L.ptr(ws_[0]), L.ptr(bs[0]), L.ptr(ys[i]), L.ptr(xs[i + 1]),
                                       L.ptr(ws_[1]), L.ptr(bs[1]), L.ptr(ys[i + 1]), B, 8, 4, 1, 4096, 0, 0, 0, None,
                                       0, L.stream()))
    ms = event_time_ms(run, 48, warm=6)
    nbytes = 2 * 4.0 * (xs[0].numel() + ys[0].numel())
    ach = nbytes / (ms * 1e-3) / 1e9
    return {"kernel": "lshm::tconv1d_stream_kernel<8, 4, false> (1-D tconv5 forward, netT+netF in one launch)",
            "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "ms": round(ms, 4), "bytes_per_launch": nbytes,
            "buffers": "rotating, HBM-cold" if cold else "re-used, cache-warm as in the step",
            "traffic": _pmc_traffic("tconv1d_stream_kernel<8, 4, false>", "step") if (not cold and B == 256) else None}


CH = (4, 8, 12, 24, 48, 96, 192)


def gemm_family_roofline(tr, dev):
    """The implicit-GEMM family (igemm_kernel + its split-K epilogue): conv2-5 / tconv0-3 of the three
    autoencoders, forward (twice per iteration), data gradient and weight gradient, plus the dense layers --
    the largest share of the step's kernel time (profiles/r02/step_timeline.txt) and 57 % of its FLOPs.
    MFMA-bound on paper (exact-fp32 v_mfma_f32_16x16x4_f32, 157.3 TFLOP/s dense): each launch is timed alone
    on the stream through the C ABI with HIP events; achieved = algorithmic FLOPs of one iteration's launches /
    the sum of their times.  (Inside the step the two backward streams overlap, so the step spends less wall
    time than this sum.)"""
    from lshm_amd import _lib as L
    lib = L.load()
    B = tr.B
    bf16 = tr.cfg.matrix_precision == "bf16"
    P = L.ptr
    tot_ms, tot_flop, nl, rows = 0.0, 0.0, 0, []

    def conv_layer(kind, Cin, Cout, Hin, Win, pair):
        nonlocal tot_ms, tot_flop, nl
        nd2 = kind < 2
        ishape = (B, Cin, Hin, Win) if nd2 else (B, Cin, Win)
        oshape = {0: (B, Cout, Hin // 2, Win // 2), 1: (B, Cout, Hin * 2, Win * 2), 2: (B, Cout, (Win - 2) // 4 + 1),
                  3: (B, Cout, Win * 4)}[kind]
        wshape = ((Cout, Cin) if kind in (0, 2) else (Cin, Cout)) + ((4, 4) if nd2 else (4,))
        x, x2 = torch.randn(ishape, device=dev), torch.randn(ishape, device=dev)
        w = torch.randn(wshape, device=dev) * 0.1
        b = torch.zeros(Cout, device=dev)
        y, y2 = torch.empty(oshape, device=dev), torch.empty(oshape, device=dev)
        dz = torch.randn(oshape, device=dev)
        dx, dw, db = torch.empty(ishape, device=dev), torch.empty_like(w), torch.empty_like(b)
        nws = lib.lshm_conv_workspace_floats(kind, B, Cin, Cout, Hin, Win)
        ws = torch.empty(2 * nws, device=dev)
        st = L.stream()
        if pair:
            f = event_time_ms(lambda: L.check(lib.lshm_conv_fwd_pair(kind, P(x), P(w), P(b), P(y), P(x2), P(w), P(b), P(y2), B,
                                                                     Cin, Cout, Hin, Win, 0, 0, 1, P(ws), 2 * nws, st)), 20, 3)
        else:
            f = event_time_ms(lambda: L.check(L.fn("lshm_conv_fwd", bf16)(kind, P(x), P(w), P(b), P(y), B, Cin, Cout, Hin, Win,
                                                                         0, 0, 1, P(ws), nws, st)), 20, 3)
        d = event_time_ms(lambda: L.check(L.fn("lshm_conv_dgrad", bf16)(kind, P(dz), P(w), P(dx), P(x), B, Cin, Cout, Hin, Win,
                                                                       0, 0, P(ws), nws, st)), 20, 3)
        g = event_time_ms(lambda: L.check(L.fn("lshm_conv_wgrad", bf16)(kind, P(x), P(dz), P(dw), P(db), B, Cin, Cout, Hin, Win,
                                                                       0, 0, P(ws), nws, 0, st)), 20, 3)
        taps = 16 if nd2 else 4
        n_out = y.numel() if kind in (0, 2) else x.numel()
        c_other = Cin if kind in (0, 2) else Cout
        flop = 2.0 * n_out * c_other * taps * (2 if pair else 1)
        mult = 2 if pair else 1   # data / weight gradient of a pair: two launches of the single form here
        ms = 2 * f + mult * (d + g)
        tot_ms += ms
        tot_flop += 4 * flop
        nl += 2 + 2 * mult
        rows.append({"layer": f"kind{kind} {Cin}->{Cout} @{Hin}x{Win}" + (" x2" if pair else ""), "fwd_us": round(f * 1e3, 1),
                     "dgrad_us": round(d * 1e3, 1), "wgrad_us": round(g * 1e3, 1),
                     "tflops": round(4 * flop / (ms * 1e-3) / 1e12, 1)})

    def dense_layer(K, N, mult):
        nonlocal tot_ms, tot_flop, nl
        x, w, b = torch.randn(B, K, device=dev), torch.randn(N, K, device=dev) * 0.05, torch.zeros(N, device=dev)
        y, dz = torch.empty(B, N, device=dev), torch.randn(B, N, device=dev)
        dx, dw, db = torch.empty_like(x), torch.empty_like(w), torch.empty_like(b)
        nws = lib.lshm_linear_workspace_floats(B, K, N)
        ws = torch.empty(nws, device=dev)
        st = L.stream()
        f = event_time_ms(lambda: L.check(L.fn("lshm_linear_fwd", bf16)(P(x), K, P(w), P(b), P(y), N, B, K, N, 1, P(ws), nws, st)), 20, 3)
        d = event_time_ms(lambda: L.check(L.fn("lshm_linear_dgrad", bf16)(P(dz), N, P(w), P(dx), K, None, 0, B, K, N, P(ws), nws, st)), 20, 3)
        g = event_time_ms(lambda: L.check(L.fn("lshm_linear_wgrad", bf16)(P(x), K, P(dz), N, P(dw), P(db), B, K, N, P(ws), nws, st)), 20, 3)
        ms = mult * (2 * f + d + g)
        tot_ms += ms
        tot_flop += mult * 4 * 2.0 * B * K * N
        nl += 4 * mult

    for i in range(2, 6):
        conv_layer(0, CH[i], CH[i + 1], 128 >> i, 128 >> i, False)
    for i in range(0, 4):
        conv_layer(1, CH[6 - i], CH[5 - i], 2 << i, 2 << i, False)
    for i in range(2, 6):
        conv_layer(2, CH[i], CH[i + 1], 1, 16384 >> (2 * i), True)
    for i in range(0, 4):
        conv_layer(3, CH[6 - i], CH[5 - i], 1, 4 << (2 * i), True)
    c = tr.cfg
    for (Ldim, mult) in ((c.L, 1), (c.Lt, 2)):
        dense_layer(768 + 16, Ldim, mult)
        dense_layer(Ldim, Ldim, 2 * mult)
        dense_layer(Ldim + 16, 768, mult)
    peak = 2500.0 if bf16 else MFMA_F32_PEAK_TFLOPS
    ach = tot_flop / (tot_ms * 1e-3) / 1e12
    return {"kernel": "lshm::igemm_kernel family (+ splitk_epilogue_kernel): conv2-5 / tconv0-3 of net, netT, netF and the "
                      "dense layers, forward x2 + data gradient + weight gradient of one ADMM iteration, each launch alone",
            "bound": "mfma", "operands": "bf16" if bf16 else "f32", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(ach / peak, 4), "ms": round(tot_ms, 4), "launches": nl, "flop": tot_flop,
            "slowest": sorted(rows, key=lambda r: r["tflops"])[:3]}


def fft_roofline(tr, dev):
    """Batched 2-D FFT feature step (Demo.ipynb:169-175): fftn(ortho) + fftshift + cat(Re,Im) + clamp on
    (B,4,128,128).  Algorithmic bytes per image = read 64 KiB + write 128 KiB = 196,608 B."""
    from lshm_amd import _lib as L
    lib = L.load()
    B = tr.B
    x = torch.randn(B, 4, 128, 128, device=dev)
    out = torch.empty(B, 8, 128, 128, device=dev)

    def run():
        L.check(lib.lshm_fft2_ortho_shift_cat_clamp(L.ptr(x), L.ptr(out), B, 4, 10.0, L.stream()))
    ms = event_time_ms(run, 30, warm=3)
    nbytes = 196608.0 * B * 4
    return {"kernel": "lshm::fft2_feature_kernel", "shape": f"({B},4,128,128)", "bound": "hbm",
            "bytes_per_launch": nbytes, "ms": round(ms, 4), "achieved": round(nbytes / ms / 1e6, 1), "unit": "GB/s",
            "frac": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4)}


MFMA_F32_PEAK_TFLOPS = 157.0  # dense fp32 matrix-core peak, MI355X_MICROARCH.md


def rica_dictionary_roofline(dev, nbatch=1152, L=4 * 128 * 128, M=256, bf16=False):
    """SURVEY 8 f4, not the headline: dictionary learning of src/rica_lofar.py at its own sizes (L = 4*128*128,
    M = 256 :36-40; default_batch = 128 baselines x 9 patches = 1152 columns).  The closure is two fp32 GEMMs
    of 2 nbatch L M flop (A S and the code gradient), the dictionary update two more -- the one
    matrix-core-bound workload of the repository, priced against the dense fp32 MFMA peak."""
    from lshm_amd.rica_lofar import RicaDictionary
    g = torch.Generator().manual_seed(11)
    rd = RicaDictionary(L, M, device=dev, A=torch.rand(L, M, generator=g), matrix_precision="bf16" if bf16 else "fp32")
    x = torch.randn(nbatch, L, generator=g).to(dev)  # resident, like the headline's inputs
    rd.set_minibatch(x)
    St = torch.rand(nbatch, M, generator=g).to(dev).requires_grad_(True)
    ms_c = event_time_ms(lambda: rd.loss(St, True), 10, warm=2)
    with torch.no_grad():
        ms_l = event_time_ms(lambda: rd.loss(St, False), 10, warm=2)
    ms_u = event_time_ms(lambda: rd.update_dictionary(St.detach().t()), 5, warm=1)
    rd.A.copy_(torch.rand(L, M, generator=g))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nit = 3
    for _ in range(nit):
        rd.iteration(x)
    torch.cuda.synchronize()
    ms_it = (time.perf_counter() - t0) / nit * 1e3
    gemm = 2.0 * nbatch * L * M
    bf16 = rd.bf16
    peak = 2500.0 if bf16 else MFMA_F32_PEAK_TFLOPS  # dense matrix peaks, MI355X_MICROARCH.md
    return {"workload": f"X ~ A S, L={L}, M={M}, nbatch={nbatch} (src/rica_lofar.py:36-40,59-95)", "bound": "mfma",
            "operands": "bf16" if bf16 else "f32",
            "closure_ms": round(ms_c, 3), "closure_no_grad_ms": round(ms_l, 3), "dictionary_update_ms": round(ms_u, 3),
            "achieved": round(2 * gemm / (ms_c * 1e-3) / 1e12, 1), "peak": peak, "unit": "TFLOP/s",
            "frac": round(2 * gemm / (ms_c * 1e-3) / 1e12 / peak, 4),
            "iteration_ms": round(ms_it, 2), "patches_per_s": round(nbatch / (ms_it * 1e-3), 1),
            "note": "iteration = fresh codes + LBFGSNew(history 7, max_iter 10, line search, batch mode).step + dictionary update"}


def other_kernel_rooflines(tr, dev):
    """Two more launches timed live, each alone on the stream as it runs in the forward part of the step:
    the largest single kernel of the step (the fused reconstruction-loss pass, src/kharmonic_lofar.py:
    137-158,175: reads x, x1, x2, x3, y1..y3 and writes the three output gradients = 10 image-sized arrays)
    and the largest convolution of the 2-D autoencoder (conv0, (B,4,128,128) -> (B,8,64,64))."""
    from lshm_amd import _lib as L
    lib = L.load()
    B = tr.B
    out = []
    img = [torch.randn(B, 4, 128, 128, device=dev) for _ in range(10)]
    sums = torch.zeros(8, device=dev, dtype=torch.float64)
    nws = lib.lshm_recon_workspace_floats(B * 4, 128)
    ws = torch.empty(nws, device=dev)

    def recon():
        L.check(lib.lshm_recon_losses_fwd_bwd(*[L.ptr(t) for t in img[:7]], 1.0, B * 4, 128, L.ptr(sums), L.ptr(img[7]),
                                              L.ptr(img[8]), L.ptr(img[9]), L.ptr(ws), L.stream()))
    ms = event_time_ms(recon, 20, warm=3)
    nbytes = 10 * 4.0 * img[0].numel()
    out.append({"kernel": "lshm::recon_kernel (+ sum7_kernel)", "bound": "hbm", "bytes_per_launch": nbytes,
                "ms": round(ms, 4), "achieved": round(nbytes / ms / 1e6, 1), "unit": "GB/s",
                "frac": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4)})
    x = img[0]
    w = torch.randn(8, 4, 4, 4, device=dev) * 0.1
    b = torch.zeros(8, device=dev)
    y = torch.empty(B, 8, 64, 64, device=dev)

    def conv0():
        L.check(lib.lshm_conv_fwd(0, L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), B, 4, 8, 128, 128, 0, 0, 1, None, 0, L.stream()))
    ms = event_time_ms(conv0, 50, warm=5)
    nbytes = 4.0 * (x.numel() + y.numel())
    out.append({"kernel": "lshm::conv2d_q4_kernel<8, 4> (2-D conv0 forward)", "bound": "hbm",
                "bytes_per_launch": nbytes, "ms": round(ms, 4), "achieved": round(nbytes / ms / 1e6, 1),
                "unit": "GB/s", "frac": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4)})
    return out


def _lscpu():
    """Host CPU description (model, sockets, cores per socket, threads per core, logical CPUs) read from
    /proc/cpuinfo in-process: a process that has initialised the GPU must not fork + exec (`lscpu`)."""
    try:
        with open("/proc/cpuinfo") as f:
            txt = f.read()
    except OSError:
        return f"CPU(s): {os.cpu_count()}"
    model, phys, cores_per, siblings, ncpu = None, set(), None, None, 0
    for line in txt.splitlines():
        k, _, v = line.partition(":")
        k, v = k.strip(), v.strip()
        if k == "processor":
            ncpu += 1
        elif k == "model name" and model is None:
            model = v
        elif k == "physical id":
            phys.add(v)
        elif k == "cpu cores" and cores_per is None:
            cores_per = int(v)
        elif k == "siblings" and siblings is None:
            siblings = int(v)
    out = [f"Model name: {model}", f"Socket(s): {max(len(phys), 1)}"]
    if cores_per:
        out.append(f"Core(s) per socket: {cores_per}")
        if siblings:
            out.append(f"Thread(s) per core: {max(siblings // cores_per, 1)}")
    out.append(f"CPU(s): {ncpu or os.cpu_count()}")
    return "; ".join(out)


# measured in the build container (8 vCPU Xeon @ 2.1 GHz, 8 threads, B=256, K=10; profiles/reference_cpu_timing.py):
# the ACTUAL reference modules 2.83 s per iteration, this port 2.49 s -> the port is a slightly generous stand-in
PORT_OVER_REFERENCE_TIME = 0.88


def cpu_baseline(args):
    """Oracle port of the same step (torch CPU ops, the reference's per-sample / per-centroid
    loop order for KHM, similarity and augmentation), bounded sample, host cores: once with 8 threads (to set
    beside the survey container's figure for the real reference) and once with --cpu-threads (default 16, the
    measured optimum of this port on the GPU box's host); the larger of the two is `value`."""
    by = {}
    for n in sorted({8, args.cpu_threads}):
        if n <= (os.cpu_count() or 1):
            by[n] = _cpu_baseline_once(args, n, max(2, args.cpu_steps // (2 if n == 8 else 1)))
    best = max(by, key=lambda n: by[n]["value"])
    out = dict(by[best])
    out["by_threads"] = {str(n): {"value": v["value"], "s_per_step": v["s_per_step"]} for n, v in by.items()}
    out["host"] = _lscpu()
    out["port_over_reference_time"] = PORT_OVER_REFERENCE_TIME
    out["reference_equivalent"] = round(out["value"] * PORT_OVER_REFERENCE_TIME, 2)
    out["note"] = ("kind 'port': the reference's Python files do not travel to the GPU box; port_over_reference_time is the "
                   "port's time per iteration over the real reference's on the same 8 cores of the build container "
                   "(profiles/reference_cpu_timing.py), reference_equivalent = value x that ratio; BASELINE.md's frozen "
                   "figure for the real reference is 131 patches/s on 8 Xeon cores")
    return out


def _cpu_baseline_once(args, threads, nsteps):
    from oracle import lshm_oracle as O
    B = args.batch
    # 16 threads is the measured optimum of this port on the GPU box's 2 x 64-core EPYC 9575F host
    # (8: 121, 16: 129, 32: 104, 64: 56, 128: 25 patches/s; profiles/cpu_threads_probe.py)
    torch.set_num_threads(min(threads, os.cpu_count() or 1))
    torch.manual_seed(0)
    cfg = O.StepConfig(K=args.K, bpb=args.bpb, batch_size=B // args.bpb)
    params, M = O.make_params(cfg)
    x = torch.randn(B, 4, 128, 128)
    uv = 1000.0 * torch.randn(B, 2)
    y = [torch.zeros(x.numel()) for _ in range(3)]
    adam = O.AdamState(O.flat_leaves(params, M), cfg.lr)
    times = []
    for it in range(nsteps + 1):
        t0 = time.perf_counter()
        _, y, _ = O.admm_iteration(params, M, x, uv, y, cfg, adam, khm_fn=O.khm_loss_loop,
                                   sim_fn=O.cluster_similarity_loop, aug_fn=O.augmented_loss_loop)
        times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": round(B / t, 2), "unit": "patches/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{nsteps} timed ADMM iterations (median) after 1 warm-up at B={B}, K={args.K}, "
                      f"bpb={args.bpb}, all four parameter groups under Adam",
            "s_per_step": round(t, 3)}


def _timed_steps(tr, steps, warmup, barrier, world, dev, log=False):
    """`steps` ADMM iterations after `warmup` untimed ones, barrier + synchronize on both sides, MAX over ranks."""
    for _ in range(warmup):
        tr.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step()
        if log:
            tr.read_terms()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    return dt


def _variant(args, dev, pg, rank, world, barrier, x, uv, **cfg_kw):
    """A second trainer with another configuration on the same synthetic minibatch: value / ms_per_step / step frac."""
    from lshm_amd import KHarmonicTrainer, TrainConfig
    B = args.batch
    K = cfg_kw.pop("Kc", args.K)
    tr = KHarmonicTrainer(TrainConfig(Kc=K, **cfg_kw), batch=B, batch_per_bline=args.bpb, default_batch=B // args.bpb,
                          device=dev, process_group=pg)
    tr.init_parameters(seed=0)
    tr.new_minibatch(x.to(dev), uv.to(dev))
    dt = _timed_steps(tr, args.steps, max(2, args.warmup), barrier, world, dev)
    ms = dt / args.steps * 1e3
    terms = tr.read_terms()
    del tr
    torch.cuda.empty_cache()
    ach = STEP_BYTES_PER_PATCH * B / (ms * 1e-3) / 1e9
    return {"value": round(world * B * args.steps / dt, 1), "unit": "patches/s", "ms_per_step": round(ms, 4),
            "step_roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(ach / HBM_PEAK_GBS, 4)},
            "loss_total": terms["total"], "nonfinite_terms": terms["nonfinite"]}


def dp_evidence(tr, dev, rank, world):
    """What the N > 1 line is worth, measured THROUGH the process group instead of read from the environment: how many
    ranks a SUM all-reduce of 1 sees, which device every rank computes on (gathered), which data-parallel path the
    trainer took, and the time of one all-reduce of the gradient arena (6.9 MB at K = 10)."""
    import torch.distributed as dist
    one = torch.ones(1, device=dev)
    dist.all_reduce(one)
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "device_index": dev.index, "name": props.name,
          "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")),
          "shared_with_other_ranks": os.environ.get("LSHM_SHARE_GPU0") == "1"}
    devices = [None] * world
    dist.all_gather_object(devices, me)
    buf = torch.zeros(tr.nparams, device=dev)
    for _ in range(3):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    dist.barrier()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        dist.all_reduce(buf)
    b.record()
    torch.cuda.synchronize()
    us = torch.tensor([a.elapsed_time(b) * 100.0], device=dev, dtype=torch.float64)  # per all-reduce, us
    dist.all_reduce(us, op=dist.ReduceOp.MAX)
    return {"ranks_seen": int(round(one.item())), "devices": devices,
            "path": "engine" if getattr(tr, "_comm", None) is not None else "torch",
            "backend": dist.get_backend(), "allreduce_bytes": 4 * tr.nparams, "allreduce_us": round(us.item(), 1)}


def config5_mode(args, dev, pg, rank, world, barrier):
    """BASELINE.json configs[4] at one GPU per rank: the loader's minibatch (int8 visibilities of a synthetic SAP -> 3 x 3
    overlapping 128-patches per baseline, normalised, lofar_tools.get_data_minibatch's device pipeline: lshm_patches_from_vis)
    -> K = 64 clusters -> LBFGSNew refinement (history 7, max_iter 4, line search, batch mode; src/kharmonic_lofar.py:93,118),
    ms per ADMM iteration.  28 baselines x 9 patches = 252 patches per minibatch (the largest whole-baseline batch <= 256)."""
    from lshm_amd import KHarmonicTrainer, TrainConfig
    from lshm_amd.lofar_tools import minibatch_from_sap
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from tests.h5_fixture import make_sap
    nb = 28
    sap, info = make_sap(nbase=10, ntime=256, nfreq=256)  # (ten baselines in the synthetic SAP: drawn with replacement, as upstream's np.random.randint does)
    t0 = time.perf_counter()
    px, py, xb, uvb = minibatch_from_sap(sap, info, batch_size=nb, patch_size=128, normalize_data=True, num_channels=4,
                                         uvdist=True, baselinelist=[(3 * i + 1) % 10 for i in range(nb)], device=dev)
    torch.cuda.synchronize()
    t_load = time.perf_counter() - t0
    bpb = px * py
    B = xb.shape[0]
    tr = KHarmonicTrainer(TrainConfig(Kc=64), batch=B, batch_per_bline=bpb, default_batch=nb, device=dev, process_group=pg)
    tr.init_parameters(seed=0)
    tr.new_minibatch(xb.to(dev), uvb.to(dev))
    opt = tr.make_lbfgs()
    for _ in range(3):
        tr.step_lbfgs(opt)
    nl = max(2, args.steps // 5)
    barrier()
    t1 = time.perf_counter()
    for _ in range(nl):
        tr.step_lbfgs(opt)
    barrier()
    dt = time.perf_counter() - t1
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    terms = tr.read_terms()
    del tr
    torch.cuda.empty_cache()
    return {"value": round(world * B * nl / dt, 1), "unit": "patches/s", "ms_per_step": round(dt / nl * 1e3, 3), "steps": nl,
            "patches_per_minibatch": B, "bpb": bpb, "K": 64, "optimizer": "LBFGSNew(history 7, max_iter 4, line search, batch mode)",
            "loader_ms": round(t_load * 1e3, 2), "loss_total": terms["total"], "nonfinite_terms": terms["nonfinite"],
            "note": "configs[4] composed on one GPU per rank: synthetic int8 SAP through the device patch pipeline (h5py itself is "
                    "absent from the image), K = 64, LBFGSNew; tests/test_gpu_step.py::test_config5_composition_loader_k64_lbfgs holds "
                    "the composition to the CPU oracle"}


def admm10_loop(args, dev, pg, rank, world, barrier, gen):
    """The loop the reference runs (src/kharmonic_lofar.py:116-131,176-181): per minibatch ten ADMM iterations with the
    eight terms read back on the host in every one; a NEW minibatch every ten iterations, handed over as host
    tensors (pinned, 64 MiB at B=256) and uploaded on a copy stream while the previous ten iterate; inside the ten,
    iteration k+1 starts from iteration k's no-grad forward (TrainConfig.reuse_forward: 9 of 10 closure forwards are
    the same computation as the no-grad forward before them, bit for bit)."""
    from lshm_amd import KHarmonicTrainer, TrainConfig
    B, nadmm = args.batch, 10
    tr = KHarmonicTrainer(TrainConfig(Kc=args.K, reuse_forward=True), batch=B, batch_per_bline=args.bpb,
                          default_batch=B // args.bpb, device=dev, process_group=pg)
    tr.init_parameters(seed=0)
    host = []
    for _ in range(3):  # three pinned host minibatches, cycled
        xb = torch.randn(B, 4, 128, 128, generator=gen)
        xb = ((xb - xb.mean()) / xb.std()).pin_memory()
        host.append((xb, (1000.0 * torch.randn(B, 2, generator=gen)).pin_memory()))
    nmb = max(2, (args.steps + nadmm - 1) // nadmm)

    def run(n):
        for mb in range(n):
            tr.swap_in_minibatch()
            tr.prefetch_minibatch(*host[(mb + 1) % 3])
            for _ in range(nadmm):
                tr.step()
                tr.read_terms()
    tr.prefetch_minibatch(*host[0])
    run(1)
    barrier()
    t0 = time.perf_counter()
    run(nmb)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    its = nmb * nadmm
    del tr
    torch.cuda.empty_cache()
    return {"value": round(world * B * its / dt, 1), "unit": "patches/s", "ms_per_iteration": round(dt / its * 1e3, 4),
            "minibatches": nmb, "iterations_per_minibatch": nadmm,
            "host_to_device_bytes_per_minibatch": B * 4 * 128 * 128 * 4 + B * 2 * 4,
            "note": "new pinned host minibatch every 10 iterations (H2D on a copy stream, overlapped), terms read back every "
                    "iteration, reuse_forward inside the 10; trajectory bitwise that of the recomputing trainer "
                    "(tests/test_gpu_step.py::test_admm_loop_with_staged_minibatches_is_the_recompute_trajectory)"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # one process per GPU, started by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the
        # environment): never report a 1-GPU measurement under an N-GPU label
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} "
                         f"--master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...`")
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if os.environ.get("LSHM_SHARE_GPU0") == "1":  # rehearsal of the N>1 path on a one-GPU box
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("LSHM_DIST_BACKEND", "nccl")  # nccl == RCCL over xGMI on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        pg = dist.group.WORLD

    if args.save_tuning:  # measuring mode: every GEMM shape of this run times its tile configurations once
        from lshm_amd import _lib as _L
        _L.load().lshm_set_tuning(1, -1)
    if args.only_khm:
        print(json.dumps({"khm_roofline": khm_roofline(dev, K=args.K)}))  # one shape per run: PMC passes key on it
        return
    from lshm_amd import KHarmonicTrainer, TrainConfig
    B = args.batch
    cfg = TrainConfig(Kc=args.K, matrix_precision="bf16" if args.bf16 else "fp32",
                      activation_storage="bf16" if (args.bf16 and not args.bf16_operands_only) else "fp32",
                      overlap_forwards=not args.sequential_forwards,
                      schedule_off=tuple(n for n in args.schedule_off.split(",") if n), tune=args.tune)
    tr = KHarmonicTrainer(cfg, batch=B, batch_per_bline=args.bpb, default_batch=B // args.bpb, device=dev,
                          process_group=pg)
    tr.init_parameters(seed=0)  # identical replicas on every rank
    gen = torch.Generator(device="cpu").manual_seed(1234 + rank)
    x = torch.randn(B, cfg.num_in_channels, 128, 128, generator=gen)
    x = (x - x.mean()) / x.std()  # mimics normalize_data=True (src/lofar_tools.py:190-193)
    uv = 1000.0 * torch.randn(B, 2, generator=gen)
    tr.new_minibatch(x.to(dev), uv.to(dev))

    # eager by default: the engine enqueues a whole forward+backward per C call (the host stays ~3x ahead
    # of the device), and the weight-gradient chain runs on its own stream beside the data-gradient chain
    # (--graph: HIP-graph replay at N=1; with RCCL collectives inside the step only with LSHM_DP_GRAPH=1)
    use_graph = args.graph and not args.no_graph and (world == 1 or os.environ.get("LSHM_DP_GRAPH") == "1")
    if use_graph:
        try:
            tr.capture_graph(warmup=1)
        except Exception as e:  # report, fall back to eager launches of the same kernels
            if rank == 0:
                print(f"[bench] graph capture unavailable ({type(e).__name__}: {e}); eager launches", file=sys.stderr)
            use_graph = False
            tr._graph = None

    def barrier():
        torch.cuda.synchronize()  # this rank's queued work first: the barrier then marks "every rank's GPU is idle"
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    dt = _timed_steps(tr, args.steps, args.warmup, barrier, world, dev)
    terms = tr.read_terms()
    ms = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    # The reference prints its eight terms in every closure (src/kharmonic_lofar.py:176-181): the same loop with the
    # step log read back every iteration (one device->host copy of ten doubles, which also stops the host from
    # running ahead of the device).  `value` above is the free-running figure.
    dtl = _timed_steps(tr, args.steps, 0, barrier, world, dev, log=True)

    # Not the headline: the same iterations with TrainConfig.reuse_forward (iteration k+1 starts from the
    # activations of iteration k's no-grad forward; bit-for-bit the same trajectory, tests/test_gpu_step.py::
    # test_reuse_forward_is_bitwise_the_same_trajectory).  Reported next to `value`, whose two forwards per iteration are
    # both run -- up to the common subexpressions config.schedule names (the last 1-D layer pair and the reconstruction
    # terms of the closure forward: roofline.bytes_executed); literal_upstream_order_mode shares nothing.
    reuse = None
    if not use_graph and not args.no_reuse_mode:
        tr.cfg.reuse_forward = True
        tr.invalidate_forward()
        for _ in range(max(2, args.warmup)):
            tr.step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            tr.step()
        barrier()
        dt2 = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([dt2], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt2 = tt.item()
        tr.cfg.reuse_forward = False
        reuse = {"value": round(world * B * args.steps / dt2, 1), "unit": "patches/s",
                 "ms_per_step": round(dt2 / args.steps * 1e3, 4),
                 "note": "TrainConfig.reuse_forward=True: one forward per ADMM iteration instead of two, identical results"}

    out = {"metric": "spectrogram-patches/sec per training step (AE+FFT+k-harmonic), 1/2/4/8 GPU",
           "value": round(value, 1), "unit": "patches/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": ("bf16" if not args.bf16_operands_only else "bf16 matrix operands, f32 storage") if args.bf16 else "f32", "data": "synthetic",
           "config": {"workload": f"1xMI355X-per-rank: (B={B},4,128,128) synthetic patches, 2D+1D AE + K={args.K} "
                                  f"k-harmonic, {'bf16 storage + bf16 MFMA conv path, f32 accumulate (BASELINE.json configs[2])' if args.bf16 else 'fp32 (BASELINE.json configs[1])'}; one ADMM iteration = closure "
                                  f"fwd+bwd + Adam (all 4 groups) + no-grad fwd + multiplier update",
                      "global_batch": world * B, "per_gpu_batch": B, "K": args.K, "bpb": args.bpb,
                      "parallelism": f"dp{world}", "launch": "hipgraph" if use_graph else "eager",
                      "schedule": ("no-grad forward of iteration k and closure forward of iteration k+1 side by side on two streams "
                                   "(TrainConfig.overlap_forwards); common subexpressions of the two are computed once: the closure "
                                   "forward does not run the last layer of netT / netF (nothing reads its copy of their reconstructions) "
                                   "and takes its seven reconstruction terms and their gradients from the pass that follows the "
                                   "no-grad forward (TrainConfig.share_recon_pass; the pass also runs the backward of netT / netF's last "
                                   "layer, so two of the three gradient images stay on chip) -- bit for bit the trajectory of the schedule that "
                                   "runs both (tests/test_gpu_step.py::test_overlapped_forwards_are_bitwise_the_same_trajectory, "
                                   "::test_shared_reconstruction_pass_is_bitwise_the_same_trajectory); literal_upstream_order_mode "
                                   "runs everything") if cfg.overlap_forwards and not use_graph else
                                  "forwards one after the other",
                      "feature_stage": "v2 path (row/column 1D AEs); FFT op benchmarked separately"},
           "loss_total": terms["total"], "nonfinite_terms": terms["nonfinite"],
           "value_with_log": {"value": round(world * B * args.steps / dtl, 1), "unit": "patches/s",
                              "ms_per_step": round(dtl / args.steps * 1e3, 4),
                              "note": "the eight logged terms read back on the host every iteration, as upstream prints them"},
           # the step is launch- and latency-bound with a flat profile (no kernel above 6 % of its time): the unit that is
           # priced against the roofline is the whole ADMM iteration (15.04 MB and 205.6 MFLOP per patch)
           "roofline": {"kernel": "one whole ADMM iteration (all launches of lshm_engine_forward_backward + Adam + "
                                  "lshm_engine_multiplier_update_next), per GPU",
                        "bound": "hbm", "achieved": round(STEP_BYTES_PER_PATCH * B / (ms * 1e-3) / 1e9, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(STEP_BYTES_PER_PATCH * B / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "ms": round(ms, 4), "bytes_per_launch": STEP_BYTES_PER_PATCH * B,
                        # the layers this schedule actually runs: the closure forward's last 1-D layer pair (read 8 x 4096,
                        # write 4 x 16384 floats per patch and network) is elided when the reconstruction pass is shared
                        "bytes_executed": (STEP_BYTES_PER_PATCH - _elided_bytes_per_patch(cfg, use_graph)) * B,
                        "frac_executed": round((STEP_BYTES_PER_PATCH - _elided_bytes_per_patch(cfg, use_graph)) * B / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "flop_frac_of_f32_matrix_peak": round(STEP_FLOP_PER_PATCH * B / (ms * 1e-3) / 1e12 / 157.3, 4),
                        "traffic": _pmc_step_traffic() if B == 256 and args.K == 10 and not args.bf16 else None,
                        "traffic_source": _pmc_file_id(),
                        "note": "algorithmic 15.04 MB/patch (SURVEY 8d: layer-wise read input + write output, glue passes "
                                "counted as zero); traffic = PMC HBM bytes summed over the iteration's launches"}}
    out["step_roofline"] = {k: out["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac")}
    if world > 1:
        out["dp"] = dp_evidence(tr, dev, rank, world)
    if reuse is not None:
        out["reuse_forward_mode"] = reuse
    if not args.no_lbfgs and not use_graph:
        # the LBFGS iteration (LBFGSNew(history 7, max_iter 4, line search, batch mode), the
        # commented-out optimiser of src/kharmonic_lofar.py:93) reported beside the Adam iteration
        opt = tr.make_lbfgs()
        tr.invalidate_forward()
        nl = max(2, args.steps // 5)
        for _ in range(3):  # the inter-batch branch of LBFGSNew only runs from its second step on (lazy kernel loads)
            tr.step_lbfgs(opt)
        barrier()
        t2 = time.perf_counter()
        for _ in range(nl):
            tr.step_lbfgs(opt)
        barrier()
        dt3 = time.perf_counter() - t2
        if world > 1:
            tt = torch.tensor([dt3], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt3 = tt.item()
        out["lbfgs_iteration"] = {"value": round(world * B * nl / dt3, 1), "unit": "patches/s", "ms_per_step": round(dt3 / nl * 1e3, 3),
                                  "steps": nl, "note": "one ADMM iteration with LBFGSNew.step(closure) instead of Adam"}
        # not upstream's evaluation pattern (secondary figure, like reuse_forward_mode): the line search starts from the
        # loss step() has just computed at the same point instead of re-evaluating the deterministic closure there
        opt = tr.make_lbfgs(reuse_known_loss=True)
        tr.invalidate_forward()
        for _ in range(3):
            tr.step_lbfgs(opt)
        barrier()
        t2 = time.perf_counter()
        for _ in range(nl):
            tr.step_lbfgs(opt)
        barrier()
        dt4 = time.perf_counter() - t2
        if world > 1:
            tt = torch.tensor([dt4], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt4 = tt.item()
        out["lbfgs_iteration"]["reuse_known_loss"] = {"value": round(world * B * nl / dt4, 1), "ms_per_step": round(dt4 / nl * 1e3, 3)}
    # the headline trainer's engine (its streams and 5 GB of workspace) goes away before anything else is measured: a
    # second engine beside it shares the process's hardware queues with it and runs 10-15 % slower
    import types
    tr = types.SimpleNamespace(B=tr.B, cfg=tr.cfg)
    torch.cuda.empty_cache()
    if not use_graph and not args.no_extra_modes and not args.bf16 and args.K == 10:
        # secondary objects, each on a trainer of its own over the same synthetic minibatch:
        #  sequential_forwards_mode  the round-2 schedule (overlap_forwards=False), for round-to-round comparison
        #  bf16_mode                 BASELINE.json configs[2]: bf16 matrix operands + bf16 activation storage
        #  k64_mode                  configs[4]'s K = 64 centroids, Adam iteration
        #  admm10_loop               the loop upstream runs: new host minibatch every 10 iterations, terms read back
        out["sequential_forwards_mode"] = _variant(args, dev, pg, rank, world, barrier, x, uv, overlap_forwards=False)
        # the literal order of src/kharmonic_lofar.py:131-202: closure forward (whole), backward, Adam, no-grad forward (whole),
        # multiplier update as a pass of its own -- nothing shared between the two forwards: what the sharing is worth
        out["literal_upstream_order_mode"] = _variant(args, dev, pg, rank, world, barrier, x, uv, overlap_forwards=False,
                                                       share_recon_pass=False)
        out["bf16_mode"] = _variant(args, dev, pg, rank, world, barrier, x, uv, matrix_precision="bf16",
                                    activation_storage="bf16")
        out["bf16_mode"]["dtype"] = "bf16 operands (v_mfma_f32_16x16x16_bf16) + bf16 storage of the image-sized tensors, f32 accumulate"
        out["bf16_mode"]["traffic"] = _pmc_step_traffic("step_bf16")
        out["k64_mode"] = _variant(args, dev, pg, rank, world, barrier, x, uv, Kc=64)
        out["admm10_loop"] = admm10_loop(args, dev, pg, rank, world, barrier, gen)
        out["config5_mode"] = config5_mode(args, dev, pg, rank, world, barrier)
    if rank == 0 and not args.no_roofline:
        out["gemm_family_roofline"] = gemm_family_roofline(tr, dev)
        out["stream_kernel_roofline"] = stream_kernel_roofline(tr, dev)
        out["khm_roofline"] = khm_roofline(dev)
        out["khm_k64_roofline"] = khm_roofline(dev, K=64)
        out["khm_distance_roofline"] = khm_distance_roofline(dev)
        out["other_kernels"] = other_kernel_rooflines(tr, dev)
        out["fft_roofline"] = fft_roofline(tr, dev)
        if not args.no_rica:
            out["rica_dictionary"] = rica_dictionary_roofline(dev, bf16=args.bf16)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and args.save_tuning:
        from lshm_amd import _lib as L
        print(f"[bench] {L.save_tuning(args.save_tuning)} tuned shapes -> {args.save_tuning}", file=sys.stderr)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
