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

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
# algorithmic bytes per patch per training step: layer-wise in+out, no recompute
STEP_BYTES_PER_PATCH = 15.04e6
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
    ap.add_argument("--roofline-cold", action="store_true",
                    help="also time the roofline kernel on rotating buffers (every byte from HBM); adds roofline_cold")
    ap.add_argument("--cpu-steps", type=int, default=6)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--save-tuning", default=None, metavar="PATH",
                    help="write the implicit-GEMM tile configurations measured in this run (lshm_amd/tuned_gfx950.txt)")
    ap.add_argument("--no-reuse-mode", action="store_true", help="skip the extra reuse_forward timing")
    ap.add_argument("--no-lbfgs", action="store_true", help="skip the extra LBFGS-iteration timing")
    ap.add_argument("--no-rica", action="store_true", help="skip the dictionary-learning (rica_lofar) timing")
    ap.add_argument("--bf16", action="store_true",
                    help="BASELINE configs[2]: bf16 operands on the matrix cores for the GEMM-shaped layers (fp32 accumulate, fp32 storage)")
    ap.add_argument("--only-khm", action="store_true", help="time only the K-harmonic kernel (dev aid)")
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
    return {"kernel": "khm_kernel+khm_reduce (fused fwd+bwd)", "shape": f"N={N},D={D},K={K}", "bound": "hbm",
            "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "ms": round(ms, 4), "traffic": _pmc_traffic("khm256_kernel<0, 12, 4>") if N == 1 << 20 else None}


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


def _pmc_traffic(kernel_key):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
    collected in separate --pmc runs; gfx950 correction 2*FETCH_SIZE; see profiles/r01/hbm_traffic.json)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01", "hbm_traffic.json")) as f:
            return json.load(f)["kernels"][kernel_key]["traffic_bytes_per_launch"]
    except Exception:
        return None


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
            "traffic": _pmc_traffic("tconv1d_stream_kernel<8, 4, false>") if B == 256 and not cold else None}


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


def cpu_baseline(args):
    """Oracle port of the same step (torch CPU ops, the reference's per-sample / per-centroid
    loop order for KHM, similarity and augmentation), bounded sample, host cores."""
    from oracle import lshm_oracle as O
    B = args.batch
    # 16 threads is the measured optimum of this port on the GPU box's 2 x 64-core EPYC 9575F host
    # (8: 121, 16: 129, 32: 104, 64: 56, 128: 25 patches/s; profiles/cpu_threads_probe.py)
    torch.set_num_threads(min(args.cpu_threads, os.cpu_count() or 1))
    torch.manual_seed(0)
    cfg = O.StepConfig(K=args.K, bpb=args.bpb, batch_size=B // args.bpb)
    params, M = O.make_params(cfg)
    x = torch.randn(B, 4, 128, 128)
    uv = 1000.0 * torch.randn(B, 2)
    y = [torch.zeros(x.numel()) for _ in range(3)]
    adam = O.AdamState(O.flat_leaves(params, M), cfg.lr)
    times = []
    for it in range(args.cpu_steps + 1):
        t0 = time.perf_counter()
        _, y, _ = O.admm_iteration(params, M, x, uv, y, cfg, adam, khm_fn=O.khm_loss_loop,
                                   sim_fn=O.cluster_similarity_loop, aug_fn=O.augmented_loss_loop)
        times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": round(B / t, 2), "unit": "patches/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{args.cpu_steps} timed ADMM iterations (median) after 1 warm-up at B={B}, K={args.K}, "
                      f"bpb={args.bpb}, all four parameter groups under Adam",
            "s_per_step": round(t, 3)}


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
        print(json.dumps({"khm_roofline": khm_roofline(dev), "khm_B256": khm_roofline(dev, N=256)}))
        return
    from lshm_amd import KHarmonicTrainer, TrainConfig
    B = args.batch
    cfg = TrainConfig(Kc=args.K, matrix_precision="bf16" if args.bf16 else "fp32")
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
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        tr.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    terms = tr.read_terms()
    ms = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    # Not the headline: the same iterations with TrainConfig.reuse_forward (iteration k+1 starts from the
    # activations of iteration k's no-grad forward; bit-for-bit the same trajectory, tests/test_gpu_step.py::
    # test_reuse_forward_is_bitwise_the_same_trajectory).  Reported next to `value`, which always recomputes.
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
           "vs_baseline": None, "dtype": "bf16 matrix operands, f32 accumulate/storage" if args.bf16 else "f32", "data": "synthetic",
           "config": {"workload": f"1xMI355X-per-rank: (B={B},4,128,128) synthetic patches, 2D+1D AE + K={args.K} "
                                  f"k-harmonic, fp32 (BASELINE.json configs[1]); one ADMM iteration = closure "
                                  f"fwd+bwd + Adam (all 4 groups) + no-grad fwd + multiplier update",
                      "global_batch": world * B, "per_gpu_batch": B, "K": args.K, "bpb": args.bpb,
                      "parallelism": f"dp{world}", "launch": "hipgraph" if use_graph else "eager",
                      "feature_stage": "v2 path (row/column 1D AEs); FFT op benchmarked separately"},
           "loss_total": terms["total"],
           "step_roofline": {"bound": "hbm", "achieved": round(STEP_BYTES_PER_PATCH * B / (ms * 1e-3) / 1e9, 1),
                             "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(STEP_BYTES_PER_PATCH * B / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "note": "whole step, algorithmic 15.04 MB/patch (SURVEY 8d), per GPU"}}
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
    if rank == 0 and not args.no_roofline:
        out["roofline"] = dominant_kernel_roofline(tr, dev)
        if args.roofline_cold:
            out["roofline_cold"] = dominant_kernel_roofline(tr, dev, cold=True)
        out["khm_roofline"] = khm_roofline(dev)
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
