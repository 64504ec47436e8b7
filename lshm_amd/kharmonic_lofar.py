"""Training-step driver: the ADMM loop body of the reference's
``src/kharmonic_lofar.py:128-202`` on the native step engine.

One :meth:`KHarmonicTrainer.step` is one iteration of ``for admm in range(Nadmm)``:
closure forward + backward over the three cascaded autoencoders and the
K-harmonic / similarity / augmentation / RICA terms, one Adam update of every
selected parameter group, then the no-grad forward and the multiplier update
``y_k += rho * r_k``.  All device work is enqueued by three C calls
(``lshm_engine_forward_backward``, ``lshm_adam_step_flat``,
``lshm_engine_multiplier_update``) on torch's current stream, so the whole
iteration can be captured in a HIP graph (``use_graph=True``).

Data parallelism: one process per GPU; patches (whole baselines) are sharded
over ranks, parameters and Adam state are replicated, and the flat gradient
arena plus the 9 loss terms are summed with one RCCL all-reduce each per
closure (``torch.distributed``, backend ``nccl`` == RCCL on ROCm).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Dict, Iterable, Optional, Sequence, Tuple

import torch

from . import _lib as L

GROUPS = ("net", "netT", "netF", "mod")
TERM_NAMES = ("loss0", "loss1", "loss2", "loss3", "kdist", "aug", "sim", "rica", "total")


def format_terms(terms, epoch: int, i: int, admm: int, use_rica: bool = True) -> str:
    """The reference's log line (src/kharmonic_lofar.py:176-181): ``'%d %d %d' + ' %f' * n`` with the columns
    epoch, i, admm, loss0, loss1, loss2, loss3, kdist, aug, sim and, with RICA, the RICA penalty (8 / 7 values)."""
    cols = [terms[k] for k in TERM_NAMES[:7]] + ([terms["rica"]] if use_rica else [])
    return ("%d %d %d " % (epoch, i, admm)) + " ".join("%f" % v for v in cols)


@dataclass
class TrainConfig:
    """Script constants of src/kharmonic_lofar.py:25-57,92 (defaults identical)."""
    L: int = 224
    Lt: int = 16
    Kc: int = 10
    Khp: float = 4
    alpha: float = 0.01
    beta: float = 0.01
    gamma: float = 0.01
    rho: float = 1.0
    use_rica: bool = True
    rica_lambda: float = 0.01
    patch_size: int = 128
    num_in_channels: int = 4
    harmonic_scales: Tuple[float, ...] = (1e-4, 1e-3, 1e-2, 1e-1)
    lr: float = 1e-4
    betas: Tuple[float, float] = (0.9, 0.999)
    adam_eps: float = 1e-8
    # which parameter groups the optimiser updates; upstream ships {"net"} (:86-90) and asks the
    # user to alternate by hand (README.md:27-30); the benchmark trains all four
    train_groups: Tuple[str, ...] = GROUPS
    # Inside the ADMM loop the no-grad forward that ends iteration k (after the optimiser step) and the
    # closure forward that opens iteration k+1 are the same computation (same parameters, same minibatch;
    # the multipliers enter after the forward).  True: iteration k+1 starts from the saved activations -
    # bit-for-bit the same trajectory, one forward per iteration instead of two.  Only valid while
    # parameters and inputs change through this trainer's own methods (see invalidate_forward).
    reuse_forward: bool = False
    # "bf16": the GEMM-shaped kernels (conv2-5 / tconv0-3, their weight gradients, the dense layers) round
    # their operands to bf16 and multiply on the bf16 matrix cores, fp32 accumulation; storage, the outer
    # layers, losses and Adam stay fp32 (BASELINE.json configs[2]).  A property of this trainer's engine
    # (lshm_step_config.precision): other trainers / modules in the process are unaffected.
    matrix_precision: str = "fp32"
    # "bf16" (with matrix_precision="bf16"): the image-sized activations and gradients of the bandwidth-bound part of
    # the step -- the three reconstructions, the row / column residuals, every image-sized gradient -- live in HBM as
    # bf16; accumulation, master weights, multipliers, losses and Adam stay fp32 (LSHM_PRECISION_BF16_STORAGE)
    activation_storage: str = "fp32"
    # The multiplier update that closes iteration k and the reconstruction terms that open iteration k+1 read
    # the same seven image-sized arrays; True (default): they share one pass (the closure forward itself is
    # still recomputed, as upstream does) -- identical results, one 0.67 GB pass less per iteration.
    share_recon_pass: bool = True
    # After the optimiser step the no-grad forward that closes iteration k (:187-196) and the closure forward
    # that opens iteration k+1 (:135-150) depend on the updated parameters alone.  True (default): they are
    # issued together, as two chains on two HIP streams with separate activation buffers
    # (lshm_engine_multiplier_update_next_ex, LSHM_NEXT_CONCURRENT_FORWARD); neither waits for the other.  Both forwards
    # run, up to their common subexpressions: the closure forward leaves out the last layer of netT / netF (nothing
    # reads its copy of their reconstructions) and, with share_recon_pass, takes its reconstruction terms and gradient
    # images from the pass behind the no-grad forward.  Same trajectory bit for bit.  False: one after the other.
    overlap_forwards: bool = True
    # Schedule choices of the engine to switch OFF (names of lshm_amd._lib.SCHEDULE_BITS, e.g. ("no_deep2d",)): each
    # restores the launch sequence the choice replaced -- for A/B measurements and the tests that hold a fused kernel's
    # trajectory to the launches it replaced.  Per trainer (lshm_step_config.schedule), not per process.
    schedule_off: Tuple[str, ...] = ()
    tune: int = 0  # lshm_step_config.tune (experimental placement word of A/B measurements; 0 = shipped)


class KHarmonicTrainer:
    def __init__(self, cfg: TrainConfig, batch: int, batch_per_bline: int, default_batch: Optional[int] = None,
                 device: Optional[torch.device] = None, process_group=None):
        self.cfg = cfg
        if cfg.matrix_precision not in ("fp32", "bf16"):
            raise ValueError("matrix_precision must be 'fp32' or 'bf16'")
        if cfg.activation_storage not in ("fp32", "bf16"):
            raise ValueError("activation_storage must be 'fp32' or 'bf16'")
        if cfg.activation_storage == "bf16" and cfg.matrix_precision != "bf16":
            raise ValueError("activation_storage='bf16' goes with matrix_precision='bf16' (BASELINE configs[2])")
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise RuntimeError("KHarmonicTrainer needs a HIP device; there is no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.lib = L.load()
        self.pg = process_group
        self.world = 1
        if process_group is not None:
            import torch.distributed as dist
            self.world = dist.get_world_size(process_group)
        self.B = int(batch)
        self.bpb = int(batch_per_bline)
        self.default_batch = int(default_batch if default_batch is not None else max(1, batch // batch_per_bline))
        sc = L.StepConfig()
        sc.B, sc.C, sc.P = self.B, cfg.num_in_channels, cfg.patch_size
        sc.L, sc.Lt, sc.K = cfg.L, cfg.Lt, cfg.Kc
        sc.p = float(cfg.Khp)
        sc.alpha, sc.beta, sc.gamma, sc.rho = cfg.alpha, cfg.beta, cfg.gamma, cfg.rho
        sc.rica_lambda, sc.rica = cfg.rica_lambda, int(cfg.use_rica)
        sc.bpb, sc.batch_size = self.bpb, self.default_batch
        sc.H = len(cfg.harmonic_scales)
        for i, s in enumerate(cfg.harmonic_scales):
            sc.scales[i] = s
        sc.world = self.world
        sc.precision = (L.PRECISION_BF16_STORAGE if cfg.activation_storage == "bf16" else
                        L.PRECISION_BF16_OPERANDS if cfg.matrix_precision == "bf16" else L.PRECISION_F32)
        sc.schedule = 0
        for name in cfg.schedule_off:
            sc.schedule |= L.SCHEDULE_BITS[name]
        sc.tune = int(cfg.tune)
        self._sc = sc
        h = C.c_void_p()
        # the engine's side stream and events are created on the device that is current now, and every
        # launch of this trainer goes to that device whatever the caller's current device is
        with L.on_device(self.device):
            L.check(self.lib.lshm_engine_create(C.byref(sc), C.byref(h)), "engine_create")
        self._h = h
        n = self.lib.lshm_engine_param_count(h)
        self.nparams = n
        dev = self.device
        self.params = torch.zeros(n, device=dev)
        self.grads = torch.zeros(n, device=dev)
        self.exp_avg = torch.zeros(n, device=dev)
        self.exp_avg_sq = torch.zeros(n, device=dev)
        # Adam's step number: a host integer for eager launches (passed by value, no extra kernel); a captured
        # graph needs it on the device (incremented by a node of the graph), see capture_graph
        self.adam_steps = 0
        self.step_count = torch.zeros(1, device=dev, dtype=torch.int32)
        self.ws_floats = self.lib.lshm_engine_workspace_floats(h)
        self.ws = torch.empty(self.ws_floats, device=dev)
        self.terms = torch.zeros(16, device=dev, dtype=torch.float64)
        # name -> (offset, shape)
        self.layout: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        buf = C.create_string_buffer(128)
        off, num, nd = C.c_long(), C.c_long(), C.c_int()
        shp = (C.c_long * 4)()
        i = 0
        while self.lib.lshm_engine_param_name(h, i, buf, 128, C.byref(off), C.byref(num), C.byref(nd), shp) == 0:
            self.layout[buf.value.decode()] = (off.value, tuple(shp[j] for j in range(nd.value)))
            i += 1
        # gradient mask for frozen groups (1 = trained)
        self.set_train_groups(cfg.train_groups)
        img = (self.B, cfg.num_in_channels, cfg.patch_size, cfg.patch_size)
        self.x = torch.zeros(img, device=dev)
        self.uv = torch.zeros((self.B, 2), device=dev)
        self.y = [torch.zeros(self.x.numel(), device=dev) for _ in range(3)]
        # Data parallelism.  Default: torch.distributed all-reduces after the closure (backend nccl == RCCL;
        # gloo in the CPU-rehearsal tests).  LSHM_DP_ENGINE=1: the collectives run inside the engine's closure on
        # its own streams (lshm_engine_set_comm: early bucket for netT / netF beside the 2-D backward) -- opt-in
        # until that path has run on a real multi-GPU node (it is rehearsed with two ranks on one GPU through a
        # host-shared-memory stand-in for RCCL, tests/test_gpu_dp.py).
        self._comm = None
        if self.world > 1 and os.environ.get("LSHM_DP_ENGINE") == "1" and os.environ.get("LSHM_DP_TORCH") != "1":
            self._attach_engine_comm(process_group)
        self._graph = None
        self._prefetched = False     # the saved forward is the next closure's own (overlap_forwards), not a re-used one
        self._saved_forward = False  # the workspace holds the forward of the current params / x / uv
        self._recon_ready = False    # ... and the reconstruction terms of the next closure (share_recon_pass)
        self._opt_generation = 0     # bumped by set_train_groups: optimisers built before it are stale

    def _attach_engine_comm(self, process_group):
        """Engine-side communicator, all ranks or none: a rank whose lshm_comm_init / lshm_engine_set_comm failed
        while the others succeeded would issue different collectives from them (torch.distributed after the closure
        against RCCL inside it) and the job would hang.  Every rank therefore reports success, the flags are
        MIN-reduced over the group, and one failure detaches the communicator everywhere.  The early bucket is
        agreed the same way (lshm_engine_comm_early_bucket / lshm_engine_set_early_bucket)."""
        import sys
        from .dist import Communicator, agree

        def fallback(why):
            print(f"lshm_amd: engine-side communicator unavailable ({why}); every rank uses torch.distributed "
                  "all-reduces after the closure", file=sys.stderr)

        # every step below ends in the same decision on every rank before the next collective is issued
        if not agree(bool(self.lib.lshm_comm_available()), process_group):
            return fallback("RCCL is not available on at least one rank")
        try:
            comm = Communicator(process_group, self.device)  # raises on every rank or on none (see dist.Communicator)
        except RuntimeError as e:
            return fallback(e)
        with L.on_device(self.device):
            rc = self.lib.lshm_engine_set_comm(self._h, comm.handle)
        if not agree(rc == 0, process_group):
            with L.on_device(self.device):
                self.lib.lshm_engine_set_comm(self._h, None)
            comm.close()
            return fallback("lshm_engine_set_comm failed on at least one rank")
        self._comm = comm
        early = agree(bool(self.lib.lshm_engine_comm_early_bucket(self._h)), process_group)
        L.check(self.lib.lshm_engine_set_early_bucket(self._h, int(early)), "engine_set_early_bucket")

    def set_schedule_off(self, names=()):
        """Replace the per-call schedule choices that are switched off (lshm_engine_set_schedule; the bits that shaped the
        engine at creation keep their value).  Returns the names in effect."""
        word = 0
        for n in names:
            word |= L.SCHEDULE_BITS[n]
        got = self.lib.lshm_engine_set_schedule(self._h, word)
        self.invalidate_forward()
        return tuple(n for n, b in L.SCHEDULE_BITS.items() if got & b)

    def _stream(self):
        return L.stream(self.device)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.lshm_engine_destroy(self._h)
                self._h = None
            if getattr(self, "_comm", None) is not None:
                self._comm.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ parameters
    def view(self, name: str, flat: Optional[torch.Tensor] = None) -> torch.Tensor:
        off, shape = self.layout[name]
        n = 1
        for s in shape:
            n *= s
        return (self.params if flat is None else flat)[off:off + n].view(shape)

    def set_train_groups(self, groups: Iterable[str]):
        """Choose the parameter groups the optimiser updates (upstream edits ``params`` at :86-90 and builds a
        new optimiser).  Same semantics here: a parameter outside the selection never changes, and the
        optimiser state starts afresh (Adam moments and step count zeroed; LBFGS optimisers made before this
        call refuse to step -- call make_lbfgs again)."""
        groups = tuple(groups)
        for g in groups:
            if g not in GROUPS:
                raise ValueError(f"unknown parameter group {g!r}")
        self.train_groups = groups
        # contiguous arena ranges of the selected groups (tensors of a group are adjacent: net, netT, netF, mod)
        ranges = []
        for name, (off, shape) in sorted(self.layout.items(), key=lambda kv: kv[1][0]):
            if name.split(".", 1)[0] not in groups:
                continue
            n = 1
            for d in shape:
                n *= d
            end = off + (n + 3) // 4 * 4
            if ranges and ranges[-1][1] == off:
                ranges[-1][1] = end
            else:
                ranges.append([off, end])
        self._ranges = [(a, min(b, self.nparams)) for a, b in ranges]
        if hasattr(self, "exp_avg"):
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            self.step_count.zero_()
            self.adam_steps = 0
        self._opt_generation = getattr(self, "_opt_generation", 0) + 1
        if set(groups) == set(GROUPS):
            self._mask = None
            return
        m = torch.zeros(self.nparams, device=self.device)
        for a, b in self._ranges:
            m[a:b] = 1.0
        self._mask = m

    def load_state_dicts(self, net=None, netT=None, netF=None, mod=None):
        """Accepts the reference's four state_dicts (keys 'conv0.weight' ... / 'M')."""
        self.invalidate_forward()
        with torch.no_grad():
            for prefix, sd in (("net", net), ("netT", netT), ("netF", netF), ("mod", mod)):
                if sd is None:
                    continue
                for k, v in sd.items():
                    name = f"{prefix}.{k}"
                    if name not in self.layout:
                        raise KeyError(f"unexpected key {k!r} for {prefix}")
                    dst = self.view(name)
                    if tuple(v.shape) != tuple(dst.shape):
                        raise RuntimeError(f"size mismatch for {name}: {tuple(v.shape)} vs {tuple(dst.shape)}")
                    dst.copy_(v.to(self.device, torch.float32))

    def state_dicts(self) -> Dict[str, Dict[str, torch.Tensor]]:
        out = {g: {} for g in GROUPS}
        for name in self.layout:
            g, k = name.split(".", 1)
            out[g][k] = self.view(name).detach().clone()
        return out

    def init_parameters(self, seed: Optional[int] = None):
        """Reference default initialisation (torch layer defaults, drawn in the reference's
        construction order net, netT, netF, mod: src/kharmonic_lofar.py:60-65)."""
        from .lofar_models import AutoEncoder1DCNN, AutoEncoderCNN2, Kmeans
        if seed is not None:
            torch.manual_seed(seed)
        c = self.cfg
        hs = torch.tensor(c.harmonic_scales)
        net = AutoEncoderCNN2(c.L, c.num_in_channels, hs, c.use_rica)
        netT = AutoEncoder1DCNN(c.Lt, c.num_in_channels, hs, c.use_rica)
        netF = AutoEncoder1DCNN(c.Lt, c.num_in_channels, hs, c.use_rica)
        mod = Kmeans(c.L + 2 * c.Lt, c.Kc, c.Khp)
        self.load_state_dicts(net.state_dict(), netT.state_dict(), netF.state_dict(), mod.state_dict())

    def save_checkpoints(self, prefix: str = "."):
        """Same four files / dict format as src/kharmonic_lofar.py:210-222."""
        import os
        sds = self.state_dicts()
        for fname, g in (("net.model", "net"), ("khm.model", "mod"), ("netT.model", "netT"), ("netF.model", "netF")):
            torch.save({"model_state_dict": {k: v.cpu() for k, v in sds[g].items()}}, os.path.join(prefix, fname))

    # ------------------------------------------------------------------ data
    def new_minibatch(self, x: torch.Tensor, uv: torch.Tensor):
        """Start a new minibatch: multipliers reset to zero (src/kharmonic_lofar.py:128-130)."""
        if tuple(x.shape) != tuple(self.x.shape) or tuple(uv.shape) != tuple(self.uv.shape):
            raise RuntimeError(f"expected x {tuple(self.x.shape)} and uv {tuple(self.uv.shape)}")
        self.x.copy_(x)
        self.uv.copy_(uv)
        for t in self.y:
            t.zero_()
        self.invalidate_forward()

    def prefetch_minibatch(self, x_host: torch.Tensor, uv_host: torch.Tensor):
        """Upload the NEXT minibatch (src/kharmonic_lofar.py:118 hands over host tensors) while the ADMM iterations
        of the current one run: an asynchronous copy from pinned host memory into a staging pair, on a copy
        stream of its own.  swap_in_minibatch() makes it current."""
        if tuple(x_host.shape) != tuple(self.x.shape) or tuple(uv_host.shape) != tuple(self.uv.shape):
            raise RuntimeError(f"expected x {tuple(self.x.shape)} and uv {tuple(self.uv.shape)}")
        with L.on_device(self.device):
            if getattr(self, "_copy_stream", None) is None:
                self._copy_stream = torch.cuda.Stream(self.device)
                self._stage = (torch.empty_like(self.x), torch.empty_like(self.uv))
                self._staged = None
            # the staging pair may still be read by iterations enqueued before the last swap
            self._copy_stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._copy_stream):
                self._stage[0].copy_(x_host, non_blocking=True)
                self._stage[1].copy_(uv_host, non_blocking=True)
                self._staged = torch.cuda.Event()
                self._staged.record(self._copy_stream)

    def swap_in_minibatch(self):
        """Start the minibatch uploaded by prefetch_minibatch: the compute stream waits for the copy (no host
        synchronisation), the staging pair and the current pair trade places, multipliers reset to zero (:128-130)."""
        if getattr(self, "_staged", None) is None:
            raise RuntimeError("swap_in_minibatch: nothing was prefetched")
        with L.on_device(self.device):
            torch.cuda.current_stream(self.device).wait_event(self._staged)
            self._staged = None
            if self._graph is not None:
                # a captured iteration has the ADDRESSES of x and uv baked into its kernel arguments: trading the
                # tensors would replay it on the old pair, which the next prefetch overwrites.  Copy into the fixed
                # pair instead (device to device, behind the wait above; the staging pair is free again afterwards).
                self.x.copy_(self._stage[0])
                self.uv.copy_(self._stage[1])
            else:
                (self.x, self.uv), self._stage = self._stage, (self.x, self.uv)
            for t in self.y:
                t.zero_()
        self.invalidate_forward()

    def invalidate_forward(self):
        """Call after changing parameters, inputs or multipliers behind the trainer's back (e.g. through
        ``view``): the next iteration recomputes its closure forward and its reconstruction terms."""
        self._saved_forward = False
        self._recon_ready = False
        self._prefetched = False

    # ------------------------------------------------------------------ one iteration
    def _closure_fwd_bwd(self):
        P = L.ptr
        with L.on_device(self.device):
            if (self.cfg.reuse_forward or self._prefetched) and self._saved_forward and self._graph is None:
                L.check(self.lib.lshm_engine_backward_saved(
                    self._h, P(self.params), P(self.grads), P(self.x), P(self.y[0]), P(self.y[1]), P(self.y[2]),
                    P(self.terms), P(self.ws), self.ws_floats, self._stream()), "engine_backward_saved")
            else:
                flags = L.STEP_RECON_READY if (self._recon_ready and self._saved_forward) else 0
                L.check(self.lib.lshm_engine_forward_backward_ex(
                    self._h, P(self.params), P(self.grads), P(self.x), P(self.uv), P(self.y[0]), P(self.y[1]),
                    P(self.y[2]), P(self.terms), P(self.ws), self.ws_floats, flags, self._stream()),
                    "engine_forward_backward")
            self._saved_forward = False  # whatever follows (optimiser, line search) moves the parameters
            self._recon_ready = False
            self._prefetched = False
            if self.world > 1 and self._comm is None:
                from .dist import allreduce_closure
                allreduce_closure(self.grads, self.terms, self.pg)
            if self._mask is not None:
                self.grads.mul_(self._mask)

    def _adam(self):
        """torch.optim.Adam over the selected groups only (frozen ranges are not touched at all: a parameter
        outside the optimiser's list never moves upstream, src/kharmonic_lofar.py:86-92)."""
        c = self.cfg
        P = L.ptr
        with L.on_device(self.device):
            in_graph = self._graph is not None
            if in_graph:
                self.step_count.add_(1)
            else:
                self.adam_steps += 1
            for a, b in self._ranges:
                L.check(self.lib.lshm_adam_step_flat(P(self.params[a:b]), P(self.grads[a:b]), P(self.exp_avg[a:b]),
                                                     P(self.exp_avg_sq[a:b]), b - a, c.lr, c.betas[0], c.betas[1],
                                                     c.adam_eps, P(self.step_count) if in_graph else None,
                                                     self.adam_steps, 1.0, self._stream()), "adam")

    def _multipliers(self, prepare_next: Optional[bool] = None):
        """No-grad forward + y_k += rho r_k (src/kharmonic_lofar.py:187-202).  prepare_next: the same pass also
        leaves the reconstruction terms of the next closure (default: cfg.share_recon_pass / reuse_forward)."""
        P = L.ptr
        if prepare_next is None:
            prepare_next = (self.cfg.share_recon_pass or self.cfg.reuse_forward)
        prepare_next = bool(prepare_next) and self._graph is None
        # two forwards side by side (cfg.overlap_forwards): only together with the shared reconstruction pass, and
        # pointless when the next closure re-uses this call's forward anyway (cfg.reuse_forward)
        concurrent = prepare_next and self.cfg.overlap_forwards and not self.cfg.reuse_forward
        with L.on_device(self.device):
            if prepare_next:
                L.check(self.lib.lshm_engine_multiplier_update_next_ex(
                    self._h, P(self.params), P(self.x), P(self.uv), P(self.y[0]), P(self.y[1]), P(self.y[2]),
                    P(self.ws), self.ws_floats, L.NEXT_CONCURRENT_FORWARD if concurrent else 0, self._stream()),
                    "engine_multiplier_update_next")
                concurrent = concurrent and bool(self.lib.lshm_engine_last_flags(self._h) & L.ENGINE_USED_CONCURRENT_FORWARD)
            else:
                L.check(self.lib.lshm_engine_multiplier_update(
                    self._h, P(self.params), P(self.x), P(self.uv), P(self.y[0]), P(self.y[1]), P(self.y[2]),
                    P(self.ws), self.ws_floats, self._stream()), "engine_multiplier_update")
        self._saved_forward = True  # forward of the parameters the next closure will see
        self._recon_ready = prepare_next
        self._prefetched = concurrent  # ... computed for it on purpose, beside the no-grad forward

    def _step_impl(self):
        self._closure_fwd_bwd()
        self._adam()
        self._multipliers()

    def capture_graph(self, warmup: int = 2):
        """Capture one iteration in a HIP graph (state is restored afterwards; a replayed iteration
        always recomputes its closure forward)."""
        self._saved_forward = False
        self._recon_ready = False
        with L.on_device(self.device):
            self.step_count.fill_(self.adam_steps)
            snap = [t.clone() for t in (self.params, self.exp_avg, self.exp_avg_sq, self.step_count, *self.y)]
            host_steps = self.adam_steps
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(warmup):
                    self._step_impl()
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            self._graph = g          # a captured iteration never relies on state left by an earlier call
            try:
                self._saved_forward = self._recon_ready = False
                with torch.cuda.graph(g):
                    self._step_impl()
            except Exception:
                self._graph = None
                raise
            finally:
                for t, v in zip((self.params, self.exp_avg, self.exp_avg_sq, self.step_count, *self.y), snap):
                    t.copy_(v)
                self.adam_steps = host_steps
                self._saved_forward = self._recon_ready = False

    def step(self):
        """One ADMM iteration.  Loss terms of the closure stay on the device (``read_terms``)."""
        if self._graph is not None:
            self._graph.replay()
            self.adam_steps += 1
            self._saved_forward = self._recon_ready = False
        else:
            self._step_impl()

    # ------------------------------------------------------------------ LBFGS (src/kharmonic_lofar.py:93)
    def make_lbfgs(self, history_size=7, max_iter=4, line_search_fn=True, batch_mode=True, **kw):
        """LBFGSNew over the flat arena (defaults = the commented-out line 93 of the upstream script).
        The arena is exposed as ONE parameter whose .grad is the engine's gradient buffer.
        `reuse_known_loss=True` (LBFGSNew option, off by default as upstream re-evaluates): the engine's closure is
        deterministic, so the line search may start from the loss step() has just computed at the same point."""
        from .lbfgsnew import LBFGSNew
        self._flat_param = torch.nn.Parameter(self.params, requires_grad=True)
        self._flat_param.grad = self.grads
        opt = LBFGSNew([self._flat_param], history_size=history_size, max_iter=max_iter,
                       line_search_fn=line_search_fn, batch_mode=batch_mode, **kw)
        # frozen groups enter as zero gradient coordinates; curvature pairs gathered under another selection
        # would move them, so an optimiser is tied to the selection it was made under
        opt._lshm_generation = self._opt_generation
        return opt

    def lbfgs_closure(self):
        """Closure protocol of the upstream script (:132-182): gradients only when autograd is enabled
        (LBFGSNew disables it inside the line search); returns the (all-reduced) total loss."""
        if torch.is_grad_enabled():
            self._closure_fwd_bwd()
        else:
            self.invalidate_forward()  # trial point of the line search
            P = L.ptr
            with L.on_device(self.device):
                L.check(self.lib.lshm_engine_forward_loss(
                    self._h, P(self.params), P(self.x), P(self.uv), P(self.y[0]), P(self.y[1]), P(self.y[2]),
                    P(self.terms), P(self.ws), self.ws_floats, self._stream()), "engine_forward_loss")
                if self.world > 1 and self._comm is None:
                    import torch.distributed as dist
                    dist.all_reduce(self.terms, group=self.pg)
        return self.terms[8]

    def step_lbfgs(self, opt):
        """One ADMM iteration with the LBFGS update instead of Adam."""
        if getattr(opt, "_lshm_generation", self._opt_generation) != self._opt_generation:
            raise RuntimeError("this LBFGS optimiser was made before set_train_groups(); call make_lbfgs() again")
        if self._flat_param.grad is not self.grads:
            self._flat_param.grad = self.grads
        with L.on_device(self.device):
            opt.step(self.lbfgs_closure)
        self._multipliers()

    def closure_only(self):
        """Closure forward + backward without the update (gradients in ``self.grads``)."""
        self._closure_fwd_bwd()

    def read_terms(self) -> Dict[str, float]:
        """The reference's log line (src/kharmonic_lofar.py:176-181) as a dict: one device->host copy of ten
        doubles.  ``nonfinite`` counts the logged terms that are NaN or infinite (job-wide under data
        parallelism): the cheap divergence check upstream leaves to the reader of the log (README.md:29)."""
        t = self.terms[:10].cpu().tolist()
        d = dict(zip(TERM_NAMES, t[:9]))
        d["nonfinite"] = t[9]
        return d

    def format_log(self, epoch: int, i: int, admm: int) -> str:
        return format_terms(self.read_terms(), epoch, i, admm, self.cfg.use_rica)

    # ------------------------------------------------------------------ inference helper
    def encode(self, want_recon: bool = False):
        """Latents Mu = [mu | muT | muF] (B, L+2Lt) for the current x, uv (no grad)."""
        self.invalidate_forward()  # conservative: the encode pass re-uses the activation workspace
        c = self.cfg
        Mu = torch.empty((self.B, c.L + 2 * c.Lt), device=self.device)
        outs = [torch.empty_like(self.x) for _ in range(3)] if want_recon else [None] * 3
        P = L.ptr
        self._recon_ready = False
        with L.on_device(self.device):
            L.check(self.lib.lshm_engine_encode(self._h, P(self.params), P(self.x), P(self.uv), P(Mu), P(outs[0]),
                                                P(outs[1]), P(outs[2]), P(self.ws), self.ws_floats, self._stream()),
                    "engine_encode")
        return (Mu, *outs) if want_recon else Mu
