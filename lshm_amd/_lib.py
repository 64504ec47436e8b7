"""ctypes binding of liblshm_hip.so (C ABI declared in include/lshm.h).

The shared library is built in-tree by ``make`` / ``__graft_entry__.build()``.
There is NO fallback: if the library is missing or no HIP device is present the
product path raises.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# LSHM_LIB lets a developer A/B two builds of the same C ABI in one process launch
LIB_PATH = os.environ.get("LSHM_LIB") or os.path.join(_HERE, "lib", "liblshm_hip.so")
TUNE_FILE = os.path.join(_HERE, "tuned_gfx950.txt")

_lib = None

c_float_p = C.c_void_p  # device pointers travel as integers
c_long = C.c_long
c_int = C.c_int
c_size_t = C.c_size_t
c_float = C.c_float
c_double = C.c_double
c_void_p = C.c_void_p


class StepConfig(C.Structure):
    """Mirror of lshm_step_config (include/lshm.h)."""
    _fields_ = [("B", c_int), ("C", c_int), ("P", c_int), ("L", c_int), ("Lt", c_int), ("K", c_int),
                ("p", c_float), ("alpha", c_float), ("beta", c_float), ("gamma", c_float),
                ("rho", c_float), ("rica_lambda", c_float), ("rica", c_int), ("bpb", c_int),
                ("batch_size", c_int), ("H", c_int), ("scales", c_float * 8), ("world", c_int),
                ("precision", c_int), ("schedule", C.c_uint), ("tune", C.c_uint)]


PRECISION_F32, PRECISION_BF16_OPERANDS, PRECISION_BF16_STORAGE = 0, 1, 2
# lshm_step_config.schedule (LSHM_SCHED_* of include/lshm.h): name -> bit
SCHEDULE_BITS = {"no_deep2d": 1 << 0, "no_deep2d_bwd": 1 << 1, "no_wgrad_batch": 1 << 2, "try_full1d": 1 << 3,
                 "no_chain1d": 1 << 4, "no_chain1d_bwd": 1 << 5, "no_dense1d": 1 << 6, "no_dense1d_bwd": 1 << 7,
                 "no_resid_conv0": 1 << 8, "no_recon_from_a": 1 << 9, "no_one_pass_bwd": 1 << 10, "no_bwd_lds": 1 << 11,
                 "no_bwd_lds_8_4": 1 << 12, "no_bwd_lds2d": 1 << 13, "no_bwd_fused2d": 1 << 14, "no_wgrad_mid": 1 << 15,
                 "no_stop_events": 1 << 16, "wgrad_inline": 1 << 17, "fork": 1 << 18, "phase_events": 1 << 19,
                 "no_khm_mfma": 1 << 20, "no_early_latent": 1 << 21, "no_resid_conv0_keep": 1 << 22,
                 "no_conv0_bwd_tile": 1 << 23, "no_recon_bwd5": 1 << 24, "no_shared_pack": 1 << 25}
SCHED_NO_DEEP2D, SCHED_NO_DEEP2D_BWD = SCHEDULE_BITS["no_deep2d"], SCHEDULE_BITS["no_deep2d_bwd"]
STEP_RECON_READY = 1
NEXT_CONCURRENT_FORWARD = 1
ENGINE_USED_EARLY_BUCKET, ENGINE_USED_CONCURRENT_FORWARD = 1, 2


_SIGNATURES = {
    "lshm_version": (c_int, []),
    "lshm_last_error_string": (C.c_char_p, []),
    "lshm_set_tuning": (None, [c_int, c_int]),
    "lshm_tuning_export": (c_size_t, [c_void_p, c_size_t]),
    "lshm_tuning_import": (c_int, [C.c_char_p]),
    "lshm_uv_harmonics": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "lshm_conv_workspace_floats": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "lshm_conv_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                              c_int, c_long, c_long, c_int, c_void_p, c_size_t, c_void_p]),
    "lshm_conv_fwd_pair": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_int, c_int, c_int, c_int, c_int, c_long, c_long, c_int, c_void_p,
                                   c_size_t, c_void_p]),
    "lshm_conv_dgrad": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                c_int, c_long, c_long, c_void_p, c_size_t, c_void_p]),
    "lshm_conv_wgrad": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                c_int, c_long, c_long, c_void_p, c_size_t, c_int, c_void_p]),
    "lshm_conv_bwd_fused": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                    c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "lshm_conv_bwd_fused_ex": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                       c_int, c_int, c_int, c_void_p, c_size_t, C.c_uint, c_void_p]),
    "lshm_conv1d_chain3": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "lshm_dense1d_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_int, c_void_p]),
    "lshm_dense1d_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lshm_deep2d_packed_floats": (c_size_t, []),
    "lshm_deep2d_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "lshm_deep2d_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "lshm_trace_begin": (c_int, [c_int]),
    "lshm_trace_begin_ex": (c_int, [c_int, c_int]),
    "lshm_trace_end": (c_int, []),
    "lshm_trace_read": (c_int, [c_int, C.c_char_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lshm_trace_free": (c_int, []),
    "lshm_chain1d_full_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p, c_void_p]),
    "lshm_elu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "lshm_linear_workspace_floats": (c_size_t, [c_int, c_int, c_int]),
    "lshm_linear_fwd": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_int,
                                c_int, c_void_p, c_size_t, c_void_p]),
    "lshm_linear_dgrad": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p, c_long, c_int,
                                  c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "lshm_linear_wgrad": (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_int, c_int,
                                  c_int, c_void_p, c_size_t, c_void_p]),
    "lshm_khm_workspace_floats": (c_size_t, [c_int, c_int, c_int]),
    "lshm_khm_fwd_bwd": (c_int, [c_void_p, c_long, c_void_p, c_int, c_int, c_int, c_float, c_float,
                                 c_double, c_float, c_void_p, c_void_p, c_long, c_void_p, c_int, c_void_p,
                                 c_size_t, c_void_p]),
    "lshm_khm_offline_partials": (c_int, [c_void_p, c_long, c_void_p, c_int, c_int, c_int, c_float, c_float,
                                          c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "lshm_khm_mean_distances": (c_int, [c_void_p, c_long, c_void_p, c_int, c_int, c_int, c_float, c_void_p,
                                        c_void_p, c_size_t, c_void_p]),
    "lshm_khm_assign": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "lshm_cluster_sim_fwd_bwd": (c_int, [c_void_p, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_int,
                                         c_void_p]),
    "lshm_aug_loss_fwd_bwd": (c_int, [c_void_p, c_long, c_int, c_int, c_int, c_int, c_float, c_void_p,
                                      c_void_p, c_long, c_int, c_void_p]),
    "lshm_logcosh_fwd_bwd": (c_int, [c_void_p, c_long, c_int, c_int, c_float, c_void_p, c_void_p, c_long,
                                     c_int, c_void_p]),
    "lshm_residual_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "lshm_resid_conv0": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lshm_recon_bwd5_workspace_floats": (c_size_t, [c_int]),
    "lshm_recon_bwd5": (c_int, [c_void_p] * 11 + [C.c_float, c_int] + [c_void_p] * 9 + [c_size_t, c_int, c_void_p]),
    "lshm_tconv5_pair_bwd": (c_int, [c_void_p] * 12 + [c_int, c_void_p, c_size_t, c_int, c_void_p]),
    "lshm_conv0_bwd_tile_workspace_floats": (c_size_t, []),
    "lshm_conv0_bwd_tile": (c_int, [c_void_p] * 11 + [c_int, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "lshm_resid_conv0_keep": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_int, c_void_p]),
    "lshm_plane_transpose": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "lshm_rica_workspace_floats": (c_size_t, [c_int, c_int, c_int]),
    "lshm_rica_loss_grad": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_float, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "lshm_rica_update_dictionary": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "lshm_recon_workspace_floats": (c_size_t, [c_int, c_int]),
    "lshm_recon_losses_fwd_bwd": (c_int, [c_void_p] * 7 + [c_float, c_int, c_int] + [c_void_p] * 5 + [c_void_p]),
    "lshm_recon_losses_from_a": (c_int, [c_void_p] * 11 + [c_float, c_int, c_int, c_int] + [c_void_p] * 5 + [c_void_p]),
    "lshm_combine_dx1": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "lshm_multiplier_update": (c_int, [c_void_p] * 7 + [c_float, c_int, c_int, c_void_p]),
    "lshm_adam_step_flat": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_float, c_float, c_float,
                                    c_float, c_void_p, c_int, c_float, c_void_p]),
    "lshm_axpy_flat": (c_int, [c_void_p, c_void_p, c_float, c_long, c_void_p]),
    "lshm_scale_flat": (c_int, [c_void_p, c_float, c_long, c_void_p]),
    "lshm_dot_flat": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_void_p]),
    "lshm_asum_flat": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p]),
    "lshm_multi_dot_workspace_doubles": (c_size_t, [c_int]),
    "lshm_multi_dot_flat": (c_int, [c_void_p, c_void_p, c_int, c_long, c_void_p, c_void_p, c_size_t, c_void_p]),
    "lshm_lbfgs_direction_workspace_doubles": (c_size_t, [c_int]),
    "lshm_lbfgs_direction": (c_int, [c_void_p, c_void_p, c_int, c_void_p, C.c_double, c_void_p, c_long, c_void_p,
                                     c_size_t, c_void_p]),
    "lshm_patches_workspace_floats": (c_size_t, []),
    "lshm_patches_from_vis": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p,
                                      c_void_p, c_void_p, c_void_p]),
    "lshm_patches_from_vis_ex": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_int,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lshm_patches_normalize": (c_int, [c_void_p, c_long, c_void_p, c_void_p]),
    "lshm_fft2_ortho_shift_cat_clamp": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "lshm_comm_available": (c_int, []),
    "lshm_comm_unique_id": (c_int, [C.c_char_p]),
    "lshm_comm_init": (c_int, [C.c_char_p, c_int, c_int, C.POINTER(c_void_p)]),
    "lshm_comm_destroy": (None, [c_void_p]),
    "lshm_comm_rank": (c_int, [c_void_p]),
    "lshm_comm_world": (c_int, [c_void_p]),
    "lshm_comm_allreduce_flat": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "lshm_engine_set_comm": (c_int, [c_void_p, c_void_p]),
    "lshm_fft2_backward_workspace_floats": (c_size_t, [c_int, c_int]),
    "lshm_fft2_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_size_t, c_void_p]),
    "lshm_engine_create": (c_int, [C.POINTER(StepConfig), C.POINTER(c_void_p)]),
    "lshm_engine_destroy": (None, [c_void_p]),
    "lshm_engine_param_count": (c_long, [c_void_p]),
    "lshm_engine_param_lookup": (c_int, [c_void_p, C.c_char_p, C.POINTER(c_long), C.POINTER(c_long)]),
    "lshm_engine_param_name": (c_int, [c_void_p, c_int, C.c_char_p, c_int, C.POINTER(c_long),
                                       C.POINTER(c_long), C.POINTER(c_int), C.POINTER(c_long)]),
    "lshm_engine_workspace_floats": (c_size_t, [c_void_p]),
    "lshm_engine_forward_backward": (c_int, [c_void_p] * 9 + [c_void_p, c_size_t, c_void_p]),
    "lshm_engine_forward_backward_ex": (c_int, [c_void_p] * 9 + [c_void_p, c_size_t, C.c_uint, c_void_p]),
    "lshm_engine_forward_loss": (c_int, [c_void_p] * 8 + [c_void_p, c_size_t, c_void_p]),
    "lshm_engine_backward_saved": (c_int, [c_void_p] * 8 + [c_void_p, c_size_t, c_void_p]),
    "lshm_engine_multiplier_update": (c_int, [c_void_p] * 7 + [c_void_p, c_size_t, c_void_p]),
    "lshm_engine_multiplier_update_next": (c_int, [c_void_p] * 7 + [c_void_p, c_size_t, c_void_p]),
    "lshm_engine_multiplier_update_next_ex": (c_int, [c_void_p] * 7 + [c_void_p, c_size_t, C.c_uint, c_void_p]),
    "lshm_engine_device": (c_int, [c_void_p]),
    "lshm_engine_set_schedule": (C.c_uint, [c_void_p, C.c_uint]),
    "lshm_engine_phase_times": (c_int, [c_void_p, c_void_p, c_int]),
    "lshm_engine_last_flags": (C.c_uint, [c_void_p]),
    "lshm_engine_comm_early_bucket": (c_int, [c_void_p]),
    "lshm_engine_set_early_bucket": (c_int, [c_void_p, c_int]),
    "lshm_engine_encode": (c_int, [c_void_p] * 8 + [c_void_p, c_size_t, c_void_p]),
}
# `_bf16` forms of the GEMM-shaped entry points: same prototypes (include/lshm.h)
for _n in ("lshm_conv_fwd", "lshm_conv_dgrad", "lshm_conv_wgrad", "lshm_linear_fwd", "lshm_linear_dgrad",
           "lshm_linear_wgrad", "lshm_rica_loss_grad", "lshm_rica_update_dictionary"):
    _SIGNATURES[_n + "_bf16"] = _SIGNATURES[_n]
del _n

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def load():
    """dlopen the library and attach prototypes (no device needed for this)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"lshm_amd: HIP library not built ({LIB_PATH} missing). Run `make` or "
            f"`python -c 'import __graft_entry__ as g; g.build()'` - there is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)  # AttributeError -> a declared symbol is missing
        except AttributeError:
            if not os.environ.get("LSHM_LIB"):
                raise
            # a developer's A/B against an older build of the same ABI: say so, and fail at the call instead
            print(f"lshm_amd: {LIB_PATH} does not export {name}", file=sys.stderr)
            continue
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    # tile configurations measured on the target GPU (bench.py --save-tuning): no timing launches, and
    # every run picks the same kernels.  LSHM_TUNE_FILE overrides the path; empty disables.
    path = os.environ.get("LSHM_TUNE_FILE", TUNE_FILE)
    if path and os.path.exists(path):
        with open(path, "rb") as f:
            lib.lshm_tuning_import(f.read())
    return lib


def save_tuning(path: str) -> int:
    """Write the tile-configuration cache of this process to `path` (merged with what was imported)."""
    lib = load()
    n = lib.lshm_tuning_export(None, 0)
    buf = C.create_string_buffer(n)
    lib.lshm_tuning_export(buf, n)
    lines = sorted(set(buf.value.decode().splitlines()), key=lambda l: [int(v) for v in l.split()])
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return len(lines)


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().lshm_last_error_string().decode("utf-8", "replace")
        raise RuntimeError(f"lshm_amd: {what or 'kernel call'} failed (code {rc}): {msg}")


def require_device(*tensors: torch.Tensor):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("lshm_amd: tensors must live on a HIP device (cuda:N); there is no CPU path")
        if t.dtype != torch.float32:
            raise RuntimeError(f"lshm_amd: expected float32 tensors, got {t.dtype}")


def ptr(t):
    return None if t is None else t.data_ptr()


_scratch = {}


def scratch(device, nfloats: int):
    """Split-K scratch of the autograd wrappers.  The DEFAULT stream of a device keeps one persistent buffer
    (grown on demand): work on one stream is ordered, so its launches may share it.  Any other stream gets a
    fresh tensor from torch's caching allocator per call -- stream-safe by construction (the allocator ties the
    block to the stream that is current now), effectively free after warm-up, and nothing is pinned for a
    stream that has gone away (a raw stream handle can be re-used by a later stream)."""
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    if torch.cuda.current_stream(index) != torch.cuda.default_stream(index):
        return torch.empty(max(int(nfloats), 1), device=torch.device("cuda", index), dtype=torch.float32)
    t = _scratch.get(index)
    if t is None or t.numel() < nfloats:
        t = torch.empty(max(int(nfloats), 1 << 20), device=torch.device("cuda", index), dtype=torch.float32)
        _scratch[index] = t
    return t


def stream(device=None):
    """hipStream_t of torch's current stream on `device` (default: the current device)."""
    return torch.cuda.current_stream(device).cuda_stream


def on_device(device):
    """Context manager: `device` is the current HIP device inside (kernels are launched on the current
    device; a tensor or stream of another one would fault or cross the fabric silently)."""
    return torch.cuda.device(device)


def fn(name: str, bf16: bool = False):
    """Entry point `name`, or its `_bf16` form (operands rounded to bf16 in the GEMM-shaped kernels)."""
    return getattr(load(), name + "_bf16" if bf16 else name)
