"""ctypes binding of libmovae_hip.so (include/movae.h).

The product path has no CPU fallback: if the library cannot be loaded every op raises.
"""
import ctypes as C
import os
import threading

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmovae_hip.so")

ACT = {"none": 0, None: 0, "lrelu": 1, "relu": 2, "tanh": 3, "sigmoid": 4}
RECON = {"mse": 0, "bce": 1, "l1": 2, "smooth_l1": 3}
UPGRAD_NORM = {"trace": 0, "min_l2": 1, "cosine": 2}
EDGE_MATCH = {"mag": 0, "signed_mse": 1, "maxnorm": 2, "angle": 3, "masked": 4, "cosine": 5}  # enum movae_edge_match
MGDA_NORM = {"none": 0, "l2": 1, "loss": 2, "loss+": 3}
AMTL_SCALE = {"min": 0, "median": 1, "rmse": 2}
MAX_K = 8

_p, _i, _f, _z, _l = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_long

class MovaeFuse(C.Structure):
    """movae_fuse_t (include/movae.h): BatchNorm fused into the neighbouring convolutions."""
    _fields_ = [("in_scale", C.c_void_p), ("in_shift", C.c_void_p), ("in_slope", C.c_float), ("stats", C.c_void_p),
                ("stats_cap", C.c_size_t), ("stats_parts", C.c_int),
                ("bn_y", C.c_void_p), ("bn_scale", C.c_void_p), ("bn_shift", C.c_void_p), ("bn_slope", C.c_float), ("bn_part", C.c_void_p),
                ("bn_cap", C.c_size_t), ("bn_ppg", C.c_int),
                ("ep_act_y", C.c_void_p), ("ep_act", C.c_int), ("ep_slope", C.c_float), ("ep_act_done", C.c_int),
                ("ep_res", C.c_void_p),
                ("fin_gamma", C.c_void_p), ("fin_beta", C.c_void_p), ("fin_eps", C.c_float), ("fin_momentum", C.c_float),
                ("fin_out", C.c_void_p), ("fin_running_mean", C.c_void_p), ("fin_running_var", C.c_void_p), ("fin_nbt", C.c_void_p),
                ("fin_done", C.c_int)]


_conv_fwd = [_p, _p, _p, _p] + [_i] * 11 + [_i, _f, _p, _z, _p]
_conv_dgrad = [_p, _p, _p] + [_i] * 11 + [_p, _z, _p]
_conv_wgrad = [_p, _p, _p, _p] + [_i] * 11 + [_i, _p, _z, _p]

SIGNATURES = {
    "movae_version": ([], _i),
    "movae_last_error": ([], C.c_char_p),
    "movae_nchw_to_nhwc": ([_p, _p, _i, _i, _i, _i, _p], _i),
    "movae_nhwc_to_nchw": ([_p, _p, _i, _i, _i, _i, _p], _i),
    "movae_conv2d_fwd": (_conv_fwd, _i),
    "movae_conv2d_dgrad": (_conv_dgrad, _i),
    "movae_conv2d_wgrad": (_conv_wgrad, _i),
    "movae_convT2d_fwd": (_conv_fwd, _i),
    "movae_convT2d_dgrad": (_conv_dgrad, _i),
    "movae_convT2d_wgrad": (_conv_wgrad, _i),
    "movae_conv2d_wgrad_grouped": ([_i] + _conv_wgrad, _i),
    "movae_convT2d_wgrad_grouped": ([_i] + _conv_wgrad, _i),
    "movae_conv2d_dgrad_wgrad_grouped": ([_i, _p, _p, _p, _p, _p, _p] + [_i] * 11 + [_i, _p, _z, _p], _i),
    "movae_convT2d_dgrad_wgrad_grouped": ([_i, _p, _p, _p, _p, _p, _p] + [_i] * 11 + [_i, _p, _z, _p], _i),
    "movae_bn_ws_bytes": ([_i, _i], _z),
    "movae_bn_act_fwd": ([_p] * 9 + [_i, _i, _f, _f, _i, _i, _f, _p, _z, _p], _i),
    "movae_bn_act_bwd": ([_p] * 9 + [_i, _i, _i, _f, _i, _p, _z, _p], _i),
    "movae_bn_act_bwd_grouped": ([_i] + [_p] * 9 + [_i, _i, _i, _f, _i, _p, _z, _p], _i),
    "movae_act_fwd": ([_p, _p, _z, _i, _f, _p], _i),
    "movae_act_bwd": ([_p, _p, _p, _z, _i, _f, _p], _i),
    "movae_act_bwd_bias_grouped": ([_i, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p, _z, _p], _i),
    "movae_add": ([_p, _p, _p, _z, _p], _i),
    "movae_axpby": ([_f, _p, _f, _p, _p, _z, _p], _i),
    "movae_copy_channels": ([_p, _p, _i, _i, _i, _i, _i, _i, _p], _i),
    "movae_colsum": ([_p, _p, _i, _i, _i, _p, _z, _p], _i),
    "movae_reparam_fwd": ([_p, _p, _p, _p, _z, _p], _i),
    "movae_reparam_rng_fwd": ([_p, _p, _p, _p, _z, _p, _i, _p], _i),
    "movae_reparam_bwd": ([_p, _p, _p, _p, _p, _z, _p], _i),
    "movae_reduce_ws_bytes": ([_z], _z),
    "movae_recon_loss_fwd": ([_p, _p, _p, _z, _i, _f, _p, _z, _p], _i),
    "movae_recon_loss_bwd": ([_p, _p, _p, _p, _z, _i, _f, _p], _i),
    "movae_recon_loss_bwd_act": ([_p, _p, _p, _p, _z, _i, _f, _i, _f, _p], _i),
    "movae_kl_fwd": ([_p, _p, _p, _i, _i, _f, _p, _z, _p], _i),
    "movae_combine_losses_fwd": ([_i, _p, _i, _p, _p, _f, _i, _i, _p, _p, _p], _i),
    "movae_combine_losses_bwd": ([_i, _i, _p, _p, _p, _i, _p, _p], _i),
    "movae_vae_losses_fwd": ([_p, _p, _z, _i, _f, _p, _p, _i, _i, _f, _p, _p, _z, _p], _i),
    "movae_kl_bwd": ([_p, _p, _p, _p, _p, _i, _i, _f, _p], _i),
    "movae_tc_decomp_fwd": ([_p] * 7 + [_i, _i, _p, _z, _p], _i),
    "movae_tc_decomp_bwd": ([_p] * 10 + [_i, _i, _p], _i),
    "movae_edge_weights": ([_p, _p, _p, _i, _i, _i, _i, _p, _z, _p], _i),
    "movae_edge_weighted_mse_fwd": ([_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p, _z, _p], _i),
    "movae_edge_weighted_mse_bwd": ([_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p], _i),
    "movae_edge_match_fwd": ([_p, _p, _p, _i, _i, _i, _i, _f, _i, _p, _p, _z, _p], _i),
    "movae_edge_match_bwd": ([_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p, _p], _i),
    "movae_vq_nearest_fwd": ([_p] * 6 + [_i, _i, _i, _p, _z, _p], _i),
    "movae_vq_nearest_fwd_mse": ([_p] * 7 + [_i, _i, _i, _p, _z, _p], _i),
    "movae_vq_bwd_ws_bytes": ([_i, _i, _i], _z),
    "movae_vq_bwd": ([_p] * 8 + [_i, _i, _i, _p, _z, _p], _i),
    "movae_gram_ws_bytes": ([_i, _z], _z),
    "movae_gram": ([_p, _z, _i, _z, _p, _p, _z, _p], _i),
    "movae_weights_upgrad": ([_p, _i, _f, _f, _p, _p, _p], _i),
    "movae_weights_upgrad_norm": ([_p, _i, _i, _f, _f, _p, _p, _p], _i),
    "movae_gram_upgrad": ([_p, _z, _i, _z, _p, _i, _f, _f, _p, _p, _i, _p, _z, _p], _i),
    "movae_weights_mgda": ([_p, _i, _i, _p, _f, _i, _p, _p, _p], _i),
    "movae_weights_mgda_stable": ([_p, _i, _i, _p, _f, _i, _f, _p, _p, _p], _i),
    "movae_weights_amtl": ([_p, _i, _i, _p, _p, _p], _i),
    "movae_weights_dualproj": ([_p, _i, _f, _f, _p, _p, _p], _i),
    "movae_weights_pcgrad": ([_p, _i, _p, _p, _p], _i),
    "movae_weights_imtlg": ([_p, _i, _p, _p], _i),
    "movae_weights_cagrad": ([_p, _i, _f, _f, _p, _p], _i),
    "movae_weights_const": ([_i, _f, _p, _p], _i),
    "movae_combine": ([_p, _z, _i, _z, _p, _p, _i, _p], _i),
    "movae_gd_similarity": ([_p, _z, _i, _z, _p, _p, _p, _z, _p], _i),
    "movae_adam_step": ([_p, _p, _p, _p, _z, _f, _f, _f, _f, _f, _i, _i, _p], _i),
    "movae_adam_multi": ([_i, _p, _p, _p, _p, _p, _f, _f, _f, _f, _f, _i, _i, _p, _p], _i),
    "movae_clip_grad_norm_multi": ([_i, _p, _p, _f, _p, _p, _z, _p], _i),
    "movae_sumsq": ([_p, _z, _p, _p, _z, _p], _i),
    "movae_scale_by_clip": ([_p, _z, _p, _f, _p], _i),
    "movae_conv2d_fwd_f": (_conv_fwd + [_p], _i),
    "movae_convT2d_fwd_f": (_conv_fwd + [_p], _i),
    "movae_conv2d_wgrad_grouped_f": ([_i] + _conv_wgrad + [_p], _i),
    "movae_convT2d_wgrad_grouped_f": ([_i] + _conv_wgrad + [_p], _i),
    "movae_conv2d_dgrad_wgrad_grouped_f": ([_i, _p, _p, _p, _p, _p, _p] + [_i] * 11 + [_i, _p, _z, _p, _p], _i),
    "movae_convT2d_dgrad_wgrad_grouped_f": ([_i, _p, _p, _p, _p, _p, _p] + [_i] * 11 + [_i, _p, _z, _p, _p], _i),
    "movae_conv2d_dgrad_f": (_conv_dgrad + [_p, _i], _i),
    "movae_convT2d_dgrad_f": (_conv_dgrad + [_p, _i], _i),
    "movae_bn_bwd_finalize": ([_p, _z, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _i, _p], _i),
    "movae_bn_bwd_apply": ([_p, _p, _p, _p, _f, _p, _p, _i, _z, _i, _p], _i),
    "movae_linear_pair_fwd": ([_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p], _i),
    "movae_linear_pair_bwd": ([_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p], _i),
    "movae_bn_bwd_finalize_apply": ([_p, _z, _i, _i, _z, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _f, _p, _p], _i),
    "movae_bn_finalize": ([_p, _z, _i, _i, _i, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _p, _p], _i),
    "movae_bn_stats": ([_p, _i, _i, _p, _z, _p, _p], _i),
    "movae_scale_shift_act": ([_p, _p, _p, _p, _z, _i, _f, _p], _i),
    "movae_embedding_fwd": ([_p, _p, _p, _z, _i, _i, _p], _i),
    "movae_embedding_bwd": ([_p, _p, _p, _i, _i, _i, _p, _z, _p], _i),
    "movae_gated_residual_fwd": ([_p, _p, _p, _p, _z, _p], _i),
    "movae_gated_residual_bwd": ([_p, _p, _p, _p, _p, _z, _p], _i),
    "movae_mul": ([_p, _p, _p, _z, _p], _i),
    "movae_cross_entropy_ws_bytes": ([_z], _z),
    "movae_cross_entropy_fwd": ([_p, _p, _p, _p, _z, _i, _p, _z, _p], _i),
    "movae_cross_entropy_bwd": ([_p, _p, _p, _p, _p, _z, _i, _p], _i),
    "movae_bench_main_kernel_only": ([_i], _i),
    "movae_bench_last_kernel": ([], C.c_char_p),
    "movae_reduce_defer": ([_i], _i),
    "movae_reduce_flush": ([], _i),
    "movae_reduce_defer_stats": ([_p, _i], _i),
    "movae_reduce_defer_max_bytes": ([C.c_longlong], C.c_longlong),
    "movae_bench_force_split": ([_i], _i),
    "movae_bench_kgemm_bn_fin": ([_i], _i),
    "movae_bench_force_kgemm": ([_i], _i),
    "movae_set_compute_dtype": ([_i], _i),
}

_lib = None
_lock = threading.Lock()


def load():
    """Loads the shared library (building it with hipcc when the toolchain is present and the
    .so is absent).  Raises RuntimeError otherwise -- there is deliberately no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        # with the toolchain present the build is always consulted: build.build() compares a digest of every source with the
        # one each object was compiled from and recompiles what changed, so an edited csrc/ never runs against a stale .so;
        # without hipcc (a box that only received the built library) the library must already exist
        from . import build as _build

        have_hipcc = False
        try:
            _build._hipcc()
            have_hipcc = os.path.isdir(_build.CSRC)
        except RuntimeError:
            pass
        if have_hipcc and os.environ.get("MOVAE_NO_REBUILD") != "1":
            try:
                _build.build(verbose=False)
            except Exception as e:  # noqa: BLE001
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"libmovae_hip.so is missing at {LIB_PATH} and could not be built ({e}); "
                        "the MI355X kernels are required -- there is no CPU fallback") from e
                raise RuntimeError(f"csrc/ changed but rebuilding libmovae_hip.so failed: {e}") from e
        elif not os.path.exists(LIB_PATH):
            raise RuntimeError(f"libmovae_hip.so is missing at {LIB_PATH} and hipcc is not available to build it; "
                               "the MI355X kernels are required -- there is no CPU fallback")
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:
            raise RuntimeError(f"cannot load {LIB_PATH}: {e}; the MI355X kernels are required") from e
        for name, (args, res) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export the declared symbol
            fn.argtypes = args
            fn.restype = res
        _lib = lib
    return _lib


DTYPES = {"fp32": 0, "f32": 0, "float32": 0, "bf16": 1, "bfloat16": 1}


def set_compute_dtype(name):
    """"fp32" (default, the parity path) | "bf16": bf16 operands / fp32 accumulation in the 128x128 implicit-GEMM kernels
    (include/movae.h: movae_set_compute_dtype).  Returns the previous setting's name."""
    if name not in DTYPES:
        raise ValueError(f"compute dtype must be one of {sorted(set(DTYPES))}, got {name!r}")
    prev = load().movae_set_compute_dtype(DTYPES[name])
    return "bf16" if prev == 1 else "fp32"


TRACE = None  # optional callable(name, args) invoked for every C-ABI launch (bench.py's recorder)

#: Deferred weight-gradient reduces (include/movae.h: movae_reduce_defer; ops.deferred_reduces is the switch).  While a reduce may
#: be parked, the calls listed here go ahead -- the backward's own ops: they neither read a convolution's weight / bias gradient nor
#: write the parked call's scratch arena (armed calls get arenas 2 / 3 in turn, everything else arena 0), and the implicit-GEMM
#: ones among them carry the parked reduce along.  Any other call -- the aggregation, the optimizer, a forward op -- first makes
#: the library launch it stand-alone (movae_reduce_flush).
DEFER_ON = [False]
DEFER_PASS = frozenset(
    [f"movae_conv{t}2d_{k}" for t in ("", "T") for k in ("dgrad", "dgrad_f", "wgrad_grouped", "wgrad_grouped_f", "dgrad_wgrad_grouped",
                                                       "dgrad_wgrad_grouped_f")] +
    ["movae_bn_bwd_finalize", "movae_bn_bwd_apply", "movae_bn_bwd_finalize_apply", "movae_bn_act_bwd", "movae_bn_act_bwd_grouped",
     "movae_act_bwd", "movae_act_bwd_bias_grouped", "movae_colsum", "movae_add", "movae_axpby", "movae_copy_channels", "movae_mul",
     "movae_nchw_to_nhwc", "movae_nhwc_to_nchw", "movae_reparam_bwd", "movae_kl_bwd", "movae_recon_loss_bwd", "movae_recon_loss_bwd_act", "movae_tc_decomp_bwd",
     "movae_combine_losses_bwd", "movae_vq_bwd", "movae_linear_pair_bwd", "movae_edge_weighted_mse_bwd", "movae_edge_match_bwd",
     "movae_gated_residual_bwd"])
_defer_arena = [0]


def defer_arm(device):
    """Arm the next weight-gradient call (ops.Conv.backward*): -> (ws pointer, ws bytes) of the arena its slabs may keep until a
    later launch has carried the reduce.  Not traced: bench.py's replay of recorded calls launches every reduce stand-alone."""
    _defer_arena[0] ^= 1
    w = workspace(device, slot=2 + _defer_arena[0])
    load().movae_reduce_defer(1)
    return w.data_ptr(), w.numel()


def defer_flush():
    """Launch a still-parked reduce now (before anything outside this library reads a weight gradient)."""
    if _lib is not None:
        check(load().movae_reduce_flush(), "movae_reduce_flush")


def call(name, *args):
    """Invoke one C-ABI entry point, raising RuntimeError with the library's message on failure."""
    if TRACE is not None:
        TRACE(name, args)
    lib = load()
    if DEFER_ON[0] and name not in DEFER_PASS:
        check(lib.movae_reduce_flush(), "movae_reduce_flush")
    check(getattr(lib, name)(*args), name)


class Unsupported(RuntimeError):
    """rc == -2: the shape dispatches to a kernel without the requested fusion; nothing was launched (movae_fuse_t)."""


def check(rc, what=""):
    if rc == -2:
        raise Unsupported(f"{what or 'movae'}: {load().movae_last_error().decode(errors='replace')}")
    if rc != 0:
        msg = load().movae_last_error().decode(errors="replace")
        raise RuntimeError(f"{what or 'movae'} failed (rc={rc}): {msg}")


def stream_ptr(device=None):
    return torch.cuda.current_stream(device).cuda_stream


_workspaces = {}
WS_BYTES = 96 << 20


# Forking conv wgrad onto a second stream measured SLOWER on MI355X (C2 graph replay 2.42 ms vs 2.24 ms: the tiny
# layers are launch/latency bound and the fork/join events cost more than the overlap wins) -> opt-in only.
SIDE_STREAM_WGRAD = os.environ.get("MOVAE_SIDE_STREAM") == "fork"  # per-layer fork/join (measured slower; experiments)
#: deferred weight gradients (ops.wgrad_side_stream) in the training loop: off by default -- eager streams overlap the two
#: launches well (333 vs 462 us over the C2 layers) but a replayed hipGraph serialises its branches with extra cross-queue
#: waits (C2 step 1.397 ms vs 1.357 ms); the paired launch (igemm2_pair) gets the overlap inside one kernel instead
DEFER_WGRAD_DEFAULT = os.environ.get("MOVAE_SIDE_STREAM", "0") == "1"
DEFER = None  # the active ops.wgrad_side_stream block, if any
_side_streams = {}


def side_stream(device):
    """One forked HIP stream per device for launches that are independent of the main chain (conv wgrad)."""
    key = (device.type, device.index)
    s = _side_streams.get(key)
    if s is None:
        s = torch.cuda.Stream(device)
        _side_streams[key] = s
    return s


def workspace(device, slot=0):
    """Persistent per-device scratch (split-K slabs, reduction partials).  Stream-ordered reuse; `slot` 1 is the
    side stream's own arena (a workspace serves one stream at a time)."""
    key = (device.type, device.index, slot)
    ws = _workspaces.get(key)
    if ws is None:
        # zero-filled: the first 4 KiB hold the in-launch hand-off counters (include/movae.h), which must start at zero
        ws = torch.zeros(WS_BYTES, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("mo-vae_amd ops run on an MI355X (cuda/HIP device) only; got a CPU tensor -- there is no CPU path")


def ptr(t):
    return 0 if t is None else t.data_ptr()
