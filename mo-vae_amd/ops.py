"""torch.autograd.Function wrappers over the C ABI (include/movae.h).

PyTorch supplies device memory, the stream and the autograd tape; every FLOP below runs in
libmovae_hip.so.  Activations are NHWC tensors of shape [N, H, W, C] (contiguous); conv weights
are the reference's parameter shapes ([Co,Ci,kh,kw] / [Ci,Co,kh,kw]) held in channels_last memory
so that `w.permute(0,2,3,1)` is the contiguous [.,kh,kw,.] image the kernels read.
"""
import ctypes as C
import os
import weakref

import torch
from torch.autograd import Function

from . import _lib as L


_call = L.call

#: data_ptr of a parameter -> 1-D destination tensor for its gradient.  autojac.mtl_backward points this at
#: the current row of the Jacobian arena so wgrad / BN-backward kernels write their results in place
#: (consumed on first use; a second use of the same parameter falls back to a fresh buffer).
GRAD_SINK = {}


#: data_ptr of every sink slice a kernel was handed for WRITING since the log was last cleared (SINK_LOG), and of every slice
#: handed out as an untouched all-zero gradient (SINK_ZERO_LOG): autojac.JacobianBuffer keeps its arena across steps without
#: re-zeroing it and uses the two logs to find slices that were written in an earlier step but not in this one.
SINK_LOG = []
SINK_ZERO_LOG = []


def _sink(param, shape):
    dst = GRAD_SINK.pop(param.data_ptr(), None) if GRAD_SINK else None
    if dst is not None and dst.numel() == param.numel():
        SINK_LOG.append(dst.data_ptr())
        return dst.view(shape)
    return torch.empty(shape, dtype=param.dtype, device=param.device)


#: data_ptr of a parameter -> its EXISTING gradient tensor: the weight-gradient kernels add into it (split-K reduce with
#: accumulate) instead of writing a fresh tensor that a torch add then folds in.  autojac.mtl_backward_begin sets this for the
#: task-side parameters that an earlier loss already reached (torchjd sums the per-loss gradients of those).
GRAD_ACCUM = {}


#: data_ptr of every parameter whose gradient a kernel accumulated in place during the current autograd call (filled by
#: _accum_targets, cleared together with GRAD_ACCUM): autojac._accumulate checks what autograd hands back against it.
GRAD_ACCUM_TAKEN = set()


def _accum_targets(w, b, need_b):
    """(dW destination in memory order, dbias destination) when BOTH of a layer's gradients can be accumulated in place."""
    if not GRAD_ACCUM:
        return None, None
    gw = GRAD_ACCUM.get(w.data_ptr())
    gb = GRAD_ACCUM.get(b.data_ptr()) if (b is not None and need_b) else None
    if gw is None or (b is not None and need_b and gb is None):
        return None, None
    v = gw.permute(0, 2, 3, 1) if gw.dim() == 4 else gw
    if not v.is_contiguous() or gw.shape != w.shape:
        return None, None
    GRAD_ACCUM.pop(w.data_ptr(), None)
    GRAD_ACCUM_TAKEN.add(w.data_ptr())
    if gb is not None:
        GRAD_ACCUM.pop(b.data_ptr(), None)
        GRAD_ACCUM_TAKEN.add(b.data_ptr())
    return v, gb


#: data_ptr of a FEATURE tensor (autojac.mtl_backward's `features`) -> destination for the gradient of the loss being differentiated
#: w.r.t. that feature: slice i of a stacked [K, ...] buffer.  The op that produces the feature's cotangent (reparameterize, the KL
#: term, ...) writes there, so the K cotangents of a feature are born stacked -- no torch.stack launch before the batched
#: pull-back.  Consumed on first use; an op that does not know about it simply returns its own buffer (and the stack copies).
COT_SINK = {}


def _cot(feature_ptr, like):
    """Destination for the gradient w.r.t. the tensor `like` (contiguous) whose memory the feature at `feature_ptr` views: the
    registered sink slice, seen with `like`'s shape (the feature may be a permuted view of the same memory -- logical NCHW over
    an NHWC buffer; the slice was allocated with the feature's strides, i.e. in that memory order), else a fresh buffer."""
    dst = COT_SINK.pop(feature_ptr, None) if COT_SINK else None
    if dst is not None and dst.numel() == like.numel() and dst.dtype == like.dtype and like.is_contiguous():
        if dst.shape == like.shape and dst.is_contiguous():
            return dst
        if sorted(zip(dst.stride(), dst.shape), reverse=True) == sorted(zip(like.stride(), like.shape), reverse=True):  # same memory order
            return dst.as_strided(like.shape, like.stride())
    return torch.empty_like(like)


#: batched pull-back (autojac._batched_pullback): one sink dict per cotangent group, same keys as GRAD_SINK
GRAD_SINK_ROWS = None


def _sink_row(g, param, shape, zeros=False):
    """Destination of group g's gradient of `param` in a batched backward (J row of that group when registered)."""
    rows = GRAD_SINK_ROWS
    dst = rows[g].pop(param.data_ptr(), None) if rows else None
    if dst is not None and dst.numel() == param.numel():
        (SINK_ZERO_LOG if zeros else SINK_LOG).append(dst.data_ptr())  # (zeros: the slice is handed out untouched)
        return dst.view(shape)
    return (torch.zeros if zeros else torch.empty)(shape, dtype=param.dtype, device=param.device)


def _stacked(t, G):
    """[G, ...] contiguous tensor from a stacked tensor or a list of G per-group tensors."""
    if isinstance(t, (list, tuple)):
        return torch.stack([_c(x) for x in t])
    return _c(t)


_ZERO_GRADS = {}


def _sink_zeros(param, shape):
    """An all-zero gradient: sinks are zero-initialised by their owner (autojac.JacobianBuffer) and written at most
    once, so a registered sink is returned untouched -- no fill launch.  Without a sink the parameter gets ONE persistent
    zero tensor for life (a bias in front of a training-mode BatchNorm: its gradient is identically zero, so whatever is
    accumulated into this tensor later is zero as well); nothing here launches a fill per step."""
    dst = GRAD_SINK.pop(param.data_ptr(), None) if GRAD_SINK else None
    if dst is not None and dst.numel() == param.numel():
        SINK_ZERO_LOG.append(dst.data_ptr())
        return dst.view(shape)
    key = (param.data_ptr(), param.numel(), param.dtype)
    z = _ZERO_GRADS.get(key)
    if z is None or z.device != param.device:
        z = _ZERO_GRADS[key] = torch.zeros(param.numel(), dtype=param.dtype, device=param.device)
    return z.view(shape)


def _ws(t):
    w = L.workspace(t.device)
    return w.data_ptr(), w.numel()


def _st(t):
    return L.stream_ptr(t.device)


def _c(t):
    """contiguous fp32 view/copy of an incoming gradient"""
    return t if t.is_contiguous() else t.contiguous()


def weight_mem(w):
    """[a, b, kh, kw] parameter -> contiguous [a, kh, kw, b] memory image (no copy when the
    parameter is channels_last, which is how nn.py creates it)."""
    v = w.permute(0, 2, 3, 1)
    if not v.is_contiguous():
        v = v.contiguous()
    return v


def conv_out_size(h, k, s, p):
    return (h + 2 * p - k) // s + 1


def convT_out_size(h, k, s, p, op):
    return (h - 1) * s - 2 * p + k + op


# ---------------------------------------------------------------------------------------------
class NchwToNhwc(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)  # an absent cotangent arrives as None: no zero-fill, no kernels on zeros
        L.require_gpu(x)
        x = _c(x)
        n, c, h, w = x.shape
        y = torch.empty((n, h, w, c), dtype=x.dtype, device=x.device)
        _call("movae_nchw_to_nhwc", x.data_ptr(), y.data_ptr(), n, c, h, w, _st(x))
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 1
        dy = _c(dy)
        n, h, w, c = dy.shape
        dx = torch.empty((n, c, h, w), dtype=dy.dtype, device=dy.device)
        _call("movae_nhwc_to_nchw", dy.data_ptr(), dx.data_ptr(), n, c, h, w, _st(dy))
        return dx

    @staticmethod
    def backward_batched(ctx, G, dy):
        dy = _stacked(dy, G)
        _, n, h, w, c = dy.shape
        dx = torch.empty((G, n, c, h, w), dtype=dy.dtype, device=dy.device)
        _call("movae_nhwc_to_nchw", dy.data_ptr(), dx.data_ptr(), G * n, c, h, w, _st(dy))
        return (dx,)


class NhwcToNchw(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)  # an absent cotangent arrives as None: no zero-fill, no kernels on zeros
        L.require_gpu(x)
        x = _c(x)
        n, h, w, c = x.shape
        y = torch.empty((n, c, h, w), dtype=x.dtype, device=x.device)
        _call("movae_nhwc_to_nchw", x.data_ptr(), y.data_ptr(), n, c, h, w, _st(x))
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 1
        dy = _c(dy)
        n, c, h, w = dy.shape
        dx = torch.empty((n, h, w, c), dtype=dy.dtype, device=dy.device)
        _call("movae_nchw_to_nhwc", dy.data_ptr(), dx.data_ptr(), n, c, h, w, _st(dy))
        return dx

    @staticmethod
    def backward_batched(ctx, G, dy):
        dy = _stacked(dy, G)
        _, n, c, h, w = dy.shape
        dx = torch.empty((G, n, h, w, c), dtype=dy.dtype, device=dy.device)
        _call("movae_nchw_to_nhwc", dy.data_ptr(), dx.data_ptr(), G * n, c, h, w, _st(dy))
        return (dx,)


_LAST_NHWC = [None]  # (weakref to the NCHW source, its version counter, data_ptr, capturing?, the NHWC copy)


def forget_nhwc():
    """Drop the remembered conversion (train.GraphedTrainStep calls this around every capture: a result produced eagerly
    must not stand in for a launch the graph has to contain, and graph-pool memory must not leak into eager code)."""
    _LAST_NHWC[0] = None



def to_nhwc(x):
    """logical NCHW tensor -> NHWC tensor (zero copy when it already is a permuted NHWC buffer).
    A step converts the same constant input batch twice -- for the encoder and again for the losses -- so the most recent
    conversion of a tensor that needs no gradient is remembered and handed out again while that tensor is unmodified."""
    v = x.permute(0, 2, 3, 1)
    if v.is_contiguous():
        return v
    if x.requires_grad or torch.is_grad_enabled() and x.grad_fn is not None:
        return NchwToNhwc.apply(x)
    last = _LAST_NHWC[0]
    cap = torch.cuda.is_current_stream_capturing()
    if last is not None and last[0]() is x and last[1] == x._version and last[2] == x.data_ptr() and last[3] == cap:
        return last[4]
    out = NchwToNhwc.apply(x)
    _LAST_NHWC[0] = (weakref.ref(x), x._version, x.data_ptr(), cap, out)
    return out


def flatten_nchw(x_nhwc):
    """nn.Flatten over the reference's NCHW ordering (models/vae.py:128)."""
    n, h, w, c = x_nhwc.shape
    if h == 1 and w == 1:
        return x_nhwc.reshape(n, c)
    return NhwcToNchw.apply(x_nhwc).reshape(n, c * h * w)


def unflatten_nchw(x, c, h, w):
    """nn.Unflatten(1, (C, H, W)) (models/vae.py:143) producing NHWC."""
    n = x.shape[0]
    if h == 1 and w == 1:
        return x.reshape(n, 1, 1, c)
    return NchwToNhwc.apply(x.reshape(n, c, h, w))


# ---------------------------------------------------------------------------------------------
class wgrad_side_stream:
    """Inside this block every convolution's weight / bias gradient is launched on a forked side stream (own scratch arena)
    and the compute stream does NOT wait for it: the backward's critical path is dgrad -> BatchNorm backward -> dgrad ...,
    the weight gradients are leaves nobody reads until the aggregation / optimizer.  Two short latency-bound launches
    then share the chip (measured on the C2 layers: 333 us for the 13 dgrad+wgrad pairs side by side vs 462 us in
    sequence).  The block's exit -- or join_wgrad() -- makes the compute stream wait for the side stream once.  Works
    under hipGraph capture (the side stream joins the capture at its first wait and returns at the join)."""

    def __init__(self, device, enabled=True):
        self.device, self.enabled = device, enabled and device.type == "cuda"
        self.keep, self.used, self.prev = [], False, None

    def __enter__(self):
        if self.enabled:
            self.prev, L.DEFER = L.DEFER, self
            self.side = L.side_stream(self.device)
        return self

    def join(self):
        if self.used:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
            self.keep.clear()  # operands the side stream was still reading (the allocator may recycle them now)
            self.used = False

    def __exit__(self, *exc):
        if self.enabled:
            self.join()
            L.DEFER = self.prev
        return False


def join_wgrad():
    """Make the compute stream wait for every deferred weight gradient (no-op outside wgrad_side_stream)."""
    if L.DEFER is not None:
        L.DEFER.join()


#: MOVAE_DEFER_REDUCE=0: every weight-gradient reduce is a launch of its own (A/B knob)
DEFER_REDUCE = os.environ.get("MOVAE_DEFER_REDUCE", "1") != "0"


class deferred_reduces:
    """Inside this block a convolution's weight-gradient split-K reduce whose destinations are gradient SINKS (a Jacobian row, an
    in-place accumulation target: autojac registers them, and reads them only through this library's aggregation / optimizer
    entry points) is not launched on the spot: the library parks it and the next implicit-GEMM launch of the backward carries it
    as extra blocks (include/movae.h: movae_reduce_defer).  The block's exit -- and every library call that
    is not one of the backward's own ops (_lib.DEFER_PASS) -- launches a reduce that is still parked; code that hands a weight
    gradient to anything else (torch arithmetic, a collective) inside the block calls flush_deferred() first."""

    def __init__(self, enabled=True):
        self.enabled = enabled and DEFER_REDUCE and L.DEFER is None
        self.prev = False

    def __enter__(self):
        if self.enabled:
            self.prev, L.DEFER_ON[0] = L.DEFER_ON[0], True
        return self

    def __exit__(self, *exc):
        if self.enabled:
            L.DEFER_ON[0] = self.prev
            if not self.prev:
                L.defer_flush()
        return False


def flush_deferred():
    L.defer_flush()


def _defer_ws(t, sunk):
    """(ws pointer, bytes) for a weight-gradient call: armed for a deferred reduce when every destination is a sink."""
    if sunk and L.DEFER_ON[0] and L.DEFER is None:
        return L.defer_arm(t.device)
    return None


#: data_ptr of the output of a conv + tanh / sigmoid pair that ends its nn.Stack -> that pair's ActLink: a reconstruction loss that is the
#: output's one reader looks the link up (out_act_link) and applies the derivative in its own backward kernel
OUT_ACT_LINKS = {}


def out_act_link(recons):
    link = OUT_ACT_LINKS.get(recons.data_ptr()) if FUSE_ACT else None
    if link is None or link.y is None or link.y.data_ptr() != recons.data_ptr() or link.y.shape != recons.shape:
        return None
    return link


class ActLink:
    """Shared by a conv whose epilogue applied an activation (the producer) and the ONE conv that consumes its output (nn.Stack
    hands it to the next module only, so the output has no other reader): the consumer's input-gradient pass multiplies its
    result by act'(y) in the epilogue / split-K reduce (movae_fuse_t::ep_act_*) and records the tensor it produced in `applied`;
    the producer's backward, handed that very tensor, starts from the pre-activation gradient -- no activation-backward pass."""
    __slots__ = ("y", "act", "slope", "applied", "res_in", "res_done")

    def __init__(self):
        self.y, self.act, self.slope, self.applied = None, None, 0.0, None
        #: ResCarrier of the residual block whose branch starts with this (stand-alone) activation: the consumer's epilogue adds
        #: the identity cotangent too (res_done: it did, for the tensor in `applied`)
        self.res_in, self.res_done = None, False


class ResCarrier:
    """out = branch(x) + x (ops.residual_add): in the backward, the gradient w.r.t. `out` reaches x twice -- through the branch and
    straight.  Instead of letting the autograd engine add the two, residual_add's backward leaves its cotangent HERE and returns
    nothing for x; the FIRST op of the branch (the one applied to x: a conv, or a stand-alone activation linked to a conv) adds it
    to the input gradient it produces -- in the input-gradient kernel's epilogue where that kernel can (movae_fuse_t::ep_res),
    with one explicit add otherwise.  The first op is told at forward time (nn.Stack: res_in)."""
    __slots__ = ("res", "armed")

    def __init__(self):
        self.res, self.armed = None, False


def _res_take(carrier):
    """The identity branch's cotangent waiting in `carrier` (and disarm it), or None."""
    if carrier is None or carrier.res is None:
        return None
    r, carrier.res = carrier.res, None
    return r


#: MOVAE_FUSE_ACT=0: every conv runs its own activation-backward pass (A/B knob)
FUSE_ACT = __import__("os").environ.get("MOVAE_FUSE_ACT", "1") != "0"


class ConvFusion:
    """What a conv call is asked to fuse of its neighbouring BatchNorms (include/movae.h: movae_fuse_t).
    in_*: the input is the RAW output y of a producer conv whose BatchNorm + activation this conv applies while loading,
    x = leaky_relu(in_scale[c] * y + in_shift[c], in_slope).  want_stats: the epilogue (or split-K reduce) also emits the
    per-channel partial sums of this conv's own output for the BatchNorm that follows it: `stats` / `parts` on return."""
    __slots__ = ("in_scale", "in_shift", "in_slope", "want_stats", "stats", "parts", "link", "act_in", "act_out", "res_in", "res_out",
                 "bn_fin", "fin")

    def __init__(self, in_scale=None, in_shift=None, in_slope=1.0, want_stats=False, link=None, act_in=None, act_out=None, res_in=None,
                 res_out=None):
        self.in_scale, self.in_shift, self.in_slope, self.want_stats = in_scale, in_shift, float(in_slope), want_stats
        self.stats, self.parts = None, 0
        #: (gamma, beta, running_mean, running_var, num_batches_tracked, eps, momentum) of the training-mode BatchNorm that follows:
        #: a producer that can (csrc/kgemm.h) finishes it inside its own launch; `fin` = [4, c] (mean, rstd, scale, shift) then
        self.bn_fin, self.fin = None, None
        #: ActLink of the producer whose activation output is this conv's input / ActLink this conv fills for its consumer
        self.act_in, self.act_out = act_in, act_out
        #: ResCarrier of the residual block whose branch starts with THIS conv (its input is the block's input)
        self.res_in = res_in
        #: (x, ResCarrier): this conv ENDS a residual branch -- its epilogue adds the block's input x, its backward hands the
        #: cotangent of the sum to the carrier (what ops.residual_add would do, without the launches)
        self.res_out = res_out
        #: dict shared with the BatchNormLazy node whose output this conv consumes: the conv's backward leaves that
        #: BatchNorm's backward sums here (emitted by the input-gradient epilogue), the BatchNorm's backward picks them up
        self.link = link


#: (entry point, geometry) for which the library answered "unsupported" to a fused input transform: not asked again
_NO_FUSE = set()
#: A/B knobs of the fusion's three parts (development): the normalise-on-load of consumers, the statistics from producer epilogues
FUSE_NORM = __import__("os").environ.get("MOVAE_FUSE_NORM", "1") != "0"
FUSE_STATS = __import__("os").environ.get("MOVAE_FUSE_STATS", "1") != "0"


#: MOVAE_BN_FIN=0: never ask a producer to finish the BatchNorm that follows (A/B knob; the kernels' own switch is MOVAE_KGEMM_BN_FIN)
BN_FIN = os.environ.get("MOVAE_BN_FIN", "1") != "0"


def _fuse_fin(f, bn_fin, out):
    """movae_fuse_t::fin_*: bn_fin = (gamma, beta, running_mean, running_var, nbt, eps, momentum), out = [4, c] result buffer"""
    gamma, beta, rm, rv, nbt, eps, mom = bn_fin
    f.fin_gamma, f.fin_beta, f.fin_eps, f.fin_momentum = gamma.data_ptr(), beta.data_ptr(), float(eps), float(mom)
    f.fin_out, f.fin_running_mean, f.fin_running_var, f.fin_nbt = out.data_ptr(), L.ptr(rm), L.ptr(rv), L.ptr(nbt)


def _fuse_struct(in_norm, stats=None, bn=None, ep=None):
    """bn = (y, scale, shift, slope, part): the dispatched input-gradient pass also emits the BatchNorm backward sums.
    ep = ActLink: its epilogue multiplies by the producer's activation derivative."""
    f = L.MovaeFuse()
    if ep is not None:
        link, res = ep[0], ep[1]
        if link is not None:
            f.ep_act_y, f.ep_act, f.ep_slope = link.y.data_ptr(), L.ACT[link.act], float(link.slope)
        if res is not None:
            f.ep_res = res.data_ptr()
    if in_norm is not None:
        f.in_scale, f.in_shift, f.in_slope = in_norm[0].data_ptr(), in_norm[1].data_ptr(), float(in_norm[2])
    if stats is not None:
        f.stats, f.stats_cap = stats.data_ptr(), stats.numel()
    if bn is not None:
        f.bn_y, f.bn_scale, f.bn_shift, f.bn_slope = bn[0].data_ptr(), bn[1].data_ptr(), bn[2].data_ptr(), float(bn[3])
        f.bn_part, f.bn_cap = bn[4].data_ptr(), bn[4].numel()
    return f


#: MOVAE_FUSE_BN_BWD=0: the BatchNorm backward runs its own reduction pass (the sums are not taken from the dgrad epilogue)
FUSE_BN_BWD = __import__("os").environ.get("MOVAE_FUSE_BN_BWD", "1") != "0"


def _bn_request(ctx, x, in_norm, G):
    """(y, scale, shift, slope, part) when this conv's input is the raw output of a fused BatchNorm whose backward is waiting
    for its sums (ctx.bn_link), else None.  Room: one pair per 8 rows, group and channel (the finest granularity in use)."""
    link = getattr(ctx, "bn_link", None)
    if not FUSE_BN_BWD or link is None or in_norm is None or not ctx.needs_input_grad[0]:
        return None
    c = x.shape[-1]
    rows = x.numel() // c
    part = torch.empty(G * (rows // 8 + 64) * 2 * c, dtype=torch.float32, device=x.device)
    return (x, in_norm[0], in_norm[1], in_norm[2], part)


def _act_request(ctx, x, bn, dx_shape=None):
    """What this conv's input-gradient epilogue is asked to do besides the plain dx: (link, res, carrier) or None.
    link: the ActLink whose activation derivative to apply; res: the identity cotangent of the residual block whose branch starts
    at this conv (carrier = ctx.res_in) or at the linked stand-alone activation (carrier = link.res_in)."""
    if bn is not None or not ctx.needs_input_grad[0]:
        return None
    link = getattr(ctx, "act_in", None)
    if (not FUSE_ACT or link is None or link.y is None or link.y.shape != x.shape or link.y.data_ptr() != x.data_ptr()):
        link = None
    carrier = link.res_in if link is not None else getattr(ctx, "res_in", None)
    res = carrier.res if (carrier is not None and carrier.res is not None) else None
    if res is not None and (not FUSE_ACT or res.data_ptr() % 16 != 0 or (dx_shape is not None and tuple(res.shape) != tuple(dx_shape))):
        res = None  # (left in the carrier: added explicitly by whoever owns it)
    if link is None and res is None:
        return None
    return (link, res, carrier)


def _act_publish(req, f, dx):
    if req is None or f is None or not int(f.ep_act_done):
        return
    link, res, carrier = req
    if res is not None:
        carrier.res = None  # consumed by the epilogue
    if link is not None:
        link.applied, link.res_done = dx.data_ptr(), res is not None


def _res_finish(ctx, dx):
    """A conv that is the first op of a residual branch: whatever is still waiting in its carrier is added to dx explicitly."""
    r = _res_take(getattr(ctx, "res_in", None))
    if r is None or dx is None:
        return dx
    r = _c(r).view_as(dx)
    _call("movae_add", dx.data_ptr(), r.data_ptr(), dx.data_ptr(), dx.numel(), _st(dx))
    return dx


def _act_take(ctx, dy):
    """True when dy already is the PRE-activation gradient (the consumer's epilogue applied act'(y) -- ActLink)."""
    link = getattr(ctx, "act_out", None)
    if link is None or link.applied is None:
        return False
    ap, link.applied = link.applied, None
    if ap != dy.data_ptr():
        raise RuntimeError("activation derivative was fused into the consumer's input gradient, but a different tensor reached the "
                           "producer's backward (its output has another reader?): set MOVAE_FUSE_ACT=0")
    return True


def _bn_publish(ctx, f, bn, dx, G):
    if bn is not None and f is not None and int(f.bn_ppg) > 0:
        ctx.bn_link["bwd"] = (dx.data_ptr(), bn[4], int(f.bn_ppg), G)


def scale_shift_act(y, scale, shift, slope):
    """leaky_relu(scale[c] * y + shift[c], slope) as a real tensor (no tape: callers wrap it)."""
    y = _c(y)
    c = y.shape[-1]
    out = torch.empty_like(y)
    _call("movae_scale_shift_act", y.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.data_ptr(), y.numel() // c, c, float(slope), _st(y))
    return out


class Conv(Function):
    """conv2d / conv_transpose2d / linear (+bias, + fused activation).  `fusion` (ConvFusion or None): BatchNorm fused into the
    conv -- the producer's normalisation + activation applied to the input while it is loaded, and / or the statistics of the
    output emitted for the BatchNorm that follows."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, out_pad, transposed, act, slope, bias_grad_is_zero=False, fusion=None):
        ctx.set_materialize_grads(False)  # an absent cotangent arrives as None: no zero-fill, no kernels on zeros
        L.require_gpu(x)
        x = _c(x)
        wm = weight_mem(w)
        n, hi, wi, ci = x.shape
        if transposed:
            assert wm.shape[0] == ci, "ConvTranspose2d weight/in_channels mismatch"
            kh, kw, co = wm.shape[1], wm.shape[2], wm.shape[3]
            ho, wo = convT_out_size(hi, kh, stride, pad, out_pad), convT_out_size(wi, kw, stride, pad, out_pad)
            fn = "movae_convT2d_fwd"
        else:
            assert wm.shape[3] == ci, "Conv2d weight/in_channels mismatch"
            co, kh, kw = wm.shape[0], wm.shape[1], wm.shape[2]
            ho, wo = conv_out_size(hi, kh, stride, pad), conv_out_size(wi, kw, stride, pad)
            fn = "movae_conv2d_fwd"
        y = torch.empty((n, ho, wo, co), dtype=x.dtype, device=x.device)
        wsp, wsb = _ws(x)
        geom = (n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad)
        in_norm = None
        if fusion is not None and fusion.in_scale is not None:
            in_norm = (fusion.in_scale, fusion.in_shift, fusion.in_slope)
        stats = None
        if fusion is not None and fusion.want_stats and not L.ACT[act] and FUSE_STATS:
            # room for the finest partial granularity of any producer (a pair per 32-row wave tile; per 8 rows in the reduce)
            stats = torch.empty((n * ho * wo // 8 + 64) * 2 * co, dtype=torch.float32, device=x.device)
        if in_norm is not None and ((fn, geom) in _NO_FUSE or not FUSE_NORM):
            x, in_norm = scale_shift_act(x, *in_norm), None
        res_out = fusion.res_out if (fusion is not None and fusion.res_out is not None and not L.ACT[act]) else None
        if res_out is not None:
            assert tuple(res_out[0].shape) == tuple(y.shape), "residual input and branch output differ in shape"
        ep_fwd = (None, _c(res_out[0])) if res_out is not None else None
        if in_norm is None and stats is None and ep_fwd is None:
            _call(fn, x.data_ptr(), wm.data_ptr(), L.ptr(b), y.data_ptr(), *geom, L.ACT[act], float(slope), wsp, wsb, _st(x))
        else:
            f = _fuse_struct(in_norm, stats, None, ep_fwd)
            fin_out = None
            if stats is not None and fusion.bn_fin is not None and BN_FIN:
                fin_out = torch.empty((4, co), dtype=torch.float32, device=x.device)
                _fuse_fin(f, fusion.bn_fin, fin_out)
            try:
                _call(fn + "_f", x.data_ptr(), wm.data_ptr(), L.ptr(b), y.data_ptr(), *geom, L.ACT[act], float(slope), wsp, wsb, _st(x),
                      C.byref(f))
            except L.Unsupported:  # nothing was launched: materialise the normalised input, then the same call without it
                if in_norm is None:
                    raise
                _NO_FUSE.add((fn, geom))
                x, in_norm = scale_shift_act(x, *in_norm), None
                f = _fuse_struct(None, stats, None, ep_fwd)
                if fin_out is not None:
                    _fuse_fin(f, fusion.bn_fin, fin_out)
                _call(fn + "_f", x.data_ptr(), wm.data_ptr(), L.ptr(b), y.data_ptr(), *geom, L.ACT[act], float(slope), wsp, wsb, _st(x),
                      C.byref(f))
            if fusion is not None:
                fusion.stats, fusion.parts = stats, int(f.stats_parts)
                fusion.fin = fin_out if (fin_out is not None and int(f.fin_done)) else None
            if ep_fwd is not None and not int(f.ep_act_done):  # this kernel has no such epilogue: one explicit add
                _call("movae_add", y.data_ptr(), ep_fwd[1].data_ptr(), y.data_ptr(), y.numel(), _st(x))
        ctx.geom = geom
        ctx.transposed, ctx.act, ctx.slope, ctx.has_bias = transposed, act, slope, b is not None
        ctx.bias_grad_is_zero = bias_grad_is_zero
        ctx.in_slope = in_norm[2] if in_norm is not None else None
        ctx.bn_link = fusion.link if (fusion is not None and in_norm is not None) else None
        # (with a transformed input: only the virtual stand-alone activation -- no BatchNorm link -- carries an ActLink too)
        ctx.act_in = fusion.act_in if (fusion is not None and (in_norm is None or fusion.link is None)) else None
        ctx.res_in = fusion.res_in if fusion is not None else None
        ctx.res_out = res_out[1] if res_out is not None else None
        ctx.act_out = None
        if fusion is not None and fusion.act_out is not None and L.ACT[act]:
            ctx.act_out = fusion.act_out
            ctx.act_out.y, ctx.act_out.act, ctx.act_out.slope, ctx.act_out.applied = y.detach(), act, slope, None
            if act in ("tanh", "sigmoid"):
                if len(OUT_ACT_LINKS) > 64:
                    OUT_ACT_LINKS.clear()
                OUT_ACT_LINKS[y.data_ptr()] = ctx.act_out
        ctx.save_for_backward(x, w, y if L.ACT[act] else None, b, *(in_norm[:2] if in_norm is not None else ()))
        return y

    @staticmethod
    def _saved(ctx):
        """(x, w, y, b, in_norm): in_norm = (scale, shift, slope) when x is the raw output of a BatchNorm-fused producer."""
        sv = ctx.saved_tensors
        in_norm = (sv[4], sv[5], ctx.in_slope) if ctx.in_slope is not None else None
        return sv[0], sv[1], sv[2], sv[3], in_norm

    @staticmethod
    def _wgrad_call(name, in_norm, geom, args, bn=None, ep=None):
        """One weight-gradient (or paired dgrad + wgrad) call; with a virtual activation operand the *_f form, falling back
        to a materialised activation where the dispatched kernel cannot apply the transform.  args: (head, x_index, tail).
        bn: BatchNorm-backward request for the paired call's dgrad (_bn_request).  Returns (x as used, fuse struct or None)."""
        head, xi, tail = args
        if in_norm is not None and (name, geom) in _NO_FUSE:
            head = list(head)
            head[xi] = scale_shift_act(head[xi], *in_norm)
            in_norm = None
        ptrs = lambda h: [t.data_ptr() if isinstance(t, torch.Tensor) else t for t in h]  # noqa: E731
        if in_norm is None and bn is None and ep is None:
            _call(name, *ptrs(head), *tail)
            return head[xi], None
        f = _fuse_struct(in_norm, None, bn, ep)
        try:
            _call(name + "_f", *ptrs(head), *tail, C.byref(f))
            return head[xi], f
        except L.Unsupported:
            if in_norm is None:
                raise
            _NO_FUSE.add((name, geom))
            head = list(head)
            head[xi] = scale_shift_act(head[xi], *in_norm)
            f = _fuse_struct(None, None, bn)
            _call(name + "_f", *ptrs(head), *tail, C.byref(f))
            return head[xi], f

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 11
        x, w, y, b, in_norm = Conv._saved(ctx)
        dy = _c(dy)
        if ctx.res_out is not None:  # this conv's output is branch + block input: dy is the identity branch's cotangent too
            ctx.res_out.res = dy
        n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad = ctx.geom
        st = _st(dy)
        wsp, wsb = _ws(dy)
        db_done = None
        if L.ACT[ctx.act] and not _act_take(ctx, dy):
            dpre = torch.empty_like(dy)
            if _fuse_bias_grad(ctx, dy, y, co):
                # activation backward and the bias gradient (column sums of dpre) in one pass over dy
                db_done = _sink(b, (co,))
                _call("movae_act_bwd_bias_grouped", 1, dy.data_ptr(), y.data_ptr(), dpre.data_ptr(), (C.c_void_p * 1)(db_done.data_ptr()),
                      dy.numel() // co, co, L.ACT[ctx.act], float(ctx.slope), 0, wsp, wsb, st)
            else:
                _call("movae_act_bwd", dy.data_ptr(), y.data_ptr(), dpre.data_ptr(), dy.numel(), L.ACT[ctx.act], float(ctx.slope), st)
            dy = dpre
        pre = "movae_convT2d_" if ctx.transposed else "movae_conv2d_"
        dx = dw = db = None
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        # dgrad and wgrad only share read-only operands: when both are needed the wgrad (+ its reduce / bias sum)
        # is issued on a forked side stream with its own scratch arena and joined afterwards, so the two short,
        # latency-bound launches overlap on the device (also inside a captured hipGraph, as parallel branches)
        defer = L.DEFER
        fork = need_w and (defer is not None or (L.SIDE_STREAM_WGRAD and ctx.needs_input_grad[0]))
        if fork:
            main, side = torch.cuda.current_stream(dy.device), L.side_stream(dy.device)
            side.wait_stream(main)
        pair = need_w and ctx.needs_input_grad[0] and not fork  # both gradients, one stream: one call, one main launch
        bn = _bn_request(ctx, x, in_norm, 1)
        ep = _act_request(ctx, x, bn, x.shape)
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            wm = weight_mem(w)
            if not pair:
                if bn is None and ep is None:
                    _call(pre + "dgrad", dy.data_ptr(), wm.data_ptr(), dx.data_ptr(), n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad,
                          wsp, wsb, st)
                else:
                    f = _fuse_struct(None, None, bn, ep)
                    _call(pre + "dgrad_f", dy.data_ptr(), wm.data_ptr(), dx.data_ptr(), n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad,
                          wsp, wsb, st, C.byref(f), 1)
                    _bn_publish(ctx, f, bn, dx, 1)
                    _act_publish(ep, f, dx)
        if fork:
            ws2 = L.workspace(dy.device, slot=1)
            wsp, wsb, st = ws2.data_ptr(), ws2.numel(), side.cuda_stream
        if need_w:
            wm_shape = (ci, kh, kw, co) if ctx.transposed else (co, kh, kw, ci)
            # a task-side parameter that an earlier loss already reached: add into its gradient inside the kernels' reduce
            plain_bias = ctx.has_bias and ctx.needs_input_grad[2] and db_done is None and not ctx.bias_grad_is_zero
            acc_w, acc_b = _accum_targets(w, b, plain_bias) if (db_done is None and not (ctx.has_bias and ctx.bias_grad_is_zero)) else (None, None)
            acc = 1 if acc_w is not None else 0
            nlog = len(SINK_LOG)
            dwm = acc_w.view(wm_shape) if acc else _sink(w, wm_shape)
            db_k = None
            if db_done is not None:
                db = db_done
            elif ctx.has_bias and ctx.needs_input_grad[2]:
                if ctx.bias_grad_is_zero:
                    # the bias feeds a training-mode BatchNorm, which subtracts the batch mean: d(loss)/d(bias) == 0
                    # identically (the reference's value is rounding noise of order 1e-9); no column-sum pass
                    db = _sink_zeros(b, (co,))
                else:
                    db = db_k = (acc_b if acc else _sink(b, (co,)))
            dbp = (C.c_void_p * 1)(db_k.data_ptr()) if db_k is not None else None
            # every destination a sink (or an in-place accumulation target): the split-K reduce may ride on a later launch
            armed = _defer_ws(dy, not fork and (acc or len(SINK_LOG) - nlog == 1 + (db_k is not None)))
            if armed is not None:
                wsp, wsb = armed
            tail = (n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, acc, wsp, wsb, st)
            if pair:
                x, f = Conv._wgrad_call(pre + "dgrad_wgrad_grouped", in_norm, ctx.geom,
                                        ((1, dy, wm, x, dx, (C.c_void_p * 1)(dwm.data_ptr()), dbp), 3, tail), bn, ep)
                _bn_publish(ctx, f, bn, dx, 1)
                _act_publish(ep, f, dx)
            else:
                x, _ = Conv._wgrad_call(pre + "wgrad_grouped", in_norm, ctx.geom, ((1, dy, x, (C.c_void_p * 1)(dwm.data_ptr()), dbp), 2, tail))
            dw = dwm.permute(0, 3, 1, 2)
        if fork and defer is not None:
            defer.keep.append((dy, x, dwm, db))  # joined once, by wgrad_side_stream
            defer.used = True
        elif fork:
            main.wait_stream(side)
        dx = _res_finish(ctx, dx)
        return dx, dw, db, None, None, None, None, None, None, None, None

    @staticmethod
    def backward_batched(ctx, G, dy):
        """The backward of G cotangents at once (autojac._batched_pullback): dy is [G, n, ho, wo, co] (or a list of
        G tensors).  dgrad runs as ONE launch over G*n images -- the pull-back is linear and per-sample, and the deep
        layers' grids are far too small to fill the chip one cotangent at a time; wgrad is one grouped launch (each
        group reduces over its own pixels, x is shared) writing straight into the groups' Jacobian rows."""
        x, w, y, b, in_norm = Conv._saved(ctx)
        dy = _stacked(dy, G)
        if ctx.res_out is not None:
            ctx.res_out.res = dy
        n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad = ctx.geom
        st = _st(dy)
        wsp, wsb = _ws(dy)
        db_done = None
        if L.ACT[ctx.act] and not _act_take(ctx, dy):
            dpre = torch.empty_like(dy)
            if _fuse_bias_grad(ctx, dy, y, co):
                db_done = [_sink_row(g, b, (co,)) for g in range(G)]
                _call("movae_act_bwd_bias_grouped", G, dy.data_ptr(), y.data_ptr(), dpre.data_ptr(),
                      (C.c_void_p * G)(*[t.data_ptr() for t in db_done]), y.numel() // co, co, L.ACT[ctx.act], float(ctx.slope), 0,
                      wsp, wsb, st)
            elif co % 4 == 0 and dy.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0:  # all groups in one launch, no bias sums
                _call("movae_act_bwd_bias_grouped", G, dy.data_ptr(), y.data_ptr(), dpre.data_ptr(), None, y.numel() // co, co,
                      L.ACT[ctx.act], float(ctx.slope), 0, wsp, wsb, st)
            else:
                for g in range(G):
                    _call("movae_act_bwd", dy[g].data_ptr(), y.data_ptr(), dpre[g].data_ptr(), y.numel(), L.ACT[ctx.act],
                          float(ctx.slope), st)
            dy = dpre
        pre = "movae_convT2d_" if ctx.transposed else "movae_conv2d_"
        dx = dw = db = None
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        need_w = ctx.needs_input_grad[1] or need_b
        defer = L.DEFER if need_w else None
        pair = need_w and ctx.needs_input_grad[0] and defer is None
        bn = _bn_request(ctx, x, in_norm, G)
        ep = _act_request(ctx, x, bn, (G,) + tuple(x.shape))
        if ctx.needs_input_grad[0]:
            dx = torch.empty((G,) + tuple(x.shape), dtype=x.dtype, device=x.device)
            wm = weight_mem(w)
            if not pair:
                if bn is None and ep is None:
                    _call(pre + "dgrad", dy.data_ptr(), wm.data_ptr(), dx.data_ptr(), G * n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad,
                          wsp, wsb, st)
                else:
                    f = _fuse_struct(None, None, bn, ep)
                    _call(pre + "dgrad_f", dy.data_ptr(), wm.data_ptr(), dx.data_ptr(), G * n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad,
                          wsp, wsb, st, C.byref(f), G)
                    _bn_publish(ctx, f, bn, dx, G)
                    _act_publish(ep, f, dx)
        if defer is not None:  # the grouped wgrad goes to the side stream (see wgrad_side_stream)
            side = L.side_stream(dy.device)
            side.wait_stream(torch.cuda.current_stream(dy.device))
            ws2 = L.workspace(dy.device, slot=1)
            wsp, wsb, st = ws2.data_ptr(), ws2.numel(), side.cuda_stream
        if need_w:
            wm_shape = (ci, kh, kw, co) if ctx.transposed else (co, kh, kw, ci)
            nlog = len(SINK_LOG)
            dwm = [_sink_row(g, w, wm_shape) for g in range(G)]
            arr = C.c_void_p * G
            if db_done is not None:
                db, dbp = db_done, None  # already produced by the fused activation-backward pass
            else:
                db = [_sink_row(g, b, (co,), zeros=ctx.bias_grad_is_zero) for g in range(G)] if need_b else None
                dbp = arr(*[t.data_ptr() for t in db]) if (need_b and not ctx.bias_grad_is_zero) else None
            # one grouped launch: blockIdx.z = group * splits + split, x is read by every group, dy by its own; with the
            # input gradient wanted too, dgrad and wgrad share the launch (igemm2_pair)
            # every destination a Jacobian row: the split-K reduce may ride on a later launch (deferred_reduces)
            armed = _defer_ws(dy, defer is None and len(SINK_LOG) - nlog == G * (1 + (dbp is not None)))
            if armed is not None:
                wsp, wsb = armed
            tail = (n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, 0, wsp, wsb, st)
            if pair:
                x, f = Conv._wgrad_call(pre + "dgrad_wgrad_grouped", in_norm, ctx.geom,
                                        ((G, dy, wm, x, dx, arr(*[t.data_ptr() for t in dwm]), dbp), 3, tail), bn, ep)
                _bn_publish(ctx, f, bn, dx, G)
                _act_publish(ep, f, dx)
            else:
                x, _ = Conv._wgrad_call(pre + "wgrad_grouped", in_norm, ctx.geom, ((G, dy, x, arr(*[t.data_ptr() for t in dwm]), dbp), 2, tail))
            dw = [t.permute(0, 3, 1, 2) for t in dwm]
            if defer is not None:
                defer.keep.append((dy, x, dwm, db))
                defer.used = True
        dx = _res_finish(ctx, dx)
        return dx, dw, db, None, None, None, None, None, None, None, None


def _fuse_bias_grad(ctx, dy, y, co):
    """The conv applied an activation in its epilogue and its bias needs a gradient: one fused pass (eltwise.hip) -- unless the
    weight-gradient kernel forms the column sums of dy itself while it stages dy (conv, 4-aligned channels: the implicit-GEMM
    wgrad; movae_conv2d_wgrad* falls back to the stand-alone column sum when its kernel cannot)."""
    ci = ctx.geom[3]
    if not ctx.transposed and ci % 4 == 0 and co % 4 == 0:
        return False
    return (ctx.has_bias and ctx.needs_input_grad[2] and not ctx.bias_grad_is_zero and co % 4 == 0 and
            dy.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0)


def conv2d(x, w, b=None, stride=1, pad=0, act=None, slope=0.01, bias_grad_is_zero=False, fusion=None):
    return Conv.apply(x, w, b, stride, pad, 0, False, act, slope, bias_grad_is_zero, fusion)


def conv_transpose2d(x, w, b=None, stride=1, pad=0, out_pad=0, act=None, slope=0.01, bias_grad_is_zero=False, fusion=None):
    return Conv.apply(x, w, b, stride, pad, out_pad, True, act, slope, bias_grad_is_zero, fusion)


def linear(x, w, b=None, act=None, slope=0.01):
    """x [B, in], w [out, in] -> [B, out]  (a 1x1 convolution on a 1x1 image)."""
    n, fin = x.shape
    y = Conv.apply(x.reshape(n, 1, 1, fin), w.view(w.shape[0], fin, 1, 1), b, 1, 0, 0, False, act, slope)
    return y.reshape(n, w.shape[0])


#: MOVAE_LINEAR_PAIR=0: fc_mu / fc_var as two ordinary linear calls
LINEAR_PAIR = __import__("os").environ.get("MOVAE_LINEAR_PAIR", "1") != "0"


def linear_pair_ok(x, w1, b1, w2, b2, groups=1):
    """The range of movae_linear_pair_* (include/movae.h): two equally shaped nn.Linear layers on one [m, k] input."""
    if not (LINEAR_PAIR and x.dim() == 2 and x.is_cuda and x.dtype == torch.float32 and w1.shape == w2.shape and b1 is not None and b2 is not None):
        return False
    m, k = x.shape
    n = w1.shape[0]
    if w1.shape[1] != k or n % 4 or k % 4 or max(m * n, m * k, n * k) > (1 << 20) or max(m, n, k) > 2048 or groups > 4:
        return False
    return all(t.is_contiguous() and t.data_ptr() % 16 == 0 for t in (x, w1, w2))


class LinearPair(Function):
    """(x w1^T + b1, x w2^T + b2) in one launch, and the two layers' backward in two (movae_linear_pair_*): fc_mu || fc_var of the
    VAE / BetaTC-VAE encoders (models/vae.py:187-192).  The caller checks linear_pair_ok."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        ctx.set_materialize_grads(False)
        L.require_gpu(x)
        x = _c(x)
        m, k = x.shape
        n = w1.shape[0]
        y1 = torch.empty((m, n), dtype=x.dtype, device=x.device)
        y2 = torch.empty_like(y1)
        _call("movae_linear_pair_fwd", x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), y1.data_ptr(), y2.data_ptr(),
              m, n, k, _st(x))
        ctx.save_for_backward(x, w1, b1, w2, b2)
        return y1, y2

    @staticmethod
    def _run(ctx, G, dy1, dy2, sink):
        """dy_i: [G, m, n] stacked (or None: that output had no cotangent).  sink(g, param, shape) -> gradient destination."""
        x, w1, b1, w2, b2 = ctx.saved_tensors
        m, k = x.shape
        n = w1.shape[0]
        ref = dy1 if dy1 is not None else dy2
        if dy1 is None:
            dy1 = torch.zeros_like(ref)
        if dy2 is None:
            dy2 = torch.zeros_like(ref)
        need_x = ctx.needs_input_grad[0]
        need_w = any(ctx.needs_input_grad[1:])
        dx = torch.empty((G, m, k), dtype=x.dtype, device=x.device) if need_x else None
        dw1 = dw2 = db1 = db2 = None
        if need_w:
            dw1, dw2 = [sink(g, w1, (n, k)) for g in range(G)], [sink(g, w2, (n, k)) for g in range(G)]
            db1, db2 = [sink(g, b1, (n,)) for g in range(G)], [sink(g, b2, (n,)) for g in range(G)]
        for g0 in range(0, G, 4):  # (the entry point takes up to four cotangent groups)
            g1 = min(G, g0 + 4)
            ptrs = lambda ts: (C.c_void_p * (g1 - g0))(*[t.data_ptr() for t in ts[g0:g1]]) if ts is not None else None  # noqa: E731
            _call("movae_linear_pair_bwd", g1 - g0, dy1[g0].data_ptr(), dy2[g0].data_ptr(), w1.data_ptr(), w2.data_ptr(), x.data_ptr(),
                  dx[g0].data_ptr() if dx is not None else 0, ptrs(dw1), ptrs(dw2), ptrs(db1), ptrs(db2), m, n, k, _st(x))
        return dx, dw1, db1, dw2, db2

    @staticmethod
    def backward(ctx, dy1, dy2):
        if dy1 is None and dy2 is None:
            return None, None, None, None, None
        one = lambda t: _c(t).unsqueeze(0) if t is not None else None  # noqa: E731
        dx, dw1, db1, dw2, db2 = LinearPair._run(ctx, 1, one(dy1), one(dy2), lambda g, p, shape: _sink(p, shape))
        first = lambda ts: ts[0] if ts is not None else None  # noqa: E731
        return (dx[0] if dx is not None else None), first(dw1), first(db1), first(dw2), first(db2)

    @staticmethod
    def backward_batched(ctx, G, dy1, dy2):
        if dy1 is None and dy2 is None:
            return None, None, None, None, None
        st = lambda t: _stacked(t, G) if t is not None else None  # noqa: E731
        return LinearPair._run(ctx, G, st(dy1), st(dy2), _sink_row)


def linear_pair(x, w1, b1, w2, b2):
    """(linear(x, w1, b1), linear(x, w2, b2)); one launch where the shapes allow (linear_pair_ok)."""
    if linear_pair_ok(x, w1, b1, w2, b2):
        return LinearPair.apply(x, w1, b1, w2, b2)
    return linear(x, w1, b1), linear(x, w2, b2)


# ---------------------------------------------------------------------------------------------
class BatchNormAct(Function):
    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, training, eps, momentum, act, slope, num_batches_tracked=None):
        ctx.set_materialize_grads(False)  # an absent cotangent arrives as None: no zero-fill, no kernels on zeros
        L.require_gpu(y)
        y = _c(y)
        c = y.shape[-1]
        rows = y.numel() // c
        out = torch.empty_like(y)
        mean = torch.empty(c, dtype=y.dtype, device=y.device)
        rstd = torch.empty(c, dtype=y.dtype, device=y.device)
        wsp, wsb = _ws(y)
        _call("movae_bn_act_fwd", y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), mean.data_ptr(),
              rstd.data_ptr(), L.ptr(running_mean), L.ptr(running_var), L.ptr(num_batches_tracked), rows, c, float(eps), float(momentum),
              1 if training else 0, L.ACT[act], float(slope), wsp, wsb, _st(y))
        ctx.act, ctx.slope, ctx.training = act, slope, training
        ctx.save_for_backward(y, gamma, beta, mean, rstd)
        ctx.mark_non_differentiable(mean, rstd)
        return out

    @staticmethod
    def backward(ctx, dout):
        if dout is None:
            return (None,) * 11
        y, gamma, beta, mean, rstd = ctx.saved_tensors
        if not ctx.training:
            raise RuntimeError("BatchNormAct backward is implemented for training-mode statistics only")
        dout = _c(dout)
        c = y.shape[-1]
        rows = y.numel() // c
        dy = torch.empty_like(y)
        dg = _sink(gamma, gamma.shape)
        db = _sink(beta, beta.shape)
        wsp, wsb = _ws(y)
        _call("movae_bn_act_bwd", dout.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
              rstd.data_ptr(), dy.data_ptr(), dg.data_ptr(), db.data_ptr(), rows, c, L.ACT[ctx.act], float(ctx.slope), 0,
              wsp, wsb, _st(y))
        return dy, dg, db, None, None, None, None, None, None, None, None

    @staticmethod
    def backward_batched(ctx, G, dout):
        y, gamma, beta, mean, rstd = ctx.saved_tensors
        if not ctx.training:
            raise RuntimeError("BatchNormAct backward is implemented for training-mode statistics only")
        dout = _stacked(dout, G)
        c = y.shape[-1]
        rows = y.numel() // c
        dy = torch.empty_like(dout)
        wsp, wsb = _ws(y)
        dgs = [_sink_row(g, gamma, gamma.shape) for g in range(G)]
        dbs = [_sink_row(g, beta, beta.shape) for g in range(G)]
        arr = C.c_void_p * G
        # one grouped call: the batch statistics of the cotangent are taken per group inside the kernels (blockIdx.y)
        _call("movae_bn_act_bwd_grouped", G, dout.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
              rstd.data_ptr(), dy.data_ptr(), arr(*[t.data_ptr() for t in dgs]), arr(*[t.data_ptr() for t in dbs]), rows, c,
              L.ACT[ctx.act], float(ctx.slope), 0, wsp, wsb, _st(y))
        return dy, dgs, dbs, None, None, None, None, None, None, None, None


# ---- BatchNorm fused into its neighbours ---------------------------------------------------------------
class LazyBN:
    """The output of a training-mode BatchNorm (+ LeakyReLU / ReLU) that has NOT been written to memory: `y` is the producer
    conv's raw output (as a node of the tape whose gradient is the gradient w.r.t. the normalised activation), scale / shift the
    folded per-channel map.  Convolutions consume it directly (ops.conv2d(..., fusion=...) applies the map while loading);
    anything else calls materialize().  Deliberately NOT a tensor: a consumer that does not know about it fails loudly."""
    __slots__ = ("y", "scale", "shift", "slope", "link")

    def __init__(self, y, scale, shift, slope, link=None):
        self.y, self.scale, self.shift, self.slope, self.link = y, scale, shift, float(slope), link

    def fusion(self, want_stats=False):
        return ConvFusion(self.scale, self.shift, self.slope, want_stats, self.link)

    def materialize(self):
        return ScaleShiftAct.apply(self.y, self.scale, self.shift, self.slope)


#: MOVAE_LAZY_ACT=1: a stand-alone ReLU / LeakyReLU in front of a conv is not written to memory -- the conv applies it while loading
#: (LazyAct).  OFF by default: measured SLOWER (C4 4.56 vs 4.44 ms, C3 10.61 vs 10.56): the operand transform in the consumer's
#: forward and weight-gradient kernels costs more than the twelve 6 us activation launches it removes.
LAZY_ACT = os.environ.get("MOVAE_LAZY_ACT", "0") == "1"
_UNIT_MAP = {}


def _unit_map(c, device):
    """(ones[c], zeros[c]): the identity scale / shift of a virtual activation, one pair per channel count and device for life"""
    key = (c, device.type, device.index)
    m = _UNIT_MAP.get(key)
    if m is None:
        m = _UNIT_MAP[key] = (torch.ones(c, dtype=torch.float32, device=device), torch.zeros(c, dtype=torch.float32, device=device))
    return m


class LazyAct(LazyBN):
    """A stand-alone ReLU / LeakyReLU whose output only the next conv reads, NOT written to memory: the conv applies it while loading
    (the fused-BatchNorm operand transform with scale 1, shift 0), its input-gradient epilogue applies the derivative (ActLink, read
    off the sign of the raw x).  `y` is the raw x as an Activation tape node (virtual=True)."""
    __slots__ = ("act_link",)

    def __init__(self, x, act, slope, link, res_in):
        sl = 0.0 if act == "relu" else float(slope)
        y = Activation.apply(x, act, sl, link, res_in, True)
        one, zero = _unit_map(x.shape[-1], x.device)
        super().__init__(y, one, zero, sl, None)
        self.act_link = link

    def fusion(self, want_stats=False):
        f = ConvFusion(self.scale, self.shift, self.slope, want_stats, None)
        f.act_in = self.act_link
        return f


def materialize(x):
    return x.materialize() if isinstance(x, LazyBN) else x


class ScaleShiftAct(Function):
    """LazyBN -> real tensor.  `y` stands for the normalised activation on the tape, so the backward is the identity."""

    @staticmethod
    def forward(ctx, y, scale, shift, slope):
        ctx.set_materialize_grads(False)  # an absent cotangent arrives as None: no zero-fill, no kernels on zeros
        return scale_shift_act(y, scale, shift, slope)

    @staticmethod
    def backward(ctx, dout):
        if dout is None:
            return (None,) * 4
        return dout, None, None, None

    @staticmethod
    def backward_batched(ctx, G, dout):
        return dout, None, None, None


_ACT_OF_SLOPE = {1.0: None, 0.0: "relu"}


class BatchNormLazy(Function):
    """Training-mode BatchNorm2d (+ LeakyReLU / ReLU) whose statistics come from the producer conv's partial sums
    (ConvFusion.stats) -- or from one stand-alone pass over y when that kernel could not emit them -- and whose output is
    not materialised: returns (y as a new tape node, scale, shift).  Backward: the ordinary BatchNorm backward kernels
    (movae_bn_act_bwd) on the saved raw y."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, num_batches_tracked, eps, momentum, slope, stats, parts, link=None, fin=None):
        L.require_gpu(y)
        c = y.shape[-1]
        rows = y.numel() // c
        st = _st(y)
        if fin is not None:  # the producer conv finished this BatchNorm inside its own launch (movae_fuse_t::fin_*)
            mean, rstd, scale, shift = fin[0], fin[1], fin[2], fin[3]
            ctx.act, ctx.slope = ("lrelu" if slope not in _ACT_OF_SLOPE else _ACT_OF_SLOPE[slope]), slope
            ctx.link = link
            ctx.save_for_backward(y, gamma, beta, mean, rstd, scale, shift)
            ctx.mark_non_differentiable(scale, shift)
            ctx.set_materialize_grads(False)
            return y.view_as(y), scale, shift
        if not parts:
            stats = torch.empty(1100 * 2 * c, dtype=torch.float32, device=y.device)  # movae_bn_stats: at most 1024 partials (+ room to fold them)
            pout = C.c_int(0)
            _call("movae_bn_stats", y.data_ptr(), rows, c, stats.data_ptr(), stats.numel(), C.byref(pout), st)
            parts = pout.value
        mean = torch.empty(c, dtype=y.dtype, device=y.device)
        rstd, scale, shift = torch.empty_like(mean), torch.empty_like(mean), torch.empty_like(mean)
        _call("movae_bn_finalize", stats.data_ptr(), stats.numel(), int(parts), rows, c, gamma.data_ptr(), beta.data_ptr(), float(eps), float(momentum),
              mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), L.ptr(running_mean), L.ptr(running_var),
              L.ptr(num_batches_tracked), st)
        ctx.act, ctx.slope = ("lrelu" if slope not in _ACT_OF_SLOPE else _ACT_OF_SLOPE[slope]), slope
        ctx.link = link
        ctx.save_for_backward(y, gamma, beta, mean, rstd, scale, shift)
        ctx.mark_non_differentiable(scale, shift)
        ctx.set_materialize_grads(False)  # no zero-fill launches for the two auxiliary outputs' (never used) gradients
        return y.view_as(y), scale, shift

    @staticmethod
    def _from_sums(ctx, G, dout, dgs, dbs):
        """The consumer conv's input-gradient pass left this BatchNorm's backward sums (ctx.link['bwd']): one tiny finalize
        launch and ONE pass that forms dy.  None when no (matching) sums are there."""
        ent = ctx.link.pop("bwd", None) if ctx.link is not None else None
        if ent is None or ent[0] != dout.data_ptr() or ent[3] != G:
            return None
        y, gamma, beta, mean, rstd, scale, shift = ctx.saved_tensors
        _, part, ppg, _ = ent
        c = y.shape[-1]
        rows = y.numel() // c
        if c % 4 != 0:
            return None
        coef = torch.empty((G, 3, c), dtype=y.dtype, device=y.device)
        arr = C.c_void_p * G
        st = _st(y)
        dy = torch.empty_like(dout)
        _call("movae_bn_bwd_finalize_apply", part.data_ptr(), part.numel(), ppg, G, rows, c, gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
              arr(*[t.data_ptr() for t in dgs]), arr(*[t.data_ptr() for t in dbs]), coef.data_ptr(), 0, dout.data_ptr(), y.data_ptr(),
              scale.data_ptr(), shift.data_ptr(), float(ctx.slope), dy.data_ptr(), st)
        return dy

    @staticmethod
    def backward(ctx, dout, _ds, _dh):
        if dout is None:
            return (None,) * 13
        y, gamma, beta, mean, rstd = ctx.saved_tensors[:5]
        dout = _c(dout)
        c = y.shape[-1]
        rows = y.numel() // c
        dg = _sink(gamma, gamma.shape)
        db = _sink(beta, beta.shape)
        dy = BatchNormLazy._from_sums(ctx, 1, dout, [dg], [db])
        if dy is not None:
            return dy, dg, db, None, None, None, None, None, None, None, None, None, None
        dy = torch.empty_like(y)
        wsp, wsb = _ws(y)
        _call("movae_bn_act_bwd", dout.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
              rstd.data_ptr(), dy.data_ptr(), dg.data_ptr(), db.data_ptr(), rows, c, L.ACT[ctx.act], float(ctx.slope), 0,
              wsp, wsb, _st(y))
        return dy, dg, db, None, None, None, None, None, None, None, None, None, None

    @staticmethod
    def backward_batched(ctx, G, dout, _ds=None, _dh=None):
        y, gamma, beta, mean, rstd = ctx.saved_tensors[:5]
        dout = _stacked(dout, G)
        c = y.shape[-1]
        rows = y.numel() // c
        dgs = [_sink_row(g, gamma, gamma.shape) for g in range(G)]
        dbs = [_sink_row(g, beta, beta.shape) for g in range(G)]
        dy = BatchNormLazy._from_sums(ctx, G, dout, dgs, dbs)
        if dy is not None:
            return dy, dgs, dbs, None, None, None, None, None, None, None, None, None, None
        dy = torch.empty_like(dout)
        wsp, wsb = _ws(y)
        arr = C.c_void_p * G
        _call("movae_bn_act_bwd_grouped", G, dout.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
              rstd.data_ptr(), dy.data_ptr(), arr(*[t.data_ptr() for t in dgs]), arr(*[t.data_ptr() for t in dbs]), rows, c,
              L.ACT[ctx.act], float(ctx.slope), 0, wsp, wsb, _st(y))
        return dy, dgs, dbs, None, None, None, None, None, None, None, None, None, None


def batch_norm_lazy(y, gamma, beta, running_mean, running_var, num_batches_tracked, eps, momentum, act, slope, fusion):
    """-> LazyBN.  act in (None, 'lrelu', 'relu'); fusion: the ConvFusion the producer conv was called with (its statistics)."""
    sl = 1.0 if act is None else (0.0 if act == "relu" else float(slope))
    stats, parts = (fusion.stats, fusion.parts) if fusion is not None else (None, 0)
    link = {}
    try:
        yv, scale, shift = BatchNormLazy.apply(y, gamma, beta, running_mean, running_var, num_batches_tracked, eps, momentum, sl, stats,
                                               parts, link, fusion.fin if fusion is not None else None)
    except L.Unsupported:  # no partial sums from the producer and a shape the stand-alone statistics pass does not take
        return batch_norm_act(y, gamma, beta, running_mean, running_var, True, eps, momentum, act, slope, num_batches_tracked)
    return LazyBN(yv, scale, shift, sl, link)


def batch_norm_act(y, gamma, beta, running_mean, running_var, training, eps=1e-5, momentum=0.1, act=None, slope=0.01,
                   num_batches_tracked=None):
    """num_batches_tracked (int64 device scalar) is incremented inside the statistics kernel in training mode."""
    return BatchNormAct.apply(y, gamma, beta, running_mean, running_var, training, eps, momentum, act, slope, num_batches_tracked)


# ---------------------------------------------------------------------------------------------
class Activation(Function):
    """A stand-alone activation.  act_out (ActLink or None): the output has exactly one reader, a conv whose input-gradient pass
    may apply this activation's derivative itself (nn.Stack arranges that); the backward then passes the gradient through."""

    @staticmethod
    def forward(ctx, x, act, slope, act_out=None, res_in=None, virtual=False):
        ctx.set_materialize_grads(False)  # an absent cotangent arrives as None: no zero-fill, no kernels on zeros
        L.require_gpu(x)
        x = _c(x)
        if virtual:
            # no kernel: the ONE consumer (a conv) applies the activation while it loads x (ops.LazyAct); `y` is x itself as a new
            # tape node, and stands in for the output wherever only its SIGN matters (ReLU / LeakyReLU derivative: act'(x) from x)
            assert act in ("relu", "lrelu") and act_out is not None
            y = x.view_as(x)
        else:
            y = torch.empty_like(x)
            _call("movae_act_fwd", x.data_ptr(), y.data_ptr(), x.numel(), L.ACT[act], float(slope), _st(x))
        ctx.act, ctx.slope = act, slope
        ctx.act_out, ctx.res_in = act_out, res_in  # res_in: this activation is the first op of a residual branch (ResCarrier)
        if act_out is not None:
            act_out.y, act_out.act, act_out.slope, act_out.applied = y.detach(), act, slope, None
            act_out.res_in, act_out.res_done = res_in, False
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 6
        (y,) = ctx.saved_tensors
        dy = _c(dy)
        if _act_take(ctx, dy):
            return Activation._add_res(ctx, dy, ctx.act_out.res_done), None, None, None, None, None
        dx = torch.empty_like(dy)
        _call("movae_act_bwd", dy.data_ptr(), y.data_ptr(), dx.data_ptr(), dy.numel(), L.ACT[ctx.act], float(ctx.slope), _st(dy))
        return Activation._add_res(ctx, dx, False), None, None, None, None, None

    @staticmethod
    def _add_res(ctx, dx, already):
        """dx (+)= the identity cotangent of the residual block this activation opens, unless the consumer's epilogue added it."""
        r = _res_take(ctx.res_in)
        if r is not None and not already:
            r = _c(r).view_as(dx)
            _call("movae_add", dx.data_ptr(), r.data_ptr(), dx.data_ptr(), dx.numel(), _st(dx))
        return dx

    @staticmethod
    def backward_batched(ctx, G, dy):
        (y,) = ctx.saved_tensors
        dy = _stacked(dy, G)
        if _act_take(ctx, dy):
            return Activation._add_res(ctx, dy, ctx.act_out.res_done), None, None, None, None, None
        dx = torch.empty_like(dy)
        c = y.shape[-1]
        if c % 4 == 0 and dy.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0:  # all groups in one launch
            wsp, wsb = _ws(dy)
            _call("movae_act_bwd_bias_grouped", G, dy.data_ptr(), y.data_ptr(), dx.data_ptr(), None, y.numel() // c, c, L.ACT[ctx.act],
                  float(ctx.slope), 0, wsp, wsb, _st(dy))
        else:
            for g in range(G):
                _call("movae_act_bwd", dy[g].data_ptr(), y.data_ptr(), dx[g].data_ptr(), y.numel(), L.ACT[ctx.act], float(ctx.slope),
                      _st(dy))
        return Activation._add_res(ctx, dx, False), None, None, None, None, None


def activation(x, act, slope=0.01, act_out=None, res_in=None):
    if not L.ACT[act]:
        return x
    return Activation.apply(x, act, slope, act_out, res_in)


class Add(Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.set_materialize_grads(False)  # an absent cotangent arrives as None: no zero-fill, no kernels on zeros
        L.require_gpu(a)
        a, b = _c(a), _c(b)
        y = torch.empty_like(a)
        _call("movae_add", a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), _st(a))
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 2
        return dy, dy

    @staticmethod
    def backward_batched(ctx, G, dy):
        return dy, dy


class ResidualAdd(Function):
    """out = branch + x with the identity cotangent handed to the branch's first op (ResCarrier) instead of to the engine."""

    @staticmethod
    def forward(ctx, branch, x, carrier):
        ctx.set_materialize_grads(False)
        L.require_gpu(x)
        branch, x = _c(branch), _c(x)
        y = torch.empty_like(x)
        _call("movae_add", branch.data_ptr(), x.data_ptr(), y.data_ptr(), x.numel(), _st(x))
        ctx.carrier = carrier
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 3
        if ctx.carrier is None or not ctx.needs_input_grad[1]:
            return dy, dy, None
        ctx.carrier.res = _c(dy)
        return dy, None, None

    @staticmethod
    def backward_batched(ctx, G, dy):
        if ctx.carrier is None or not ctx.needs_input_grad[1]:
            return dy, dy, None
        ctx.carrier.res = _stacked(dy, G)
        return dy, None, None


def residual_add(branch, x, carrier):
    """branch(x) + x; carrier: the ResCarrier the branch's first op was given (None: a plain add)."""
    assert branch.shape == x.shape
    return ResidualAdd.apply(branch, x, carrier)


def add(a, b):
    assert a.shape == b.shape
    return Add.apply(a, b)


class ConcatChannels(Function):
    """torch.cat([a, b], dim=channel) for NHWC tensors (models/vq_vae2.py:228,241)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.set_materialize_grads(False)  # an absent cotangent arrives as None: no zero-fill, no kernels on zeros
        L.require_gpu(a)
        a, b = _c(a), _c(b)
        ca, cb = a.shape[-1], b.shape[-1]
        rows = a.numel() // ca
        y = torch.empty(a.shape[:-1] + (ca + cb,), dtype=a.dtype, device=a.device)
        st = _st(a)
        _call("movae_copy_channels", a.data_ptr(), y.data_ptr(), rows, ca, ca + cb, 0, 0, ca, st)
        _call("movae_copy_channels", b.data_ptr(), y.data_ptr(), rows, cb, ca + cb, 0, ca, cb, st)
        ctx.ca, ctx.cb = ca, cb
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 2
        dy = _c(dy)
        ca, cb = ctx.ca, ctx.cb
        rows = dy.numel() // (ca + cb)
        da = torch.empty(dy.shape[:-1] + (ca,), dtype=dy.dtype, device=dy.device)
        db = torch.empty(dy.shape[:-1] + (cb,), dtype=dy.dtype, device=dy.device)
        st = _st(dy)
        _call("movae_copy_channels", dy.data_ptr(), da.data_ptr(), rows, ca + cb, ca, 0, 0, ca, st)
        _call("movae_copy_channels", dy.data_ptr(), db.data_ptr(), rows, ca + cb, cb, ca, 0, cb, st)
        return da, db


def concat_channels(a, b):
    return ConcatChannels.apply(a, b)


# ---------------------------------------------------------------------------------------------
class Reparameterize(Function):
    @staticmethod
    def forward(ctx, mu, log_var, eps):
        ctx.set_materialize_grads(False)  # an absent cotangent arrives as None: no zero-fill, no kernels on zeros
        L.require_gpu(mu)
        mu, log_var, eps = _c(mu), _c(log_var), _c(eps)
        z = torch.empty_like(mu)
        _call("movae_reparam_fwd", mu.data_ptr(), log_var.data_ptr(), eps.data_ptr(), z.data_ptr(), mu.numel(), _st(mu))
        ctx.save_for_backward(log_var, eps)
        ctx.in_ptrs = (mu.data_ptr(), log_var.data_ptr())  # (the cotangent sinks are keyed by the feature tensors)
        return z

    @staticmethod
    def backward(ctx, dz):
        if dz is None:
            return (None,) * 3
        log_var, eps = ctx.saved_tensors
        dz = _c(dz)
        dmu, dlv = _cot(ctx.in_ptrs[0], dz), _cot(ctx.in_ptrs[1], dz)
        _call("movae_reparam_bwd", dz.data_ptr(), log_var.data_ptr(), eps.data_ptr(), dmu.data_ptr(), dlv.data_ptr(), dz.numel(), _st(dz))
        return dmu, dlv, None


def reparameterize(mu, log_var, eps):
    return Reparameterize.apply(mu, log_var, eps)


class ReparameterizeRNG(Function):
    """z = mu + exp(0.5 * log_var) * eps with eps ~ N(0, 1) drawn inside the kernel (movae_reparam_rng_fwd): `state` is a device
    int64[2] = {seed, draws so far}, advanced by the launch itself -- one launch per step where torch.randn_like under graph capture
    plus the reparameterisation are four.  The backward is Reparameterize's (eps is kept)."""

    @staticmethod
    def forward(ctx, mu, log_var, state):
        ctx.set_materialize_grads(False)
        L.require_gpu(mu)
        mu, log_var = _c(mu), _c(log_var)
        assert state.dtype == torch.int64 and state.numel() == 2 and state.device == mu.device
        z, eps = torch.empty_like(mu), torch.empty_like(mu)
        _call("movae_reparam_rng_fwd", mu.data_ptr(), log_var.data_ptr(), eps.data_ptr(), z.data_ptr(), mu.numel(), state.data_ptr(), 1, _st(mu))
        ctx.save_for_backward(log_var, eps)
        ctx.in_ptrs = (mu.data_ptr(), log_var.data_ptr())
        return z

    @staticmethod
    def backward(ctx, dz):
        if dz is None:
            return (None,) * 3
        log_var, eps = ctx.saved_tensors
        dz = _c(dz)
        dmu, dlv = _cot(ctx.in_ptrs[0], dz), _cot(ctx.in_ptrs[1], dz)
        _call("movae_reparam_bwd", dz.data_ptr(), log_var.data_ptr(), eps.data_ptr(), dmu.data_ptr(), dlv.data_ptr(), dz.numel(), _st(dz))
        return dmu, dlv, None


def reparameterize_rng(mu, log_var, state):
    return ReparameterizeRNG.apply(mu, log_var, state)


# ---------------------------------------------------------------------------------------------
class ReconLoss(Function):
    """scale * mean(objective(recons, inputs)); both tensors must share one memory order."""

    @staticmethod
    def forward(ctx, recons, inputs, kind, scale, act_link=None):
        ctx.set_materialize_grads(False)  # an absent cotangent arrives as None: no zero-fill, no kernels on zeros
        L.require_gpu(recons)
        recons, inputs = _c(recons), _c(inputs)
        assert recons.shape == inputs.shape, (recons.shape, inputs.shape)
        out = torch.empty((), dtype=recons.dtype, device=recons.device)
        wsp, wsb = _ws(recons)
        _call("movae_recon_loss_fwd", recons.data_ptr(), inputs.data_ptr(), out.data_ptr(), recons.numel(), L.RECON[kind],
              float(scale), wsp, wsb, _st(recons))
        ctx.kind, ctx.scale = kind, scale
        # recons = act(pre), this loss its one reader (ActLink): the backward kernel hands the producing conv the PRE-activation gradient
        ctx.act_link = act_link if (FUSE_ACT and act_link is not None and act_link.act in ("tanh", "sigmoid") and act_link.y is not None
                                    and act_link.y.data_ptr() == recons.data_ptr()) else None
        ctx.save_for_backward(recons, inputs)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 5
        recons, inputs = ctx.saved_tensors
        g = _c(g)
        dr = torch.empty_like(recons)
        link = ctx.act_link
        if link is not None:
            _call("movae_recon_loss_bwd_act", recons.data_ptr(), inputs.data_ptr(), g.data_ptr(), dr.data_ptr(), recons.numel(),
                  L.RECON[ctx.kind], float(ctx.scale), L.ACT[link.act], float(link.slope), _st(recons))
            link.applied = dr.data_ptr()
        else:
            _call("movae_recon_loss_bwd", recons.data_ptr(), inputs.data_ptr(), g.data_ptr(), dr.data_ptr(), recons.numel(),
                  L.RECON[ctx.kind], float(ctx.scale), _st(recons))
        return dr, None, None, None, None


class EdgeWeightedPixelLoss(Function):
    """models/gg_vae.py:125-137: scale * mean(w * (recons - inputs)^2), w = Sobel magnitude of `inputs` (max over channels,
    normalised by the batch maximum).  NHWC operands; no gradient flows into `inputs`."""

    @staticmethod
    def forward(ctx, recons, inputs, scale):
        ctx.set_materialize_grads(False)  # an absent cotangent arrives as None: no zero-fill, no kernels on zeros
        L.require_gpu(recons)
        recons, inputs = _c(recons), _c(inputs)
        assert recons.shape == inputs.shape and recons.dim() == 4, (recons.shape, inputs.shape)
        n, h, w, c = recons.shape
        w_raw = torch.empty((n, h, w), dtype=recons.dtype, device=recons.device)
        wmax = torch.empty((), dtype=recons.dtype, device=recons.device)
        out = torch.empty((), dtype=recons.dtype, device=recons.device)
        wsp, wsb = _ws(recons)
        st = _st(recons)
        _call("movae_edge_weights", inputs.data_ptr(), w_raw.data_ptr(), wmax.data_ptr(), n, h, w, c, wsp, wsb, st)
        _call("movae_edge_weighted_mse_fwd", recons.data_ptr(), inputs.data_ptr(), w_raw.data_ptr(), wmax.data_ptr(), out.data_ptr(),
              n, h, w, c, float(scale), wsp, wsb, st)
        ctx.scale = scale
        ctx.save_for_backward(recons, inputs, w_raw, wmax)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 3
        recons, inputs, w_raw, wmax = ctx.saved_tensors
        g = _c(g)
        n, h, w, c = recons.shape
        dr = torch.empty_like(recons)
        _call("movae_edge_weighted_mse_bwd", recons.data_ptr(), inputs.data_ptr(), w_raw.data_ptr(), wmax.data_ptr(), g.data_ptr(),
              dr.data_ptr(), n, h, w, c, float(ctx.scale), _st(recons))
        return dr, None, None


class EdgeMatchingLoss(Function):
    """scale * mean f(sobel recons, sobel inputs) for the reference's edge-matching variants (`mode`, a key of
    _lib.EDGE_MATCH; include/movae.h lists each variant's reference lines).  NHWC; no gradient flows into `inputs`."""

    @staticmethod
    def forward(ctx, recons, inputs, scale, mode):
        ctx.set_materialize_grads(False)  # an absent cotangent arrives as None: no zero-fill, no kernels on zeros
        L.require_gpu(recons)
        recons, inputs = _c(recons), _c(inputs)
        assert recons.shape == inputs.shape and recons.dim() == 4, (recons.shape, inputs.shape)
        n, h, w, c = recons.shape
        out = torch.empty((), dtype=recons.dtype, device=recons.device)
        stats = torch.empty(8, dtype=recons.dtype, device=recons.device)
        wsp, wsb = _ws(recons)
        _call("movae_edge_match_fwd", recons.data_ptr(), inputs.data_ptr(), out.data_ptr(), n, h, w, c, float(scale),
              L.EDGE_MATCH[mode], stats.data_ptr(), wsp, wsb, _st(recons))
        ctx.scale, ctx.mode = scale, mode
        ctx.save_for_backward(recons, inputs, stats)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 4
        recons, inputs, stats = ctx.saved_tensors
        g = _c(g)
        n, h, w, c = recons.shape
        dr, ta, tb = torch.empty_like(recons), torch.empty_like(recons), torch.empty_like(recons)
        _call("movae_edge_match_bwd", recons.data_ptr(), inputs.data_ptr(), g.data_ptr(), dr.data_ptr(), ta.data_ptr(), tb.data_ptr(),
              n, h, w, c, float(ctx.scale), L.EDGE_MATCH[ctx.mode], stats.data_ptr(), _st(recons))
        return dr, None, None, None


def edge_weighted_pixel_loss(recons, inputs, scale=1.0):
    return EdgeWeightedPixelLoss.apply(recons, inputs, scale)


def edge_matching_loss(recons, inputs, scale=1.0, mode="mag"):
    return EdgeMatchingLoss.apply(recons, inputs, scale, mode)


def recon_loss(recons, inputs, kind, scale, act_link=None):
    return ReconLoss.apply(recons, inputs, kind, scale, act_link)


class KLDivergence(Function):
    @staticmethod
    def forward(ctx, mu, log_var, scale):
        ctx.set_materialize_grads(False)  # an absent cotangent arrives as None: no zero-fill, no kernels on zeros
        L.require_gpu(mu)
        mu, log_var = _c(mu), _c(log_var)
        b, d = mu.shape
        out = torch.empty((), dtype=mu.dtype, device=mu.device)
        wsp, wsb = _ws(mu)
        _call("movae_kl_fwd", mu.data_ptr(), log_var.data_ptr(), out.data_ptr(), b, d, float(scale), wsp, wsb, _st(mu))
        ctx.scale = scale
        ctx.save_for_backward(mu, log_var)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 3
        mu, log_var = ctx.saved_tensors
        g = _c(g)
        b, d = mu.shape
        dmu, dlv = _cot(mu.data_ptr(), mu), _cot(log_var.data_ptr(), mu)
        _call("movae_kl_bwd", mu.data_ptr(), log_var.data_ptr(), g.data_ptr(), dmu.data_ptr(), dlv.data_ptr(), b, d,
              float(ctx.scale), _st(mu))
        return dmu, dlv, None


def kl_divergence(mu, log_var, scale=1.0):
    return KLDivergence.apply(mu, log_var, scale)


class VAELosses(Function):
    """(reconstruction_loss, kld_loss, total_loss) of models/vae.py:211-228 in three launches instead of five (the two partial-sum
    passes and ONE final kernel that also adds the two): scale_r * mean(objective(recons, inputs)), scale_k * kl(mu, log_var) and
    their fp32 sum.  Bit-identical to ReconLoss + KLDivergence + a tensor add.  The backward serves whichever of the three
    cotangents arrive (mtl_backward differentiates the components one at a time, `--agg sum` the total)."""

    @staticmethod
    def forward(ctx, recons, inputs, kind, scale_r, mu, log_var, scale_k, act_link=None):
        ctx.set_materialize_grads(False)
        L.require_gpu(recons)
        recons, inputs, mu, log_var = _c(recons), _c(inputs), _c(mu), _c(log_var)
        assert recons.shape == inputs.shape, (recons.shape, inputs.shape)
        b, d = mu.shape
        out = torch.empty(3, dtype=recons.dtype, device=recons.device)
        wsp, wsb = _ws(recons)
        _call("movae_vae_losses_fwd", recons.data_ptr(), inputs.data_ptr(), recons.numel(), L.RECON[kind], float(scale_r), mu.data_ptr(),
              log_var.data_ptr(), b, d, float(scale_k), out.data_ptr(), wsp, wsb, _st(recons))
        ctx.kind, ctx.scale_r, ctx.scale_k = kind, scale_r, scale_k
        # recons = act(pre) is the decoder's output activation and this loss its one reader (ActLink): the backward kernel applies
        # act'(pre) and hands the producing conv the PRE-activation gradient
        ctx.act_link = act_link if (FUSE_ACT and act_link is not None and act_link.act in ("tanh", "sigmoid")) else None
        ctx.save_for_backward(recons, inputs, mu, log_var)
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g_rec, g_kld, g_tot):
        recons, inputs, mu, log_var = ctx.saved_tensors
        both = lambda a, b: a if b is None else (b if a is None else a + b)  # noqa: E731  (total_loss feeds both terms)
        gr, gk = both(g_rec, g_tot), both(g_kld, g_tot)
        dr = dmu = dlv = None
        if gr is not None and ctx.needs_input_grad[0]:
            gr = _c(gr)
            dr = torch.empty_like(recons)
            link = ctx.act_link
            if link is not None:
                _call("movae_recon_loss_bwd_act", recons.data_ptr(), inputs.data_ptr(), gr.data_ptr(), dr.data_ptr(), recons.numel(),
                      L.RECON[ctx.kind], float(ctx.scale_r), L.ACT[link.act], float(link.slope), _st(recons))
                link.applied = dr.data_ptr()
            else:
                _call("movae_recon_loss_bwd", recons.data_ptr(), inputs.data_ptr(), gr.data_ptr(), dr.data_ptr(), recons.numel(),
                      L.RECON[ctx.kind], float(ctx.scale_r), _st(recons))
        if gk is not None and (ctx.needs_input_grad[4] or ctx.needs_input_grad[5]):
            gk = _c(gk)
            b, d = mu.shape
            dmu, dlv = _cot(mu.data_ptr(), mu), _cot(log_var.data_ptr(), mu)
            _call("movae_kl_bwd", mu.data_ptr(), log_var.data_ptr(), gk.data_ptr(), dmu.data_ptr(), dlv.data_ptr(), b, d,
                  float(ctx.scale_k), _st(mu))
        return dr, None, None, None, dmu, dlv, None, None


def vae_losses(recons, inputs, kind, scale_r, mu, log_var, scale_k, act_link=None):
    return VAELosses.apply(recons, inputs, kind, scale_r, mu, log_var, scale_k, act_link)


class CombineLosses(Function):
    """The scalar arithmetic of a loss_function (models/vq_vae.py:381-391, vq_vae2.py:313-334, betatc_vae.py:298-324) in one
    launch forward and one backward: `inputs` are device tensors of 1..n fp32 scalars each (T scalars in all, in order), `coef`
    a K x T list of rows, out[k] = f_k * w_k * (sum of the row's terms) and out[K] = out[0] + out[1] + ...  Returns the K + 1
    scalars.  `anneal`: None or (row, iter_dev, steps, training) -- BetaTC's annealing counter on the device.  The backward
    returns a cotangent only for the inputs a present cotangent reaches (the others stay None, so a per-loss backward of
    mtl_backward does not walk the other losses' subgraphs)."""

    @staticmethod
    def forward(ctx, coef, anneal, *inputs):
        ctx.set_materialize_grads(False)
        L.require_gpu(inputs[0])
        inputs = [_c(t) for t in inputs]
        sizes = [t.numel() for t in inputs]
        T, K = sum(sizes), len(coef)
        assert all(len(r) == T for r in coef) and all(t.dtype == torch.float32 for t in inputs), (T, [len(r) for r in coef])
        ptrs = (C.c_void_p * T)(*[t.data_ptr() + 4 * j for t, n in zip(inputs, sizes) for j in range(n)])
        flat = (C.c_float * (K * T))(*[float(c) for r in coef for c in r])
        out = torch.empty(K + 1, dtype=torch.float32, device=inputs[0].device)
        row, it_dev, steps, training = anneal if anneal is not None else (-1, None, 1.0, False)
        fac = torch.empty(1, dtype=torch.float32, device=out.device) if it_dev is not None else None
        # (the host arrays are passed as ctypes objects, not addresses: a recorded call -- bench.py's replay -- keeps them alive)
        _call("movae_combine_losses_fwd", T, ptrs, K, flat, it_dev.data_ptr() if it_dev is not None else 0,
              float(steps), int(row), int(bool(training)), out.data_ptr(), fac.data_ptr() if fac is not None else 0, _st(out))
        ctx.coef, ctx.sizes, ctx.row, ctx.fac, ctx.flat = coef, sizes, int(row), fac, flat
        ctx.shapes = [t.shape for t in inputs]
        return tuple(out[k] for k in range(K + 1))

    @staticmethod
    def backward(ctx, *g):
        K, T = len(ctx.coef), sum(ctx.sizes)
        if all(x is None for x in g):
            return (None,) * (2 + len(ctx.sizes))
        g = [None if x is None else _c(x) for x in g]
        dev = next(x for x in g if x is not None).device
        gp = (C.c_void_p * (K + 1))(*[0 if x is None else x.data_ptr() for x in g])
        gterms = torch.empty(T, dtype=torch.float32, device=dev)
        _call("movae_combine_losses_bwd", T, K, gp, ctx.flat,
              ctx.fac.data_ptr() if ctx.fac is not None else 0, ctx.row, gterms.data_ptr(), _st(gterms))
        reached = [any(ctx.coef[k][t] != 0 and (g[k] is not None or g[K] is not None) for k in range(K)) for t in range(T)]
        grads, off = [], 0
        for i, n in enumerate(ctx.sizes):
            if ctx.needs_input_grad[2 + i] and any(reached[off: off + n]):
                grads.append(gterms[off: off + n].view(ctx.shapes[i]))
            else:
                grads.append(None)
            off += n
        return (None, None, *grads)


def combine_losses(inputs, coef, anneal=None):
    return CombineLosses.apply(coef, anneal, *inputs)


class TCDecomposition(Function):
    """-> tensor [3] = (mi, tc, kld) of models/betatc_vae.py:294-296 (unweighted)."""

    @staticmethod
    def forward(ctx, z, mu, log_var, log_iw):
        ctx.set_materialize_grads(False)  # an absent cotangent arrives as None: no zero-fill, no kernels on zeros
        L.require_gpu(z)
        z, mu, log_var, log_iw = _c(z), _c(mu), _c(log_var), _c(log_iw)
        b, d = z.shape
        out = torch.empty(3, dtype=z.dtype, device=z.device)
        lj = torch.empty(b + b * b, dtype=z.dtype, device=z.device)
        lm = torch.empty(b, d, dtype=z.dtype, device=z.device)
        wsp, wsb = _ws(z)
        _call("movae_tc_decomp_fwd", z.data_ptr(), mu.data_ptr(), log_var.data_ptr(), log_iw.data_ptr(), out.data_ptr(),
              lj.data_ptr(), lm.data_ptr(), b, d, wsp, wsb, _st(z))
        ctx.save_for_backward(z, mu, log_var, log_iw, lj, lm)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 4
        z, mu, log_var, log_iw, lj, lm = ctx.saved_tensors
        g = _c(g)
        b, d = z.shape
        dz, dmu, dlv = torch.empty_like(z), torch.empty_like(z), torch.empty_like(z)
        _call("movae_tc_decomp_bwd", z.data_ptr(), mu.data_ptr(), log_var.data_ptr(), log_iw.data_ptr(), lj.data_ptr(),
              lm.data_ptr(), g.data_ptr(), dz.data_ptr(), dmu.data_ptr(), dlv.data_ptr(), b, d, _st(z))
        return dz, dmu, dlv, None


def tc_decomposition(z, mu, log_var, log_iw):
    return TCDecomposition.apply(z, mu, log_var, log_iw)


# ---------------------------------------------------------------------------------------------
class VectorQuantize(Function):
    """x [.., D] latents (NHWC), codebook [K, D] -> (q_st, commitment, embedding, idx, used_count).

    q_st carries the straight-through gradient to x; commitment = mse(q.detach(), x),
    embedding = mse(q, x.detach()) (models/vq_vae.py:51-55)."""

    @staticmethod
    def forward(ctx, x, codebook):
        L.require_gpu(x)
        x, e = _c(x), _c(codebook)
        k, d = e.shape
        rows = x.numel() // d
        q = torch.empty_like(x)
        idx = torch.empty(rows, dtype=torch.int64, device=x.device)
        sse = torch.empty((), dtype=x.dtype, device=x.device)
        used = torch.empty((), dtype=torch.int32, device=x.device)
        mse2 = torch.empty(2, dtype=x.dtype, device=x.device)
        wsp, wsb = _ws(x)
        # commitment = mse(q.detach(), x) and embedding = mse(q, x.detach()): one value, written twice by the finalize kernel
        _call("movae_vq_nearest_fwd_mse", x.data_ptr(), e.data_ptr(), q.data_ptr(), idx.data_ptr(), sse.data_ptr(), used.data_ptr(),
              mse2.data_ptr(), rows, k, d, wsp, wsb, _st(x))
        commitment, embedding = mse2[0], mse2[1]
        ctx.save_for_backward(x, q, idx)
        ctx.x_ptr = x.data_ptr()
        ctx.kd = (k, d)
        ctx.mark_non_differentiable(idx, used)
        ctx.set_materialize_grads(False)  # a loss that does not reach an output hands None (not zeros): that part is skipped
        return q, commitment, embedding, idx, used

    @staticmethod
    def backward(ctx, dq, gc, ge, _gi, _gu):
        x, q, idx = ctx.saved_tensors
        k, d = ctx.kd
        rows = x.numel() // d
        dq = _c(dq) if dq is not None else None
        need_x, need_e = ctx.needs_input_grad
        # a cotangent that reaches neither the straight-through output nor the commitment term (the embedding loss alone) has NO
        # gradient w.r.t. x: None, not a tensor of zeros -- mtl_backward then skips that loss's pull-back through the encoder
        # (a zero Jacobian row costs nothing); likewise no codebook gradient without an embedding-loss cotangent
        dx = _cot(ctx.x_ptr, x) if (need_x and (dq is not None or gc is not None)) else None  # (x is a feature: born stacked)
        de = torch.empty((k, d), dtype=x.dtype, device=x.device) if (need_e and ge is not None) else None
        if dx is None and de is None:
            return None, None
        wsp, wsb = _ws(x)
        _call("movae_vq_bwd", x.data_ptr(), q.data_ptr(), idx.data_ptr(), L.ptr(dq), L.ptr(_c(gc) if gc is not None else None),
              L.ptr(_c(ge) if ge is not None else None), L.ptr(dx), L.ptr(de), rows, k, d, wsp, wsb, _st(x))
        return dx, de


def vector_quantize(x, codebook):
    return VectorQuantize.apply(x, codebook)


# ---------------------------------------------------------------------------------------------
# PixelCNN prior (models/pixelcnn_prior.py)
class EmbeddingLookup(Function):
    """nn.Embedding forward over a [B, H, W] int64 code grid -> NHWC activations [B, H, W, D] (pixelcnn_prior.py:306,371); the
    weight gradient is a fixed-order segmented sum over the sorted codes (no float atomics)."""

    @staticmethod
    def forward(ctx, idx, weight):
        ctx.set_materialize_grads(False)  # an absent cotangent arrives as None: no zero-fill, no kernels on zeros
        L.require_gpu(weight)
        idx = idx.contiguous()
        w = _c(weight)
        k, d = w.shape
        y = torch.empty(tuple(idx.shape) + (d,), dtype=w.dtype, device=w.device)
        _call("movae_embedding_fwd", w.data_ptr(), idx.data_ptr(), y.data_ptr(), idx.numel(), k, d, _st(w))
        ctx.save_for_backward(idx)
        ctx.kd = (k, d)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 2
        (idx,) = ctx.saved_tensors
        k, d = ctx.kd
        dy = _c(dy)
        dw = torch.empty((k, d), dtype=dy.dtype, device=dy.device)
        wsp, wsb = _ws(dy)
        _call("movae_embedding_bwd", dy.data_ptr(), idx.data_ptr(), dw.data_ptr(), idx.numel(), k, d, wsp, wsb, _st(dy))
        return None, dw


def embedding(idx, weight):
    return EmbeddingLookup.apply(idx, weight)


class GatedResidual(Function):
    """out = res + gate * feat (gate / feat already activated by the conv epilogues; pixelcnn_prior.py:85-90)."""

    @staticmethod
    def forward(ctx, res, gate, feat):
        ctx.set_materialize_grads(False)  # an absent cotangent arrives as None: no zero-fill, no kernels on zeros
        L.require_gpu(res)
        res, gate, feat = _c(res), _c(gate), _c(feat)
        out = torch.empty_like(res)
        _call("movae_gated_residual_fwd", res.data_ptr(), gate.data_ptr(), feat.data_ptr(), out.data_ptr(), res.numel(), _st(res))
        ctx.save_for_backward(gate, feat)
        return out

    @staticmethod
    def backward(ctx, dout):
        if dout is None:
            return (None,) * 3
        gate, feat = ctx.saved_tensors
        dout = _c(dout)
        dg, df = torch.empty_like(dout), torch.empty_like(dout)
        _call("movae_gated_residual_bwd", dout.data_ptr(), gate.data_ptr(), feat.data_ptr(), dg.data_ptr(), df.data_ptr(), dout.numel(),
              _st(dout))
        return dout, dg, df


def gated_residual(res, gate, feat):
    assert res.shape == gate.shape == feat.shape and res.numel() % 4 == 0
    return GatedResidual.apply(res, gate, feat)


@torch.no_grad()
def mask_weight_(weight, mask):
    """MaskedConv2d's `self.weight.data *= self.mask` (pixelcnn_prior.py:52): in place, outside the tape.  Both tensors share one
    dense memory layout (channels_last), so the flat product is the element-wise one."""
    L.require_gpu(weight)
    assert weight.shape == mask.shape and weight.stride() == mask.stride(), "mask must share the weight's memory layout"
    _call("movae_mul", weight.data_ptr(), mask.data_ptr(), weight.data_ptr(), weight.numel(), _st(weight))
    return weight


class CrossEntropy(Function):
    """F.cross_entropy(logits[rows, K], target[rows]) with mean reduction (main.py:1003-1006; pixelcnn_prior.py:392-395)."""

    @staticmethod
    def forward(ctx, logits, target):
        ctx.set_materialize_grads(False)  # an absent cotangent arrives as None: no zero-fill, no kernels on zeros
        L.require_gpu(logits)
        logits, target = _c(logits), target.contiguous()
        rows, k = logits.shape
        assert target.numel() == rows and target.dtype == torch.int64
        loss = torch.empty((), dtype=logits.dtype, device=logits.device)
        lse = torch.empty(rows, dtype=logits.dtype, device=logits.device)
        wsp, wsb = _ws(logits)
        _call("movae_cross_entropy_fwd", logits.data_ptr(), target.data_ptr(), loss.data_ptr(), lse.data_ptr(), rows, k, wsp, wsb,
              _st(logits))
        ctx.save_for_backward(logits, target, lse)
        return loss

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 2
        logits, target, lse = ctx.saved_tensors
        g = _c(g)
        rows, k = logits.shape
        dl = torch.empty_like(logits)
        _call("movae_cross_entropy_bwd", logits.data_ptr(), target.data_ptr(), lse.data_ptr(), g.data_ptr(), dl.data_ptr(), rows, k,
              _st(logits))
        return dl, None


def cross_entropy(logits, target):
    return CrossEntropy.apply(logits, target)
