"""Optimizer tail of the step: torch.optim.Adam / AdamW as the reference builds them
(main.py:1169-1178, stepped at main.py:214) with the per-tensor update fused into ONE launch.

torch's own paths cost ~15 foreach launches per step (eager) or ~110 launches (capturable=True, which falls
back to one elementwise division per parameter); on the 32x32 configurations that is a quarter of the whole
step.  `FusedAdam` keeps torch.optim.Adam's constructor, param_groups, state keys (`step`, `exp_avg`,
`exp_avg_sq`) and state_dict format -- checkpoints are interchangeable -- and replaces only `step()` by
`movae_adam_multi` (include/movae.h).  There is no CPU path.
"""
import ctypes as C

import torch

from . import _lib as L


def _dense(t):
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


def _layout_differs(a, p):
    """Do a and p (same shape) lay their elements out differently?  Strides of size-1 dimensions carry no layout: a 1x1
    convolution's weight [o, i, 1, 1] is the same memory contiguous (strides i, 1, 1, 1) and channels_last (i, 1, i, i)."""
    return a.shape != p.shape or any(n > 1 and x != y for n, x, y in zip(p.shape, a.stride(), p.stride()))


class FusedAdam(torch.optim.Adam):
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) with a one-launch step.

    device_step=True keeps {step, lr} in a device float[2] that the kernel reads and increments, so a captured
    hipGraph of `step()` stays live across replays (the role of torch's capturable=True); call `sync_hyper()`
    outside the capture after changing `param_groups[i]['lr']` (train.GraphedTrainStep does)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled_weight_decay=False,
                 device_step=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, foreach=False, capturable=False)
        self._decoupled = bool(decoupled_weight_decay)
        self._device_step = bool(device_step)
        self._hyper = {}      # group index -> device float[2] {step, lr}
        self._hyper_lr = {}   # the lr value last written to the device

    # -- device-resident hyper-parameters ---------------------------------------------------------
    def _group_hyper(self, gi, group, device):
        h = self._hyper.get(gi)
        if h is None:
            h = torch.tensor([0.0, float(group["lr"])], dtype=torch.float32, device=device)
            self._hyper[gi] = h
            self._hyper_lr[gi] = float(group["lr"])
        return h

    def sync_hyper(self):
        """Push host-side learning-rate changes (lr schedulers) to the device copies; never called under capture."""
        for gi, group in enumerate(self.param_groups):
            h = self._hyper.get(gi)
            if h is not None and self._hyper_lr[gi] != float(group["lr"]):
                h[1].fill_(float(group["lr"]))
                self._hyper_lr[gi] = float(group["lr"])

    # -- the step ----------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            if group.get("amsgrad") or group.get("maximize"):
                raise NotImplementedError("FusedAdam: amsgrad / maximize are not part of the reference's configuration")
            beta1, beta2 = group["betas"]
            buckets = {}  # step value -> rows; torch keeps one counter per parameter, so parameters whose gradient was
            #               None on some steps are updated with their own bias correction
            hyper = None
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                L.require_gpu(p)
                if g.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients")
                if p.dtype != torch.float32 or not _dense(p):
                    raise NotImplementedError("FusedAdam: parameters must be dense fp32 tensors")
                if g.dtype != p.dtype or _layout_differs(g, p):
                    g = torch.empty_like(p).copy_(g)
                st = self.state[p]
                if len(st) == 0:
                    if self._device_step:
                        hyper = self._group_hyper(gi, group, p.device)
                        st["step"] = hyper[0]
                    else:
                        st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if _layout_differs(m, p) or _layout_differs(v, p):  # state loaded from a differently laid out checkpoint
                    m = st["exp_avg"] = torch.empty_like(p).copy_(m)
                    v = st["exp_avg_sq"] = torch.empty_like(p).copy_(v)
                if self._device_step:
                    key = 0
                    if hyper is None:
                        hyper = self._group_hyper(gi, group, p.device)
                    if st["step"].data_ptr() != hyper.data_ptr():  # state came from load_state_dict: adopt its counter
                        hyper[0].copy_(st["step"])
                        st["step"] = hyper[0]
                else:
                    st["step"] += 1
                    key = int(st["step"].item())
                buckets.setdefault(key, []).append((p, g, m, v))
            for key, rows in buckets.items():
                n = len(rows)
                arr = C.c_void_p * n
                ps = arr(*[r[0].data_ptr() for r in rows])
                gs = arr(*[r[1].data_ptr() for r in rows])
                ms = arr(*[r[2].data_ptr() for r in rows])
                vs = arr(*[r[3].data_ptr() for r in rows])
                ns = (C.c_size_t * n)(*[r[0].numel() for r in rows])
                dev = rows[0][0].device
                if self._device_step and not torch.cuda.is_current_stream_capturing():
                    self.sync_hyper()
                L.call("movae_adam_multi", n, ps, gs, ms, vs, ns, float(group["lr"]), float(beta1), float(beta2),
                       float(group["eps"]), float(group["weight_decay"]), 1 if self._decoupled else 0,
                       max(1, key), L.ptr(hyper) if self._device_step else 0, L.stream_ptr(dev))
        return loss


class FusedAdamW(FusedAdam):
    """torch.optim.AdamW: decoupled weight decay, default 1e-2 (main.py:1173-1174)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, device_step=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decoupled_weight_decay=True,
                         device_step=device_step)


def clip_grad_norm_(parameters, max_norm, norm_type=2.0):
    """torch.nn.utils.clip_grad_norm_(parameters, max_norm) as called at main.py:211-212, in three launches for the
    whole parameter list and without a host sync (so it can sit inside a captured hipGraph).  Returns the total norm
    as a 0-dim device tensor, like torch."""
    if float(norm_type) != 2.0:
        raise NotImplementedError("clip_grad_norm_: only the L2 norm of the reference's call is implemented")
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return torch.tensor(0.0)
    for g in grads:
        L.require_gpu(g)
        if g.dtype != torch.float32 or not _dense(g):
            raise NotImplementedError("clip_grad_norm_: gradients must be dense fp32 tensors")
    dev = grads[0].device
    n = len(grads)
    sumsq = torch.empty((), dtype=torch.float32, device=dev)
    ws = L.workspace(dev)
    L.call("movae_clip_grad_norm_multi", n, (C.c_void_p * n)(*[g.data_ptr() for g in grads]),
           (C.c_size_t * n)(*[g.numel() for g in grads]), float(max_norm), sumsq.data_ptr(), ws.data_ptr(), ws.numel(),
           L.stream_ptr(dev))
    return sumsq.sqrt()
