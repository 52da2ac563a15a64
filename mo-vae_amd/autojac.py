"""Jacobian-descent drivers with torchjd.autojac's call signatures (call sites main.py:188-196):

    mtl_backward(losses, features, aggregator, retain_graph=True) -> None   (accumulates into .grad)
    backward(tensors, aggregator=...)                              -> None

Semantics restated from torchjd's published behaviour (third-party, not vendored by the reference):
per loss, gradients w.r.t. the task-specific parameters (leaves reachable from the loss without
passing through a feature) are summed into .grad; the per-loss feature gradients are pulled back
through the shared graph into the rows of J [K, m]; the shared parameters receive aggregator(J).
J lives in one padded device buffer whose columns follow the parameters' memory order, so the
aggregated vector is handed to .grad as views (no scatter copies), and a loss with no path to the
features (e.g. VQ embedding_loss) costs no backward pass -- its row is simply left zero.
"""
import torch

from . import ops


def _leaf_tensors(roots, excluded=()):
    stop = {t.grad_fn for t in excluded if t.grad_fn is not None}
    seen, leaves, stack = set(), [], [t.grad_fn for t in roots if t.grad_fn is not None]
    while stack:
        fn = stack.pop()
        if fn is None or fn in seen or fn in stop:
            continue
        seen.add(fn)
        var = getattr(fn, "variable", None)
        if var is not None:
            if var.requires_grad:
                leaves.append(var)
            continue
        stack.extend(nf for nf, _ in fn.next_functions)
    # deterministic order: de-duplicate keeping first occurrence
    out, ids = [], set()
    for v in leaves:
        if id(v) not in ids:
            ids.add(id(v))
            out.append(v)
    return out


def _mem_flat(t):
    """1-D view of a tensor in MEMORY order (channels_last conv weights flatten as [o][kh][kw][i])."""
    if t.dim() == 4 and not t.is_contiguous():
        v = t.permute(0, 2, 3, 1)
        if v.is_contiguous():
            return v.reshape(-1)
    return t.contiguous().reshape(-1) if not t.is_contiguous() else t.reshape(-1)


def _view_like(flat, p):
    """View of a flat memory-order slice with p's logical shape and strides."""
    return flat.as_strided(p.shape, p.stride())


def _accumulate(p, g):
    if p.grad is None:
        p.grad = g
    else:
        p.grad = p.grad + g


class JacobianBuffer:
    """[K, ld] fp32 arena; ld is padded to a multiple of 4 floats so every row is 16-byte aligned."""

    def __init__(self, params, k, device):
        self.params = params
        self.offsets, off = [], 0
        for p in params:
            self.offsets.append(off)
            off += p.numel()
        self.m = off
        self.ld = (off + 3) // 4 * 4
        self.buf = torch.zeros((k, max(self.ld, 4)), dtype=torch.float32, device=device)

    @property
    def J(self):
        return self.buf[:, : self.m]

    def sinks(self, i):
        """{param data_ptr: row slice}: lets the backward kernels write row i of J in place (ops.GRAD_SINK)."""
        row = self.buf[i]
        return {p.data_ptr(): row[off: off + p.numel()] for p, off in zip(self.params, self.offsets)}

    def write_row(self, i, grads):
        row = self.buf[i]
        for p, off, g in zip(self.params, self.offsets, grads):
            if g is not None:
                dst = row[off: off + p.numel()]
                if g.data_ptr() != dst.data_ptr():  # already written in place through the sink
                    dst.copy_(_mem_flat(g))


def _aggregate_into_grads(jb, aggregator):
    g = aggregator(jb.J)
    for p, off in zip(jb.params, jb.offsets):
        _accumulate(p, _view_like(g[off: off + p.numel()], p))
    return g


def mtl_backward(losses, features, aggregator, tasks_params=None, shared_params=None, retain_graph=False,
                 parallel_chunk_size=None):
    losses, features = list(losses), list(features)
    if len(losses) == 0:
        raise ValueError("`losses` cannot be empty")
    if len(features) == 0:
        raise ValueError("`features` cannot be empty.")
    if shared_params is None:
        shared_params = _leaf_tensors(features)
    if tasks_params is None:
        tasks_params = [_leaf_tensors([l], excluded=features) for l in losses]
    if len(tasks_params) != len(losses):
        raise ValueError("`losses` and `tasks_params` should have the same size.")
    shared_params = list(shared_params)
    jb = JacobianBuffer(shared_params, len(losses), features[0].device)
    feat_diff = [f for f in features if f.requires_grad]
    for i, (loss, tp) in enumerate(zip(losses, tasks_params)):
        tp = list(tp)
        got = torch.autograd.grad(loss, tp + feat_diff, retain_graph=True, allow_unused=True)
        for p, g in zip(tp, got[: len(tp)]):
            if g is not None:
                _accumulate(p, g)
        gf = got[len(tp):]
        live = [(f, g) for f, g in zip(feat_diff, gf) if g is not None]
        if not live or not shared_params:
            continue  # no path from this loss to the shared parameters: zero Jacobian row
        ops.GRAD_SINK.clear()
        ops.GRAD_SINK.update(jb.sinks(i))
        try:
            js = torch.autograd.grad([f for f, _ in live], shared_params, grad_outputs=[g for _, g in live],
                                     retain_graph=True, allow_unused=True)
        finally:
            ops.GRAD_SINK.clear()
        jb.write_row(i, js)
    if shared_params:
        _aggregate_into_grads(jb, aggregator)
    if not retain_graph:
        # torchjd frees the graph on the last differentiation; torch offers no explicit free, dropping
        # the references is enough for the caller's tensors to release their buffers
        del jb


def backward(tensors, aggregator, inputs=None, retain_graph=False, parallel_chunk_size=None):
    tensors = list(tensors) if isinstance(tensors, (list, tuple)) else [tensors]
    if len(tensors) == 0:
        raise ValueError("`tensors` cannot be empty")
    if inputs is None:
        inputs = _leaf_tensors(tensors)
    inputs = list(inputs)
    if not inputs:
        return
    jb = JacobianBuffer(inputs, len(tensors), tensors[0].device)
    for i, t in enumerate(tensors):
        js = torch.autograd.grad(t, inputs, retain_graph=True, allow_unused=True)
        jb.write_row(i, js)
    _aggregate_into_grads(jb, aggregator)
