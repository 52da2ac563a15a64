"""Oracle: Jacobian descent drivers (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates the semantics of torchjd.autojac.mtl_backward / backward at the reference's call site
main.py:176-196 (torchjd is third-party and absent: parity unpinned beyond the invariant that
unit weights reproduce ``total_loss.backward()`` for non-nested features, checked in tests/).
Functional: gradients are returned, not accumulated into ``.grad``.
"""
from collections import OrderedDict

import torch

from . import aggregation as A


def _leaves(roots, excluded=()):
    """Leaf tensors reachable from ``roots`` without passing through an ``excluded`` tensor's
    graph node (torchjd get_leaf_tensors)."""
    stop = {t.grad_fn for t in excluded if t.grad_fn is not None}
    seen, out, stack = set(), [], [t.grad_fn for t in roots if t.grad_fn is not None]
    while stack:
        fn = stack.pop()
        if fn is None or fn in seen or fn in stop:
            continue
        seen.add(fn)
        if hasattr(fn, "variable"):
            out.append(fn.variable)
        stack.extend(nf for nf, _ in fn.next_functions)
    return out


def split_params(named_params, losses, features):
    """-> (shared names, [task-specific names per loss])."""
    ident = {id(p): n for n, p in named_params.items()}
    shared_ids = {id(t) for t in _leaves(features)}
    shared = [n for n, p in named_params.items() if id(p) in shared_ids]
    tasks = []
    for l in losses:
        ids = {id(t) for t in _leaves([l], excluded=features)}
        tasks.append([n for n, p in named_params.items() if id(p) in ids])
    return shared, tasks


def jacobian_mtl(named_params, losses, features):
    """Rows of J over the shared parameters (concatenated in ``named_params`` order) and the
    summed task-specific gradients."""
    shared, tasks = split_params(named_params, losses, features)
    sp = [named_params[n] for n in shared]
    task_grads = OrderedDict()
    rows = []
    for l, tn in zip(losses, tasks):
        tp = [named_params[n] for n in tn]
        got = torch.autograd.grad(l, tp + list(features), retain_graph=True, allow_unused=True)
        for n, p, g in zip(tn, tp, got[: len(tp)]):
            g = torch.zeros_like(p) if g is None else g
            task_grads[n] = task_grads[n] + g if n in task_grads else g
        gf = [torch.zeros_like(f) if g is None else g for f, g in zip(features, got[len(tp):])]
        js = torch.autograd.grad(list(features), sp, grad_outputs=gf, retain_graph=True, allow_unused=True)
        rows.append(torch.cat([(torch.zeros_like(p) if g is None else g).reshape(-1) for p, g in zip(sp, js)]))
    return torch.stack(rows), shared, task_grads


def jacobian_full(named_params, losses):
    """torchjd.autojac.backward: every parameter is shared."""
    names = list(named_params)
    ps = [named_params[n] for n in names]
    rows = []
    for l in losses:
        gs = torch.autograd.grad(l, ps, retain_graph=True, allow_unused=True)
        rows.append(torch.cat([(torch.zeros_like(p) if g is None else g).reshape(-1) for p, g in zip(ps, gs)]))
    return torch.stack(rows), names


def mtl_backward(named_params, losses, features, weighting, loss_values=None):
    """-> (grads dict name->tensor, info dict with J, G, w)."""
    J, shared, task_grads = jacobian_mtl(named_params, losses, features)
    g, w, G = A.aggregate(J, weighting, loss_values)
    grads, off = OrderedDict(), 0
    for n in shared:
        p = named_params[n]
        grads[n] = g[off: off + p.numel()].reshape(p.shape)
        off += p.numel()
    for n, t in task_grads.items():
        grads[n] = grads[n] + t if n in grads else t
    return grads, dict(J=J, G=G, w=w, shared=shared)
