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
import os

import torch

from . import ops

#: MOVAE_BATCHED_VJP=0 falls back to one torch.autograd pass per loss through the shared graph
BATCHED_VJP = os.environ.get("MOVAE_BATCHED_VJP", "1") != "0"


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


_ONES = {}


def _ones_like(t):
    key = (t.device, t.dtype)
    o = _ONES.get(key)
    if o is None:
        with torch.no_grad():
            o = _ONES[key] = torch.ones((), dtype=t.dtype, device=t.device)
    return o


def _accumulate(p, g, taken=()):
    """p.grad += g.  `taken`: parameters whose gradient a kernel already added into p.grad in place (ops.GRAD_ACCUM_TAKEN).
    Such a parameter must come back as that very tensor: a parameter that feeds TWO nodes of one loss (weight sharing) is
    accumulated in place by the first node only, autograd then sums the two results into a NEW tensor (old + g1) + g2, and adding
    that to p.grad would count old + g1 twice -- raised, not silently added."""
    if p.grad is None:
        p.grad = g
    elif g.data_ptr() == p.grad.data_ptr() and g.shape == p.grad.shape:
        pass  # the kernels already added this loss's gradient into p.grad (ops.GRAD_ACCUM)
    elif p.data_ptr() in taken:
        raise RuntimeError("a parameter's gradient was accumulated in place by a weight-gradient kernel, but autograd returned a "
                           "different tensor for it (the parameter feeds more than one node of this loss?): in-place accumulation "
                           "does not support shared weights")
    else:
        ops.join_wgrad()  # both operands may still be in flight on the weight-gradient side stream
        ops.flush_deferred()  # ... or wait in a parked split-K reduce (ops.deferred_reduces)
        p.grad = p.grad + g


#: MOVAE_PERSISTENT_J=0: a fresh zero-filled Jacobian arena per step (A/B knob)
PERSISTENT_J = os.environ.get("MOVAE_PERSISTENT_J", "1") != "0"
_J_CACHE = {}  # (device, K, parameter identities) -> JacobianBuffer, the few most recently used


class JacobianBuffer:
    """[K, ld] fp32 arena; ld is padded to a multiple of 4 floats so every row is 16-byte aligned.

    A slice nobody writes must read zero (a loss with no path to a parameter; the bias in front of a training-mode BatchNorm).
    `JacobianBuffer.persistent` keeps the arena across steps instead of zero-filling K x m floats per step (13.6 MB at C2,
    180 MB at C5): the slices written in a step are known from ops.SINK_LOG / write_row, `settle()` zeroes exactly those that
    an EARLIER step wrote and this one did not (normally none: the step's graph does not change)."""

    def __init__(self, params, k, device):
        self.params = params
        self.offsets, off = [], 0
        for p in params:
            self.offsets.append(off)
            off += p.numel()
        self.m = off
        self.ld = (off + 3) // 4 * 4
        self.buf = torch.zeros((k, max(self.ld, 4)), dtype=torch.float32, device=device)
        self.dirty = None      # persistent use: data_ptr -> (row, parameter index) of every slice some step wrote
        self.copied = set()    # slices written by write_row / the walker's leaf copies in the current step

    @classmethod
    def persistent(cls, params, k, device):
        if not PERSISTENT_J:
            return cls(params, k, device)
        key = (str(device), k, tuple((id(p), p.numel()) for p in params))
        jb = _J_CACHE.pop(key, None)
        if jb is None or any(a is not b for a, b in zip(jb.params, params)):
            jb = cls(list(params), k, device)
            jb.dirty = {}
        _J_CACHE[key] = jb  # (re-inserted last: most recently used)
        while len(_J_CACHE) > 4:
            _J_CACHE.pop(next(iter(_J_CACHE)))
        jb.copied = set()
        ops.SINK_LOG.clear()
        ops.SINK_ZERO_LOG.clear()
        return jb

    def settle(self):
        """After the step's backward: zero the slices an earlier step wrote and this one did not."""
        if self.dirty is None:
            return
        where = {}
        for i in range(self.buf.shape[0]):
            base = self.buf[i].data_ptr()
            for j, (p, off) in enumerate(zip(self.params, self.offsets)):
                where[base + 4 * off] = (i, j)
        now = {ptr for ptr in list(ops.SINK_LOG) + list(self.copied) if ptr in where}
        for ptr in [q for q in self.dirty if q not in now]:
            i, j = self.dirty[ptr]
            self.buf[i][self.offsets[j]: self.offsets[j] + self.params[j].numel()].zero_()
            del self.dirty[ptr]
        for ptr in now:
            self.dirty[ptr] = where[ptr]
        ops.SINK_LOG.clear()
        ops.SINK_ZERO_LOG.clear()

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
                self.copied.add(dst.data_ptr())
                if g.data_ptr() != dst.data_ptr():  # already written in place through the sink
                    ops.join_wgrad()
                    dst.copy_(_mem_flat(g))


def _group(t, g):
    """Group g of a stacked gradient (a [G, ...] tensor or a list of G tensors)."""
    return None if t is None else t[g]


def _dense(t):
    """Does `t` cover its numel() elements of memory exactly once (contiguous up to a permutation of its axes)?"""
    if t.is_contiguous():
        return True
    order = sorted(range(t.dim()), key=lambda i: (-t.stride(i), -t.shape[i]))
    return t.permute(order).is_contiguous()


def _restack(lst):
    """Per-group results of a torch-native view node are usually G equal slices of one stacked buffer: hand them on as the
    [G, ...] view they are (no copy), so that later accumulations / kernels see one tensor instead of a list.  The slices may be
    permuted views (a logical-NCHW view of an NHWC buffer): dense is enough."""
    t0 = lst[0]
    if not isinstance(t0, torch.Tensor) or t0.numel() == 0 or not _dense(t0):
        return lst
    nbytes = t0.numel() * t0.element_size()
    base = t0.untyped_storage().data_ptr()
    for g, t in enumerate(lst):
        if (not isinstance(t, torch.Tensor) or t.shape != t0.shape or t.dtype != t0.dtype or t.stride() != t0.stride() or
                t.untyped_storage().data_ptr() != base or t.data_ptr() != t0.data_ptr() + g * nbytes):
            return lst
    return t0.as_strided((len(lst),) + tuple(t0.shape), (t0.numel(),) + tuple(t0.stride()), t0.storage_offset())


def _accumulate_stacked(a, b, G):
    if a is None:
        return b
    if isinstance(a, (list, tuple)) or isinstance(b, (list, tuple)):
        return [a[g] + b[g] for g in range(G)]
    return a + b


def _expected_shape(node, input_nr):
    """Shape autograd expects for gradient `input_nr` flowing into `node` (None when unknown)."""
    var = getattr(node, "variable", None)
    if var is not None:
        return tuple(var.shape)
    meta = getattr(node, "_input_metadata", None)
    if meta is not None and input_nr < len(meta):
        return tuple(meta[input_nr].shape)
    return None


def _fit_groups(t, shape, G):
    """torch.autograd reduces a node's output to the shape of the forward input it belongs to (broadcast operands);
    calling nodes directly skips that step, so it is applied here, per group."""
    if shape is None:
        return t
    if isinstance(t, (list, tuple)):
        return [x if x is None or tuple(x.shape) == shape else x.sum_to_size(shape) for x in t]
    if tuple(t.shape[1:]) == shape:
        return t
    return [t[g].sum_to_size(shape) for g in range(G)]


@torch.no_grad()  # nodes are called directly; without this their formulas would be recorded as a new graph
def _batched_pullback(features, feat_grads, rows, jb):
    """Pull G = len(rows) cotangents of `features` back through the shared graph in ONE traversal and write row
    rows[g] of the Jacobian arena for group g.

    torch.autograd runs one pass per cotangent; the shared encoder's layers are tiny at the step's batch sizes,
    so each pass is a chain of launches that cannot fill the chip.  The pull-back is linear in the cotangent and
    per-sample for every op except BatchNorm (whose batch means are taken per group), so this walker executes
    each node once on the stacked cotangents [G, ...]: nodes created by ops.py expose `backward_batched`
    (dgrad over G*n images in one launch); any other node (views, reshapes, slices ...) is called once per group.
    feat_grads[g][f] is d(loss rows[g]) / d(features[f]) or None."""
    G = len(rows)
    # ---- discover the graph below the features and count incoming edges -------------------------------------------
    roots = []
    for f, feat in enumerate(features):
        fn = feat.grad_fn
        if fn is None:
            continue
        gs = [feat_grads[g][f] for g in range(G)]
        if all(x is None for x in gs):
            continue
        ref = next(x for x in gs if x is not None)
        gs = [x if x is not None else torch.zeros_like(ref) for x in gs]
        roots.append((fn, feat.output_nr, _restack(gs)))  # (adjacent slices of one buffer -- ops.COT_SINK -- become one [G, ...] view)
    deps, seen, stack = {}, set(), [fn for fn, _, _ in roots]
    while stack:
        fn = stack.pop()
        if fn in seen:
            continue
        seen.add(fn)
        for nf, _ in fn.next_functions:
            if nf is not None:
                deps[nf] = deps.get(nf, 0) + 1
                stack.append(nf)
    pending = {}  # node -> {input_nr: stacked grad}
    for fn, nr, gs in roots:
        slot = pending.setdefault(fn, {})
        slot[nr] = _accumulate_stacked(slot.get(nr), gs, G)
    ready = [fn for fn in {r[0] for r in roots} if deps.get(fn, 0) == 0]
    col = {id(p): (p, off) for p, off in zip(jb.params, jb.offsets)}
    ops.GRAD_SINK_ROWS = [jb.sinks(r) for r in rows]
    try:
        while ready:
            fn = ready.pop()
            got = pending.pop(fn, {})
            var = getattr(fn, "variable", None)
            outs = None
            if var is not None:  # AccumulateGrad: a leaf
                ent = col.get(id(var))
                if ent is not None and 0 in got:
                    p, off = ent
                    for g in range(G):
                        src = _group(got[0], g)
                        dst = jb.buf[rows[g]][off: off + p.numel()]
                        jb.copied.add(dst.data_ptr())
                        if src.data_ptr() != dst.data_ptr():  # not already written in place through the sink
                            dst.copy_(_mem_flat(src))
                continue
            edges = list(fn.next_functions)
            if got:
                n_in = max(got) + 1
                args = [got.get(i) for i in range(n_in)]
                cls = getattr(fn, "_forward_cls", None)
                if cls is not None:  # a torch.autograd.Function node (ops.py): backward returns one value per forward ARGUMENT
                    if hasattr(cls, "backward_batched"):
                        outs = cls.backward_batched(fn, G, *args)
                    else:
                        per = []
                        for g in range(G):
                            r = cls.backward(fn, *[_group(a, g) for a in args])
                            per.append(r if isinstance(r, tuple) else (r,))
                        outs = tuple(None if all(per[g][i] is None for g in range(G)) else [per[g][i] for g in range(G)]
                                     for i in range(len(per[0])))
                    if not isinstance(outs, tuple):
                        outs = (outs,)
                    # edges exist per TENSOR argument; the ones that are not None are, in order, the arguments that need grad
                    need = [i for i, nd in enumerate(fn.needs_input_grad) if nd]
                    live = [j for j, (nf, _) in enumerate(edges) if nf is not None]
                    if len(need) != len(live):
                        raise NotImplementedError("cannot align backward outputs with graph edges")
                    aligned = [None] * len(edges)
                    for j, i in zip(live, need):
                        aligned[j] = outs[i] if i < len(outs) else None
                    outs = tuple(aligned)
                else:  # a torch-native node (views, reshapes, slices ...): once per group
                    per = []
                    for g in range(G):
                        r = fn(*[_group(a, g) for a in args])
                        per.append(r if isinstance(r, tuple) else (r,))
                    outs = tuple(None if all(per[g][i] is None for g in range(G)) else _restack([per[g][i] for g in range(G)])
                                 for i in range(len(per[0])))
            for i, (nf, nr) in enumerate(edges):
                if nf is None:
                    continue
                o = outs[i] if outs is not None and i < len(outs) else None
                if o is not None:
                    o = _fit_groups(o, _expected_shape(nf, nr), G)  # the engine's validate_outputs: un-broadcast
                    slot = pending.setdefault(nf, {})
                    slot[nr] = _accumulate_stacked(slot.get(nr), o, G)
                deps[nf] -= 1
                if deps[nf] == 0:
                    ready.append(nf)
    finally:
        ops.GRAD_SINK_ROWS = None


def _aggregate_into_grads(jb, aggregator):
    g = aggregator(jb.J)
    for p, off in zip(jb.params, jb.offsets):
        _accumulate(p, _view_like(g[off: off + p.numel()], p))
    return g


class _MtlState:
    """What mtl_backward_begin hands to mtl_backward_finish: the Jacobian buffer and the per-loss feature cotangents."""
    __slots__ = ("jb", "feat_diff", "feat_grads", "shared_params", "aggregator", "task_params", "device")


def mtl_backward_begin(losses, features, aggregator, tasks_params=None, shared_params=None):
    """First half of mtl_backward: every loss is differentiated down to the features; the task-side parameters
    (decoder, codebooks, ...) receive their final .grad here.  Nothing of the shared trunk has been touched yet, which
    is what lets a data-parallel step start reducing the task-side gradients while mtl_backward_finish runs."""
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
    st = _MtlState()
    st.shared_params = list(shared_params)
    st.aggregator = aggregator
    st.jb = JacobianBuffer.persistent(st.shared_params, len(losses), features[0].device)
    st.feat_diff = [f for f in features if f.requires_grad]
    st.feat_grads = []
    st.task_params = []
    # the K cotangents of every feature are born stacked: [K, ...] buffers whose slice i the op that produces d(loss i)/d(feature)
    # writes directly (ops.COT_SINK) -- no torch.stack launch in front of the batched pull-back
    cot = [torch.empty_strided((len(losses),) + tuple(f.shape), (f.numel(),) + tuple(f.stride()), dtype=f.dtype, device=f.device)
           if _dense(f) else None for f in st.feat_diff]  # (slice i has the feature's own strides: NHWC memory under an NCHW view)
    for i, (loss, tp) in enumerate(zip(losses, tasks_params)):
        tp = list(tp)
        # the seed cotangent is a persistent ones tensor: autograd would launch a fill per loss for its implicit ones_like
        seed = _ones_like(loss) if loss.dim() == 0 and loss.dtype == torch.float32 else None
        # task-side parameters an earlier loss already reached get this loss's gradient added inside the weight-gradient kernels
        ops.GRAD_ACCUM.clear()
        ops.GRAD_ACCUM_TAKEN.clear()
        if i > 0 and ops.L.DEFER is None:
            ops.GRAD_ACCUM.update({p.data_ptr(): p.grad for p in tp if p.grad is not None})
        ops.COT_SINK.clear()
        ops.COT_SINK.update({f.data_ptr(): c[i] for f, c in zip(st.feat_diff, cot) if c is not None})
        # a task-side parameter no earlier loss reached gets its gradient written into a buffer kept across steps (no allocation;
        # and a destination the kernels know nobody reads before the optimizer: its split-K reduce may ride on a later launch)
        ops.GRAD_SINK.clear()
        owned = {}
        if ops.L.DEFER is None:
            for p in tp:
                if p.grad is None:
                    buf = ops.GRAD_SINK[p.data_ptr()] = _task_grad_buffer(p)
                    owned[buf.data_ptr()] = p
        n_w, n_z = len(ops.SINK_LOG), len(ops.SINK_ZERO_LOG)
        try:
            with ops.deferred_reduces():
                got = torch.autograd.grad(loss, tp + st.feat_diff, grad_outputs=seed, retain_graph=True, allow_unused=True)
        finally:
            # a buffer handed out as an untouched all-zero gradient (ops._sink_zeros) must BE zero: it is, unless an earlier
            # step wrote it (a BatchNorm switched between eval and training mode) -- re-zeroed then, once
            for ptr in ops.SINK_LOG[n_w:]:
                if ptr in owned:
                    owned[ptr]._movae_grad_written = True
            for ptr in ops.SINK_ZERO_LOG[n_z:]:
                if ptr in owned and getattr(owned[ptr], "_movae_grad_written", False):
                    owned[ptr]._movae_grad_buf.zero_()
                    owned[ptr]._movae_grad_written = False
            ops.GRAD_SINK.clear()
            ops.COT_SINK.clear()
            ops.GRAD_ACCUM.clear()
            taken = frozenset(ops.GRAD_ACCUM_TAKEN)
            ops.GRAD_ACCUM_TAKEN.clear()
        for p, g in zip(tp, got[: len(tp)]):
            if g is not None:
                _accumulate(p, g, taken)
                st.task_params.append(p)
        st.feat_grads.append(got[len(tp):])
    return st


def _task_grad_buffer(p):
    """Flat fp32 buffer of p's size, one per parameter for life (the memory image of the gradient: ops._sink views it)."""
    buf = getattr(p, "_movae_grad_buf", None)
    if buf is None or buf.numel() != p.numel() or buf.device != p.device:
        buf = torch.zeros(p.numel(), dtype=p.dtype, device=p.device)  # (zero: ops._sink_zeros hands a sink out untouched)
        p._movae_grad_buf, p._movae_grad_written = buf, False
    return buf


def mtl_backward_finish(st):
    with ops.deferred_reduces():
        return _mtl_backward_finish(st)


def _mtl_backward_finish(st):
    """Second half: the K feature cotangents are pulled back through the shared trunk (batched), the Jacobian is
    aggregated and the result lands in the shared parameters' .grad."""
    jb, feat_diff, feat_grads, shared_params = st.jb, st.feat_diff, st.feat_grads, st.shared_params
    live_rows = [i for i, gf in enumerate(feat_grads) if any(g is not None for g in gf)]
    batched = BATCHED_VJP and shared_params and len(live_rows) > 1
    if batched:
        try:
            _batched_pullback(feat_diff, [feat_grads[i] for i in live_rows], live_rows, jb)
        except NotImplementedError:
            batched = False  # a node the walker cannot align: one autograd pass per loss instead (rows are rewritten)
    for i in ([] if batched else live_rows):
        gf = feat_grads[i]
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
        ops.join_wgrad()  # the Jacobian rows are written by deferred weight-gradient launches
        jb.settle()
        _aggregate_into_grads(jb, st.aggregator)


def mtl_backward(losses, features, aggregator, tasks_params=None, shared_params=None, retain_graph=False,
                 parallel_chunk_size=None):
    st = mtl_backward_begin(losses, features, aggregator, tasks_params, shared_params)
    mtl_backward_finish(st)
    if not retain_graph:
        # torchjd frees the graph on the last differentiation; torch offers no explicit free, dropping
        # the references is enough for the caller's tensors to release their buffers
        del st


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
    ops.join_wgrad()
    _aggregate_into_grads(jb, aggregator)
