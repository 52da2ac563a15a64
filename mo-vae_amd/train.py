"""Training loop and CLI of the hot path -- drop-in for the reference's main.py:125-235 (train_epoch),
:1088-1497 (main) and :1500-1670 (argparse), restricted to what the per-step path needs.

Differences that are by design: the step's device->host reads (losses, hook values) are
batched into ONE copy per step instead of ~K+5 `.item()` syncs; evaluation metrics that need
pretrained networks (FID/IS/LPIPS) and dataset downloads are out of scope -- `--dataset synthetic_*`
provides seeded in-memory data of the right shape; with torchrun (WORLD_SIZE>1) the batch is sharded
and the aggregated gradient is all-reduced once per step over RCCL.
"""
import json
import math
import os
import time
from argparse import ArgumentParser

import torch
import torch.optim as optim

from . import _lib as L
from . import aggregation
from . import autojac
from . import ops
from .aggregation import COMFORT, MGDA
from .models import get_network
from .optim import FusedAdam, FusedAdamW, clip_grad_norm_
from .parallel import DataParallelGrads

_current_step = 0
_hook_values = {}


class AverageMeter:
    """utils/utils.py:62-109."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = 0.0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def set_seed(seed):
    """utils/utils.py:58-60."""
    import numpy as np

    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)


# ---- aggregator hooks (main.py:71-122): computed on device, read with the step's single host copy ----
def print_weights(_, __, weights):
    _hook_values["weights"] = weights


def print_gd_similarity(_, inputs, weights):
    _hook_values["similarity"] = aggregation.gd_similarity(inputs[0], weights)


def _stacked(losses):
    """torch.stack of the detached component losses -- as a view when they already lie next to each other in one buffer (the
    fused loss kernels write them that way: ops.VAELosses / ops.CombineLosses), so no launch is spent on it."""
    first = losses[0]
    if all(c.dim() == 0 and c.dtype == first.dtype and c.untyped_storage().data_ptr() == first.untyped_storage().data_ptr()
           and c.storage_offset() == first.storage_offset() + i for i, c in enumerate(losses)):
        return first.detach().as_strided((len(losses),), (1,))
    return torch.stack([c.detach() for c in losses])


def forward_backward_begin(net, images, optimizer, aggregator):
    """zero_grad, forward, losses and the backward down to the features (main.py:157-196, first half of torchjd's
    mtl_backward).  Returns (loss_dict, outputs, pending): `pending` is None when the whole backward is already done
    (sum / no features), else the state forward_backward_finish completes."""
    optimizer.zero_grad()
    outputs = net(images)
    loss_dict = net.loss_function(images, args=outputs)
    # weight gradients leave the backward's critical path (ops.wgrad_side_stream); the block's exit is the one join
    with ops.wgrad_side_stream(images.device, enabled=L.DEFER_WGRAD_DEFAULT):
        if aggregator is None or aggregator == "sum":
            loss_dict["total_loss"].backward()
            return loss_dict, outputs, None
        features = [outputs[f] for f in net.features] if net.features is not None else None
        component_losses = [v for k, v in loss_dict.items() if k != "total_loss"]  # main.py:184
        if isinstance(aggregator, (MGDA, COMFORT)):  # main.py:185
            aggregator.set_losses(_stacked(component_losses))
        if features is None:
            autojac.backward(component_losses, aggregator=aggregator)
            return loss_dict, outputs, None
        pending = autojac.mtl_backward_begin(component_losses, features, aggregator)
    pending.device = images.device
    return loss_dict, outputs, pending


def forward_backward_finish(pending):
    if pending is not None:
        with ops.wgrad_side_stream(pending.device, enabled=L.DEFER_WGRAD_DEFAULT):
            autojac.mtl_backward_finish(pending)


def forward_backward(net, images, optimizer, aggregator):
    """zero_grad, forward, losses and the (aggregated) backward of one step (main.py:157-196)."""
    loss_dict, outputs, pending = forward_backward_begin(net, images, optimizer, aggregator)
    forward_backward_finish(pending)
    return loss_dict, outputs


def train_step(net, images, optimizer, aggregator, args, dp=None):
    """One optimisation step (main.py:155-214).  Returns the loss dict (device tensors) and outputs."""
    loss_dict, outputs = forward_backward(net, images, optimizer, aggregator)
    if dp is not None:
        dp.all_reduce_grads()
    if getattr(args, "max_grad_norm", None) is not None:
        clip_grad_norm_(net.parameters(), max_norm=args.max_grad_norm)  # optim.py: 3 launches, no host sync
    optimizer.step()
    return loss_dict, outputs




DP_OVERLAP_MIN_BYTES = 32 << 20


def _state_tensors(net):
    # BetaTC's device-side step counter; the in-kernel noise generator's {seed, draws} (models/_base.py)
    extra = [t for t in (getattr(net, "_iter_dev", None), getattr(net, "_noise_state_t", None)) if isinstance(t, torch.Tensor)]
    return list(net.parameters()) + list(net.buffers()) + extra


def _snapshot_state(net, optimizer, device):
    return {"tensors": [t.detach().clone() for t in _state_tensors(net)],
            "opt": {id(p): {k: v.detach().clone() for k, v in st.items() if torch.is_tensor(v)} for p, st in optimizer.state.items()},
            "cuda_rng": torch.cuda.get_rng_state(device), "cpu_rng": torch.get_rng_state()}


@torch.no_grad()
def _restore_state(snap, net, optimizer, device):
    torch.cuda.synchronize(device)
    for t, s in zip(_state_tensors(net), snap["tensors"]):
        t.copy_(s)
    for p, st in optimizer.state.items():  # the state tensors keep their addresses (the captured graph points at them)
        old = snap["opt"].get(id(p), {})
        for k, v in st.items():
            if torch.is_tensor(v):
                if k in old:
                    v.copy_(old[k])
                else:
                    v.zero_()
    torch.cuda.set_rng_state(snap["cuda_rng"], device)
    torch.set_rng_state(snap["cpu_rng"])


class GraphedTrainStep:
    """The whole optimisation step (forward, losses, K per-loss backward passes, Gram / solve / combine,
    optimizer) captured ONCE into a hipGraph and replayed per batch: the step is ~250 short kernels, so
    eager launches are host-bound; a replay is one submission.  No tracing compiler is involved -- the
    graph holds exactly the launches the eager step made.  The optimizer must keep its step counter on
    the device: make_optimizer(..., capturable=True) -> FusedAdam(device_step=True).

    Data parallel (dp given): the step is three graphs around two eager collectives over RCCL --
    graph 1 = forward, losses and the backward down to the features, task-side gradients flattened into bucket A;
    graph 1b = the shared trunk's batched pull-back + aggregation, flattened into bucket B, running while bucket A is
    all-reduced on RCCL's stream; then all-reduce(B); graph 2 = gradient clipping + optimizer step on views of the
    buckets (the 1/N rides in the collective: ncclAvg).  `--agg sum` has no second half: one bucket, one collective.
    Gradients under DP_OVERLAP_MIN_BYTES keep the single-bucket form (the split's fixed cost exceeds what it hides);
    MOVAE_DP_OVERLAP=0 / 1 forces either."""

    def __init__(self, net, optimizer, aggregator, args, example, warmup=3, dp=None, record_calls=False, preserve_state=False):
        if not getattr(net, "graph_safe", False):
            raise NotImplementedError(f"{type(net).__name__}: forward syncs with the host; use the eager train_step")
        net.prepare_for_graph()
        # preserve_state: the warm-up steps a capture needs are real optimisation steps; with this flag everything they
        # touched (parameters, buffers, optimizer state, RNG streams) is rewound afterwards, so the first replay is step 1
        # of the run exactly as the eager loop would have made it (train_epoch uses this)
        snap = _snapshot_state(net, optimizer, example.device) if preserve_state else None
        try:
            self._build(net, optimizer, aggregator, args, example, warmup, dp, record_calls)
        finally:
            if snap is not None:
                _restore_state(snap, net, optimizer, example.device)

    def _build(self, net, optimizer, aggregator, args, example, warmup, dp, record_calls):
        self.net, self.opt, self.agg, self.args, self.dp = net, optimizer, aggregator, args, dp
        self.static_x = example.clone()
        from . import _lib as L
        from .parallel import flatten_grads, unflatten_into_grads

        L.workspace(example.device)  # allocate the scratch arena outside the capture
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                train_step(net, self.static_x, optimizer, aggregator, args, dp)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self.graph2 = None
        self.dp_form = "single device"
        optimizer.zero_grad(set_to_none=True)
        #: (name, args) of every C-ABI launch in the captured step; the pointers stay valid (graph-pool memory) for the
        #: life of this object, which lets bench.py re-issue and time each launch on the step's real operands
        self.calls = []
        if record_calls:
            L.TRACE = lambda name, cargs: self.calls.append((name, cargs))
        try:
            if dp is None:
                ops.forget_nhwc()
                with torch.cuda.graph(self.graph):
                    self.loss_dict, self.outputs = train_step(net, self.static_x, optimizer, aggregator, args)
                ops.forget_nhwc()
                return
        finally:
            L.TRACE = None
        params = [p for p in net.parameters() if p.requires_grad]
        # the bucket handed to the collective is an ordinary allocation (not graph-pool memory)
        self.flat = torch.zeros(sum(p.numel() for p in params) + 64, dtype=torch.float32, device=example.device)
        # RCCL averages inside the collective (ncclAvg); gloo (CPU rehearsal of the N>1 path) has no AVG: sum, then scale
        self._avg = dp.backend == "nccl"
        # the process group's watchdog thread polls events of earlier collectives; under the default "global" capture mode
        # such a query from another thread invalidates the capture (seen at C4, whose capture is long enough to collide)
        cap = dict(capture_error_mode="thread_local")
        torch.cuda.synchronize()
        # splitting the step costs one more graph launch and two cross-stream waits (~60 us measured on MI355X); it pays
        # once the task-side bucket's exchange is longer than that, i.e. not for the 13 MB of the 32x32 VAEs
        mode = os.environ.get("MOVAE_DP_OVERLAP", "auto")
        overlap = mode == "1" or (mode == "auto" and self.flat.numel() * 4 >= DP_OVERLAP_MIN_BYTES)
        # One bucket over RCCL: the collective is captured too, so the whole data-parallel step is ONE replay.  Three pieces
        # (graph, eager all-reduce, graph) cost ~110 us per step in cross-queue hand-offs on MI355X (1.415 vs 1.306 ms with one
        # hardware queue); inside one graph the collective is just another kernel node.  gloo cannot be captured, and
        # MOVAE_DP_CAPTURE_COLLECTIVE=0 (or a failing capture) falls back to the pieces.
        # MOVAE_DP_CAPTURE_COLLECTIVE: "auto" (default) capture the collective and fall back to the pieces if that fails -- but
        # only when EVERY rank agrees (MIN all-reduce of an ok-flag: a rank that fell back alone would issue collectives its
        # peers never post: the pieces all-reduce while they are built); "1" capture it and RAISE if that fails (strict
        # opt-in); "0" never capture it.  The form taken is recorded in self.dp_form and logged once on rank 0 (train.main).
        cc_mode = os.environ.get("MOVAE_DP_CAPTURE_COLLECTIVE", "auto")
        self.dp_form = "graph | all-reduce | graph"
        if not overlap and dp.backend == "nccl" and cc_mode != "0":
            err = None
            try:
                ops.forget_nhwc()
                whole = torch.cuda.CUDAGraph()
                with torch.cuda.graph(whole, **cap):
                    self.loss_dict, self.outputs = forward_backward(net, self.static_x, optimizer, aggregator)
                    self.flat_a = self.flat[: sum(p.numel() for p in params)]
                    flatten_grads(params, out=self.flat_a)
                    self.reduce(self.flat_a)
                    unflatten_into_grads(self.flat_a, params)
                    if getattr(args, "max_grad_norm", None) is not None:
                        clip_grad_norm_(params, max_norm=args.max_grad_norm)
                    optimizer.step()
                ops.forget_nhwc()
            except Exception as e:  # noqa: BLE001
                if cc_mode != "auto":
                    raise
                err = e
            ok = torch.tensor([0 if err is not None else 1], device=example.device, dtype=torch.int32)
            if cc_mode == "auto":  # agree on ONE form: every rank takes the pieces if any rank's capture failed
                torch.cuda.synchronize()
                torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
            if int(ok.item()) == 1:
                self.graph, self.graph_b, self.flat_b = whole, None, None
                self.dp_form = "1 graph + captured all-reduce"
                return
            print(f"[movae] capturing the all-reduce failed on some rank ({type(err).__name__ if err else 'peer'}: {err}); "
                  "all ranks use graph / all-reduce / graph", flush=True)
            torch.cuda.synchronize()
            optimizer.zero_grad(set_to_none=True)
            self.graph = torch.cuda.CUDAGraph()
        ops.forget_nhwc()
        with torch.cuda.graph(self.graph, **cap):
            self.loss_dict, self.outputs, pending = forward_backward_begin(net, self.static_x, optimizer, aggregator)
            if pending is not None and overlap:
                shared = {id(p) for p in pending.shared_params}
                early = [p for p in params if p.grad is not None and id(p) not in shared]  # final already: task-side
            else:
                forward_backward_finish(pending)
                pending, early = None, list(params)
            n_early = sum(p.numel() for p in early)
            self.flat_a = self.flat[:n_early]
            flatten_grads(early, out=self.flat_a)
        ops.forget_nhwc()
        taken = {id(p) for p in early}
        late = [p for p in params if id(p) not in taken]
        self.graph_b, self.flat_b = None, None
        if late:
            self.dp_form = "3 graphs, two overlapped all-reduces"
            off = (n_early + 63) // 64 * 64  # keep the second bucket 256-byte aligned
            self.flat_b = self.flat[off: off + sum(p.numel() for p in late)]
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_b, pool=self.graph.pool(), **cap):
                forward_backward_finish(pending)
                flatten_grads(late, out=self.flat_b)
        # the capture passes recorded but did not execute: materialise real gradients once
        self._reduced_replay()
        unflatten_into_grads(self.flat_a, early)  # .grad := static views of the buckets, kept for every replay
        if late:
            unflatten_into_grads(self.flat_b, late)
        self.graph2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph2, pool=self.graph.pool(), **cap):
            if not self._avg:
                self.flat.div_(dp.world_size)
            if getattr(args, "max_grad_norm", None) is not None:  # same tail as train_step: clip the averaged gradient
                clip_grad_norm_(params, max_norm=args.max_grad_norm)
            optimizer.step()
        self.graph2.replay()  # completes the step whose gradients were just reduced

    def reduce(self, buf=None, async_op=False):
        """Mean over ranks of one flat gradient bucket, in place (the 1/N of gloo's sum happens in graph 2)."""
        rop = torch.distributed.ReduceOp
        return torch.distributed.all_reduce(self.flat_a if buf is None else buf, op=rop.AVG if self._avg else rop.SUM,
                                            async_op=async_op)

    def _reduced_replay(self):
        """graph 1 -> all-reduce(task-side bucket) running under graph 1b (shared trunk: batched pull-back, Gram / solve /
        combine) -> all-reduce(shared bucket).  The collectives go to RCCL's own stream (async_op) and the compute stream
        waits for both only before the optimizer graph: the first bucket's exchange is hidden behind the second half
        of the backward."""
        self.graph.replay()
        pending = [self.reduce(self.flat_a, async_op=True)]
        if self.graph_b is not None:
            self.graph_b.replay()
            pending.append(self.reduce(self.flat_b, async_op=True))
        for w in pending:
            w.wait()

    def step(self, images):
        self.static_x.copy_(images, non_blocking=True)
        if hasattr(self.opt, "sync_hyper"):
            self.opt.sync_hyper()  # lr schedulers change the host value between replays
        if self.graph2 is None:
            self.graph.replay()
        else:
            self._reduced_replay()
            self.graph2.replay()
        return self.loss_dict, self.outputs


def _host_values(loss_dict):
    """ONE device->host transfer for every scalar the step logs."""
    keys = list(loss_dict.keys())
    vals = [loss_dict[k].detach().reshape(1) for k in keys]
    extra = []
    if "weights" in _hook_values:
        extra.append(_hook_values["weights"].detach().reshape(-1))
    if "similarity" in _hook_values:
        extra.append(_hook_values["similarity"].detach().reshape(1))
    flat = torch.cat(vals + extra).cpu().tolist()
    out = dict(zip(keys, flat[: len(keys)]))
    rest = flat[len(keys):]
    if "weights" in _hook_values:
        k = _hook_values["weights"].numel()
        out["_weights"] = rest[:k]
        rest = rest[k:]
    if "similarity" in _hook_values:
        out["_similarity"] = rest[0]
    return out


def train_epoch(net, train_loader, optimizer, aggregator, step, device, args, dp=None, log=None, graphed=None):
    """main.py:125-235.  `graphed` (a dict, `--graph on`): the step is captured into a hipGraph on the first full batch
    (train.GraphedTrainStep) and replayed for every batch of that shape; ragged batches take the eager step."""
    global _current_step
    net.train()
    meters = {k: AverageMeter() for k in net.objectives.keys()}
    meters["total_loss"] = AverageMeter()
    usage = AverageMeter()
    for images, _ in train_loader:
        if getattr(args, "max_steps", None) is not None and step >= args.max_steps:
            break
        images = images.to(device, non_blocking=True)
        _current_step = step + 1
        _hook_values.clear()
        # the capture sits OUTSIDE the skip-batch handler: a failing capture must not be mistaken for a bad batch (it would
        # be retried and "skipped" on every batch while the run trains nothing); it is reported once and the loop goes eager
        if graphed is not None and graphed.get("step") is None and not graphed.get("failed") and images.size(0) == graphed["batch"]:
            try:
                graphed["step"] = GraphedTrainStep(net, optimizer, aggregator, args, images, dp=dp, preserve_state=True)
                graphed["hooks"] = dict(_hook_values)  # the weighting's forward hooks ran while capturing: static tensors
                if dp is not None and dp.rank == 0:
                    print(f"[movae] data-parallel step form: {graphed['step'].dp_form}", flush=True)
            except NotImplementedError as e:
                graphed["failed"] = True
                print(f"[movae] hipGraph capture not possible ({e}); running the eager step", flush=True)
            except RuntimeError as e:
                if dp is not None or getattr(args, "graph", "auto") == "on":
                    raise  # ranks must not diverge in form; --graph on asked for the graph
                graphed["failed"] = True
                print(f"[movae] hipGraph capture failed ({type(e).__name__}: {e}); running the eager step", flush=True)
                torch.cuda.synchronize()
        gs = graphed.get("step") if graphed is not None else None
        if gs is not None and images.shape == gs.static_x.shape:
            loss_dict, outputs = gs.step(images)
            _hook_values.update(graphed["hooks"])
        else:
            try:
                loss_dict, outputs = train_step(net, images, optimizer, aggregator, args, dp)
            except RuntimeError as e:  # main.py:197-208: skip the batch on device-side assertion errors
                recoverable = "cuda" in str(e).lower() or "hip" in str(e).lower() or "assert" in str(e).lower()
                if not recoverable or dp is not None:
                    # under data parallelism a rank that skipped alone would leave its peers waiting in the all-reduce
                    raise
                print(f"Step {step}: device error during backward: {e}\n  Skipping this batch...")
                continue
        host = _host_values(loss_dict)
        if host["total_loss"] > 1e15:
            print(f"Step {step}: EXPLODING: Total loss: {host['total_loss']:.6e}")
        if "codebook_usage_percentage" in outputs:
            usage.update(float(outputs["codebook_usage_percentage"]), n=images.size(0))  # LazyScalar: one device read
        for k in meters:
            meters[k].update(host[k])
        step += 1
        if log is not None:
            rec = {f"train/{k}": m.avg for k, m in meters.items()}
            rec.update({f"train/{k}_curr": m.val for k, m in meters.items()})
            if usage.count > 0:
                rec["train/codebook_usage_percentage"] = usage.avg
            for i, w in enumerate(host.get("_weights", [])):
                rec[f"train/task_{i}_weight"] = w
            if "_similarity" in host:
                rec["train/gradient_similarity"] = host["_similarity"]
            log(rec, step)
    if usage.count > 0:
        meters["codebook_usage_percentage"] = usage
    return meters, step


@torch.no_grad()
def evaluate(net, loader, device, args):
    """main.py:238-332 -- losses and codebook usage over a whole loader.  Same meters as the reference (one update per
    batch, un-weighted, `total_loss` included; codebook usage = distinct codes seen over ALL batches, the two VQ-VAE-2
    codebooks averaged), but nothing leaves the device inside the loop: the reference reads every loss with `.item()`
    and concatenates every batch's indices on the host; here each batch appends one stacked loss row and ORs its codes
    into a K-entry device mask, and the host reads both once at the end."""
    net.eval()
    meters = {k: AverageMeter() for k in net.objectives.keys()}
    meters["total_loss"] = AverageMeter()
    rows, keys = [], None
    used = {}  # output key -> bool[K] on the device

    def mark(name, inds, K):
        if name not in used:
            used[name] = torch.zeros(K, dtype=torch.bool, device=inds.device)
        used[name][inds.reshape(-1)] = True

    for images, _ in loader:
        images = images.to(device)
        out = net(images)
        ld = net.loss_function(images, args=out)
        if keys is None:
            keys = list(ld.keys())
        rows.append(torch.stack([ld[k].detach().float().reshape(()) for k in keys]))
        if out.get("encoding_inds") is not None and hasattr(net, "vq_layer"):
            mark("encoding_inds", out["encoding_inds"], net.vq_layer.K)
        elif out.get("encoding_inds_top") is not None and out.get("encoding_inds_bottom") is not None and hasattr(net, "vq_top"):
            mark("encoding_inds_top", out["encoding_inds_top"], net.vq_top.K)
            mark("encoding_inds_bottom", out["encoding_inds_bottom"], net.vq_top.K)  # main.py:296 takes K from vq_top for both
    if rows:
        for vals in torch.stack(rows).cpu().tolist():
            for k, v in zip(keys, vals):
                meters[k].update(v)
    if used:
        usage = AverageMeter()
        usage.update(sum(float(m.sum().item()) / m.numel() * 100.0 for m in used.values()) / len(used))
        meters["codebook_usage_percentage"] = usage
    return meters


# ---- data ----------------------------------------------------------------------------------------
SYNTHETIC = {"synthetic_cifar10": (32, 50000), "synthetic_celeba": (64, 162770), "synthetic_celeba_hq": (256, 30000),
             "synthetic_imagenet": (256, 1281167)}


class SyntheticImages(torch.utils.data.Dataset):
    """Seeded uniform [0,1) images (the un-normalised ToTensor range, utils/utils.py:187-192).  Item i is entry
    i mod POOL of a pool drawn once from the seed: the loader then costs a view per item instead of a generator per item
    (the per-item generator held a 32x32 step at 38 ms; the device needs 1.2 ms)."""

    POOL = 4096

    def __init__(self, n, size, seed, normalize=False):
        self.n, self.size, self.seed, self.normalize = n, size, seed, normalize
        pool = min(n, max(64, min(self.POOL, (256 << 20) // (12 * size * size))))  # <= 256 MB of host memory
        x = torch.rand(pool, 3, size, size, generator=torch.Generator().manual_seed(seed))
        self.pool = 2 * x - 1 if normalize else x

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return self.pool[i % self.pool.size(0)], 0


def get_dataset(name, data_dir="./data", normalize=False, max_items=None):
    """utils/utils.py:144-426 for real datasets needs torchvision / HF `datasets` downloads, which are
    unavailable offline; synthetic_* names give shape-compatible data."""
    key = name.lower()
    if key in SYNTHETIC:
        size, n = SYNTHETIC[key]
        n = min(n, max_items) if max_items else n
        return SyntheticImages(n, size, 0, normalize), SyntheticImages(max(1, n // 10), size, 1, normalize), size
    raise NotImplementedError(
        f"dataset {name!r}: real datasets are fetched over the network by the reference (torchvision / HF hub) and are "
        f"outside the hot path; use one of {sorted(SYNTHETIC)}")


# ---- CLI -----------------------------------------------------------------------------------------
#: main.py:1603-1623,1636-1640 -- options that only the out-of-scope architectures read
IGNORED_FLAGS = {
    "--recursive_kld_anneal_steps": dict(type=int, default=25000),
    "--sigma_max_angle_deg": dict(type=float, default=80.0),
    "--sigma_mix_prob": dict(type=float, default=0.0),
    "--sigma_mix_angle_min_deg": dict(type=float, default=None),
    "--sigma_mix_angle_max_deg": dict(type=float, default=None),
    "--lambda_pix_recon": dict(type=float, default=1.0),
    "--lambda_pix_con": dict(type=float, default=0.5),
    "--lambda_lat_con": dict(type=float, default=0.1),
    "--patch_size": dict(type=int, default=None),
    "--vit_embed_dim": dict(type=int, default=1024),
    "--vit_depth": dict(type=int, default=24),
    "--vit_num_heads": dict(type=int, default=16),
    "--vit_mixer_depth": dict(type=int, default=2),
    "--num_classes": dict(type=int, default=0),
    "--pixelsnail_num_blocks": dict(type=int, default=8),
    "--pixelsnail_num_res_blocks": dict(type=int, default=2),
    "--pixelsnail_num_heads": dict(type=int, default=8),
    "--pixelsnail_dropout": dict(type=float, default=0.1),
}


def build_parser():
    """Flag names, aliases and defaults of main.py:1500-1651 (hot-path subset keeps every name)."""
    p = ArgumentParser()
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--device", type=str, default="cuda:0" if torch.cuda.is_available() else "cpu")
    p.add_argument("--data_dir", type=str, default="./data")
    p.add_argument("--save_path", type=str, default="logs/")
    p.add_argument("--epochs", type=int, default=50)
    p.add_argument("--dataset", type=str, default="CIFAR10")
    p.add_argument("--normalize_inputs", action="store_true", dest="normalize_inputs")
    p.add_argument("--batch_size", type=int, default=128)
    p.add_argument("--num_workers", type=int, default=0)
    p.add_argument("--aggregator", "--agg", type=str, default=None)
    p.add_argument("--agg_norm_eps", "--agg-norm-eps", "--norm_eps", "--norm-eps", type=float, default=1e-4)
    p.add_argument("--agg_reg_eps", "--agg-reg-eps", "--reg_eps", "--reg-eps", type=float, default=1e-4)
    p.add_argument("--mgda_epsilon", "--mgda-epsilon", type=float, default=1e-5)
    p.add_argument("--mgda_max_iters", "--mgda-max-iters", type=int, default=250)
    p.add_argument("--mgda_min_eigenvalue_eps", "--mgda-min-eigenvalue-eps", type=float, default=1e-10)
    p.add_argument("--comfort_mgda_norm_type", "--comfort-mgda-norm-type", type=str, default="none", choices=["none", "l2", "loss", "loss+"])
    p.add_argument("--comfort_mgda_stable", "--comfort-mgda-stable", action="store_true")
    p.add_argument("--comfort_beta_k", type=float, default=1.0)
    p.add_argument("--comfort_beta_a", type=float, default=1.0)
    p.add_argument("--comfort_beta_l", type=float, default=0.01)
    p.add_argument("--comfort_beta_u", type=float, default=1.0)
    p.add_argument("--arch", type=str, default="vae")
    p.add_argument("--layer_norm", type=str, default="batch")
    p.add_argument("--latent_dim", type=int, default=128)
    p.add_argument("--hidden_dims", type=int, nargs="+", default=[32, 64, 128, 256, 512])
    p.add_argument("--num_residual_layers", type=int, default=2)
    p.add_argument("--recons_objective", type=str, default="mse", choices=["mse", "bce", "l1", "smooth_l1", "perceptual"])
    p.add_argument("--recons_activation", type=str, default=None, choices=["tanh", "sigmoid", "none"])
    p.add_argument("--loss_weights", type=str, nargs="*", default=None)
    p.add_argument("--pref_weights", type=str, nargs="*", default=None)
    p.add_argument("--optimizer", type=str, default="adam")
    p.add_argument("--momentum", type=float, default=0.9)
    p.add_argument("--max_grad_norm", type=float, default=None)
    p.add_argument("--lr", type=float, default=0.001)
    p.add_argument("--wd", "--weight_decay", type=float, default=0)
    p.add_argument("--scheduler", type=str, default=None)
    p.add_argument("--scheduler_lr_min", type=float, default=0.0)
    p.add_argument("--scheduler_gamma", type=float, default=0.1)
    p.add_argument("--scheduler_milestones", type=int, nargs="+", default=None)
    p.add_argument("--embedding_dim", type=int, default=None)
    p.add_argument("--num_embeddings", type=int, default=None)
    p.add_argument("--anneal_steps", type=int, default=None)
    p.add_argument("--hv_ref", type=str, nargs="*", default=None)
    p.add_argument("--num_vis_samples", type=int, default=4, dest="num_vis_samples")
    p.add_argument("--save_freq", type=int, default=10)
    p.add_argument("--eval_freq", type=int, default=1)
    p.add_argument("--use_wandb", action="store_true")
    p.add_argument("--wandb_project", type=str, default="mo-vae")
    p.add_argument("--wandb_entity", type=str, default=None)
    p.add_argument("--wandb_name", type=str, default=None)
    p.add_argument("--wandb_group", type=str, default=None)
    p.add_argument("--wandb_tags", type=str, nargs="+", default=None)
    p.add_argument("--max_fid_samples", type=int, default=10000)
    p.add_argument("--max_gen_metrics_samples", type=int, default=10000)
    # PixelCNN prior over the VQ codes (main.py:1625-1651; prior.py).  PixelSNAIL is outside SURVEY 8f.4's row: refused by name.
    p.add_argument("--prior_type", type=str, default="pixelcnn", choices=["pixelcnn", "pixelsnail"])
    p.add_argument("--skip_pixelcnn", action="store_true")
    p.add_argument("--pixelcnn_epochs", type=int, default=100)
    p.add_argument("--pixelcnn_hidden_channels", type=int, default=128)
    p.add_argument("--pixelcnn_num_layers", type=int, default=15)
    p.add_argument("--pixelcnn_lr", type=float, default=3e-4)
    p.add_argument("--pixelcnn_temperature", type=float, default=1.0)
    p.add_argument("--prior_use_lmdb_codes", action="store_true", default=True)
    p.add_argument("--no_prior_lmdb_codes", action="store_false", dest="prior_use_lmdb_codes")
    p.add_argument("--prior_force_extract_codes", action="store_true")
    p.add_argument("--prior_lmdb_map_size_gb", type=float, default=150)
    # flags of architectures outside the hot path (recursive / sphere / ViT models, PixelSNAIL): accepted with the reference's
    # types and defaults so that any reference YAML -> runner.py argv parses, ignored with one warning when given
    for name, kw in IGNORED_FLAGS.items():
        p.add_argument(name, **kw)
    # additions of this build (not in the reference)
    p.add_argument("--graph", choices=["auto", "off", "on"], default="auto",
                   help="auto / on: capture the step into a hipGraph on the first full batch and replay it (device-bound "
                        "instead of host-bound; the capture's warm-up steps are rewound, so the run makes the eager loop's "
                        "steps -- only random draws inside the step come from the graph's own Philox offsets); auto falls back to the eager step silently where replay is not possible, on says so")
    p.add_argument("--dtype", choices=["fp32", "bf16"], default="fp32",
                   help="fp32 (default): the reference's arithmetic, the parity path.  bf16 (opt-in): bf16 operands / fp32 "
                        "accumulation in the 128x128 implicit-GEMM convolution kernels (the MFMA-bound layers of the 64x64+ "
                        "configurations); master weights, BatchNorm, losses, aggregation and optimizer stay fp32")
    p.add_argument("--max_items", type=int, default=None, help="cap the synthetic dataset length")
    p.add_argument("--max_steps", type=int, default=None, help="stop after this many optimisation steps")
    return p


def parse_args(argv=None):
    import sys

    parser = build_parser()
    args = parser.parse_args(argv)
    given = [a.split("=", 1)[0] for a in (sys.argv[1:] if argv is None else argv) if isinstance(a, str) and a.startswith("--")]
    ignored = sorted(set(given) & set(IGNORED_FLAGS))
    if ignored:
        print(f"[movae] ignoring options of architectures outside the MI355X hot path: {', '.join(ignored)}", flush=True)
    if args.loss_weights is not None and len(args.loss_weights) > 0:  # main.py:1655-1659
        if len(args.loss_weights) == 1 and args.loss_weights[0].strip().startswith("{"):
            args.loss_weights = json.loads(args.loss_weights[0])
        else:
            args.loss_weights = [float(x) for x in args.loss_weights]
    if args.hv_ref is not None and len(args.hv_ref) > 0:
        if len(args.hv_ref) == 1 and args.hv_ref[0].strip().startswith("{"):
            args.hv_ref = {k: float(v) for k, v in json.loads(args.hv_ref[0]).items()}
        else:
            args.hv_ref = [float(x) for x in args.hv_ref]
    return args


def make_optimizer(net, args, capturable=False):
    """main.py:1169-1178.  capturable=True keeps Adam's step counter on the device (hipGraph replay)."""
    if args.optimizer == "sgd":
        return optim.SGD(net.parameters(), lr=args.lr, momentum=args.momentum, weight_decay=args.wd)
    if args.optimizer == "adam":  # same arithmetic and state_dict as torch.optim.Adam, one launch per step (optim.py)
        return FusedAdam(net.parameters(), lr=args.lr, weight_decay=args.wd, device_step=capturable)
    if args.optimizer == "adamw":
        return FusedAdamW(net.parameters(), lr=args.lr, weight_decay=args.wd, device_step=capturable)
    if args.optimizer == "rmsprop":
        return optim.RMSprop(net.parameters(), lr=args.lr, weight_decay=args.wd)
    raise ValueError(f"Optimizer {args.optimizer} not supported")


def make_scheduler(optimizer, args):
    """main.py:1180-1189."""
    if args.scheduler is None:
        return None
    if args.scheduler == "cosine":
        return optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=args.epochs, eta_min=args.scheduler_lr_min)
    if args.scheduler == "multi_step":
        return optim.lr_scheduler.MultiStepLR(optimizer, milestones=args.scheduler_milestones, gamma=args.scheduler_gamma)
    if args.scheduler == "exponential":
        return optim.lr_scheduler.ExponentialLR(optimizer, gamma=args.scheduler_gamma)
    raise ValueError(f"Scheduler {args.scheduler} not supported")


def main(args):
    dp = DataParallelGrads.from_env()
    if dp is not None:
        args.device = f"cuda:{dp.local_rank}"
    device = torch.device(args.device)
    if device.type != "cuda":
        raise RuntimeError("the MI355X hot path needs a HIP device (--device cuda:N); there is no CPU path")
    torch.cuda.set_device(device)
    L.set_compute_dtype(getattr(args, "dtype", "fp32"))
    train_ds, test_ds, input_size = get_dataset(args.dataset, data_dir=args.data_dir, normalize=args.normalize_inputs,
                                                max_items=args.max_items)
    per_rank_bs = args.batch_size if dp is None else max(1, args.batch_size // dp.world_size)
    sampler = None
    if dp is not None:
        sampler = torch.utils.data.distributed.DistributedSampler(train_ds, num_replicas=dp.world_size, rank=dp.rank,
                                                                  shuffle=True, seed=args.seed or 0)
    # pinning in the main thread (num_workers == 0) allocates and frees page-locked memory per batch: ~20 ms per batch on ROCm
    loader_kw = dict(num_workers=args.num_workers, pin_memory=args.num_workers > 0, drop_last=False,
                     persistent_workers=args.num_workers > 0)
    train_loader = torch.utils.data.DataLoader(train_ds, batch_size=per_rank_bs, shuffle=sampler is None, sampler=sampler, **loader_kw)
    test_loader = torch.utils.data.DataLoader(test_ds, batch_size=per_rank_bs, shuffle=False, **loader_kw)
    args.dataset_size = len(train_ds)
    net = get_network(input_size, num_channels=3, args=args, device=device).to(device)
    args.total_params = net.total_trainable_params()
    for name, w in net.lambda_weights.items():
        setattr(args, f"{name}_weight", w)
    if dp is not None:
        dp.attach(net)
    aggregator = aggregation.make_aggregator(args)
    graphed = None
    if getattr(args, "graph", "off") in ("on", "auto"):
        # PNUPGrad / PCGrad draw torch's CPU generator per call, COMFORT's blend factor is a host scalar that changes per epoch
        host_rng = isinstance(aggregator, (aggregation.PNUPGrad, aggregation.PCGrad, aggregation.COMFORT))
        if getattr(net, "graph_safe", False) and not host_rng and args.optimizer.lower() in ("adam", "adamw"):
            graphed = {"batch": per_rank_bs, "step": None}
        elif args.graph == "on":
            print("--graph on: this model / aggregator / optimizer combination is not replayable; running the eager step")
    optimizer = make_optimizer(net, args, capturable=graphed is not None)
    scheduler = make_scheduler(optimizer, args)
    if aggregator is not None and aggregator != "sum":
        aggregator.weighting.register_forward_hook(print_weights)
        aggregator.weighting.register_forward_hook(print_gd_similarity)
    rank0 = dp is None or dp.rank == 0
    stamp = time.strftime("%Y%m%d_%H%M%S")
    save_root = os.path.join(args.save_path, args.dataset, args.arch, args.optimizer, args.aggregator, stamp)
    if rank0:
        os.makedirs(os.path.join(save_root, "checkpoints"), exist_ok=True)
    log = None
    if args.use_wandb:
        try:
            import wandb

            wandb.init(project=args.wandb_project, entity=args.wandb_entity, name=args.wandb_name, config=vars(args),
                       dir=save_root, group=args.wandb_group, tags=args.wandb_tags)
            log = lambda rec, step: wandb.log(rec, step=step)  # noqa: E731
        except ImportError:
            print("wandb is not installed; continuing without it")
    if hasattr(net, "print_model_summary") and args.device.endswith("0") and rank0:
        net.print_model_summary()
    step, history, best = 0, [], float("inf")
    eval_rec = {}
    for epoch in range(1, args.epochs + 1):
        if sampler is not None:
            sampler.set_epoch(epoch)
        if isinstance(aggregator, COMFORT):  # main.py:1290-1291
            aggregator.set_epoch(epoch, args.epochs)
        t0 = time.time()
        step0 = step
        meters, step = train_epoch(net, train_loader, optimizer, aggregator, step, device, args, dp, log, graphed)
        torch.cuda.synchronize()
        dt = time.time() - t0
        rec = {k: m.avg for k, m in meters.items()}
        history.append(rec)
        if rank0:
            n_img = (step - step0) * per_rank_bs * (1 if dp is None else dp.world_size)
            print(f"epoch {epoch}: " + ", ".join(f"{k}: {v:.6e}" for k, v in rec.items()) + f"  [{n_img / dt:.0f} img/s]")
        if args.eval_freq and epoch % args.eval_freq == 0:
            ev = evaluate(net, test_loader, device, args)
            eval_rec = {k: m.avg for k, m in ev.items()}
            best = min(best, ev["total_loss"].avg)
            if rank0:
                print(f"  eval: " + ", ".join(f"{k}: {m.avg:.6e}" for k, m in ev.items()))
        if scheduler is not None:
            scheduler.step()
        if args.max_steps is not None and step >= args.max_steps:
            break
    if rank0:  # main.py:1422-1436: save-only checkpoint with the reference's keys and value shapes
        ckpt = {"epoch": args.epochs, "model_state_dict": {k: v.contiguous() for k, v in net.state_dict().items()},
                "args": vars(args), "train_losses": dict(history[-1]) if history else {}, "eval_losses": dict(eval_rec),
                "best_eval_loss": best,
                "train_history": history}  # addition of this build: every epoch's averages (the reference keeps the last only)
        if scheduler is not None:
            ckpt["scheduler_state_dict"] = scheduler.state_dict()
        torch.save(ckpt, os.path.join(save_root, "checkpoints", "final_checkpoint.pth"))
    if rank0 and _is_vq_arch(args.arch) and not getattr(args, "skip_pixelcnn", False):  # main.py:1438-1497
        from . import prior as _prior

        # The prior sees the WHOLE training set at the CLI batch size, like the single-process reference: under data parallelism
        # the training loader is a DistributedSampler shard at the per-rank batch size, so rank 0 builds a plain loader of its own.
        prior_loader = train_loader
        if dp is not None:
            prior_loader = torch.utils.data.DataLoader(train_ds, batch_size=args.batch_size, shuffle=True, **loader_kw)
        _prior.train_pixelcnn_prior(net, prior_loader, device, args, save_root)
    if dp is not None:
        dp.shutdown()
    return history


def _is_vq_arch(arch):
    a = (arch or "").lower()
    return a in ("vq_vae", "vq_vae2", "gg_vq_vae2") or a.startswith("gg_vq_vae")


def cli(argv=None):
    args = parse_args(argv)
    if args.seed is not None:
        set_seed(args.seed)
    return main(args)
