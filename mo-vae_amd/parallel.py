"""Data-parallel step exchange: one process per GPU, the minibatch is sharded,
every rank runs its K per-loss backward passes and its own aggregation locally, and only the
AGGREGATED gradient crosses xGMI -- a single all-reduce of one flat fp32 bucket per step (RCCL,
`backend="nccl"`; gloo for the CPU rehearsal in tests/).  The reference has no multi-device code.

Semantics (documented limitation): aggregation is non-linear in J and BatchNorm statistics are
per-rank, so N ranks reproduce "the mean over shards of the single-device result on each shard",
not the single-device large-batch step.
"""
import os

import torch
import torch.distributed as dist


def flatten_grads(params, out=None):
    """All gradients in one contiguous buffer; params without grad contribute zeros.  Every chunk is laid out in the
    PARAMETER's memory order (channels_last conv weights: [o][kh][kw][i]) because unflatten_into_grads re-views the bucket
    with the parameter's strides: a gradient whose own strides differ (a contiguous gradient of a channels_last weight,
    or the reverse) is first copied into the parameter's layout instead of being packed as it lies."""
    chunks = []
    for p in params:
        if p.grad is None:
            chunks.append(torch.zeros(p.numel(), dtype=p.dtype, device=p.device))
            continue
        g = p.grad
        if g.shape != p.shape:
            raise RuntimeError(f"gradient shape {tuple(g.shape)} does not match its parameter {tuple(p.shape)}")
        if any(n > 1 and a != b for n, a, b in zip(p.shape, g.stride(), p.stride())):  # (strides of size-1 dims carry no layout)
            g = torch.empty_like(p).copy_(g)  # empty_like preserves the parameter's (dense) strides
        if g.dim() == 4 and not g.is_contiguous() and g.permute(0, 2, 3, 1).is_contiguous():
            chunks.append(g.permute(0, 2, 3, 1).reshape(-1))
        else:
            chunks.append(g.contiguous().reshape(-1))
    if out is not None:
        return torch.cat(chunks, out=out)
    return torch.cat(chunks)


def unflatten_into_grads(flat, params):
    off = 0
    for p in params:
        n = p.numel()
        p.grad = flat[off: off + n].as_strided(p.shape, p.stride())  # params are dense (contiguous or channels_last)
        off += n


class DataParallelGrads:
    def __init__(self, rank, world_size, local_rank, backend):
        self.rank, self.world_size, self.local_rank, self.backend = rank, world_size, local_rank, backend
        self.params = []
        self.flat, self.views, self._arena_sig = None, None, None  # the eager path's persistent flat arena and one view per parameter

    def _arena(self):
        """One flat fp32 buffer for the whole gradient, allocated once; views[i] has parameter i's shape AND strides (conv
        weights are channels_last), so a gradient copied into it lies in the parameter's memory order and `.grad = views[i]`
        afterwards needs no unflatten pass."""
        if self.flat is None or self.flat.device != self.params[0].device or self._arena_sig != [(id(p), p.numel()) for p in self.params]:
            n = sum(p.numel() for p in self.params)
            self.flat = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)
            self.views, off = [], 0
            for p in self.params:
                self.views.append(self.flat[off: off + p.numel()].as_strided(p.shape, p.stride()))
                off += p.numel()
            self._arena_sig = [(id(p), p.numel()) for p in self.params]
        return self.flat, self.views

    @classmethod
    def from_env(cls, backend=None):
        """torchrun / torch.distributed.run environment -> initialised process group, or None."""
        ws = int(os.environ.get("WORLD_SIZE", "1"))
        if ws <= 1 and not os.environ.get("MOVAE_FORCE_DP"):  # MOVAE_FORCE_DP: exercise the collective path with one rank
            return None
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("MASTER_PORT", "29533")
        rank, local_rank = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))
        # MOVAE_DIST_BACKEND=gloo rehearses the N>1 path on a box with fewer GPUs than ranks (RCCL refuses two
        # ranks on one device); the device index then wraps around the visible devices.
        backend = backend or os.environ.get("MOVAE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            local_rank = local_rank % torch.cuda.device_count()
            torch.cuda.set_device(local_rank)
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend=backend, rank=rank, world_size=ws)
        return cls(rank, ws, local_rank, backend)

    def attach(self, net):
        """Broadcast rank 0's parameters / buffers so every replica starts identical."""
        self.params = [p for p in net.parameters() if p.requires_grad]
        for t in list(net.parameters()) + list(net.buffers()):
            dist.broadcast(t.data, src=0)

    @torch.no_grad()
    def all_reduce_grads(self):
        """mean over ranks of the full flat gradient: ONE collective per step, on a PERSISTENT arena -- no per-step
        torch.cat into a fresh 13-68 MB tensor: the step's gradients are copied into the arena's views by one multi-tensor
        copy, RCCL averages in the collective (ncclAvg; gloo, the CPU rehearsal, has no AVG: sum, then scale), and `.grad`
        becomes the arena's views (the optimizer reads the bucket in place)."""
        flat, views = self._arena()
        src, dst = [], []
        for p, v in zip(self.params, views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                if p.grad.shape != p.shape:
                    raise RuntimeError(f"gradient shape {tuple(p.grad.shape)} does not match its parameter {tuple(p.shape)}")
                src.append(p.grad)
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
        if self.backend == "nccl":
            dist.all_reduce(flat, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.div_(self.world_size)
        for p, v in zip(self.params, views):
            p.grad = v
        return flat

    def barrier(self):
        dist.barrier()

    def shutdown(self):
        if dist.is_initialized():
            dist.destroy_process_group()
