"""World-size-2 rehearsal of the data-parallel exchange on CPU tensors over gloo: one flat
all-reduce of the full gradient per step, mean over ranks, channels_last gradients preserved, replicas
identical after broadcast."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import movae_amd  # noqa: F401
    from movae_amd.parallel import DataParallelGrads, flatten_grads

    dp = DataParallelGrads.from_env(backend="gloo")
    assert dp is not None and dp.world_size == world and dp.rank == rank
    torch.manual_seed(100 + rank)  # different initial replicas on purpose
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Linear(5, 2))
    net[0].weight.data = net[0].weight.data.contiguous(memory_format=torch.channels_last)
    dp.attach(net)
    sd = torch.cat([p.detach().reshape(-1) for p in net.parameters()] + [b.detach().float().reshape(-1) for b in net.buffers()])
    # per-rank gradients: rank r contributes (r + 1) * pattern; one parameter has no gradient on rank 1
    for i, p in enumerate(net.parameters()):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    if rank == 1:
        list(net.parameters())[3].grad = None
    calls_before = getattr(dist, "_movae_calls", 0)
    flat = dp.all_reduce_grads()
    grads = [p.grad.clone() for p in net.parameters()]
    strides_ok = net[0].weight.grad.stride() == net[0].weight.stride()
    # every .grad is a view into the single bucket
    base_ok = all(p.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for p in net.parameters())
    ret[rank] = dict(sd=sd, grads=grads, strides_ok=strides_ok, base_ok=base_ok, n=flat.numel(),
                     flat_ok=torch.equal(flatten_grads(list(net.parameters())), flat), calls=calls_before)
    dp.barrier()
    dp.shutdown()


def test_flat_bucket_allreduce_world2():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        r0, r1 = ret[0], ret[1]
        assert torch.equal(r0["sd"], r1["sd"]), "replicas differ after broadcast"
        for i, (g0, g1) in enumerate(zip(r0["grads"], r1["grads"])):
            assert torch.equal(g0, g1), f"gradient {i} differs across ranks"
            want = (1 + 2) / 2 * (i + 1) if i != 3 else (1 * (i + 1) + 0) / 2
            assert torch.allclose(g0, torch.full_like(g0, want)), (i, g0.flatten()[:3], want)
        assert r0["strides_ok"] and r1["strides_ok"] and r0["base_ok"] and r0["flat_ok"]
        assert r0["n"] == sum(g.numel() for g in r0["grads"])


def _shard_worker(rank, world, port, ret):
    """One rank of the N > 1 parity definition: the rank's own aggregated gradient (here produced by the CPU
    oracle on the rank's shard -- no GPU in this test) goes through DataParallelGrads.all_reduce_grads; a gradient whose
    strides differ from its parameter's (contiguous gradient of a channels_last weight) must not be permuted."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import movae_amd  # noqa: F401
    from movae_amd.parallel import DataParallelGrads
    from oracle import nets
    from oracle.step import OracleTrainer

    torch.set_num_threads(2)
    dp = DataParallelGrads.from_env(backend="gloo")
    cfg = nets.make_cfg("vae", 16, 8, 1000, latent_dim=8, hidden_dims=[8, 16])
    tr = OracleTrainer(cfg, seed=3, agg="upgrad")
    g = torch.Generator().manual_seed(9)
    x, eps = torch.rand(8, 3, 16, 16, generator=g), torch.randn(8, 8, generator=g)
    sl = slice(rank * 4, rank * 4 + 4)
    _, _, grads, _ = tr.grads(x[sl], eps[sl])
    holder = torch.nn.ParameterList()
    for n, p in tr.params.items():
        q = torch.nn.Parameter(p.detach().clone())
        if q.dim() == 4:
            q.data = q.data.contiguous(memory_format=torch.channels_last)  # the HIP models keep conv weights channels_last
        q.grad = grads[n].detach().clone().contiguous()  # logical order, contiguous: strides differ from the parameter's
        holder.append(q)
    dp.params = list(holder)
    dp.all_reduce_grads()
    ret[rank] = {n: q.grad.detach().clone().contiguous() for n, q in zip(tr.params, holder)}
    ret[f"own{rank}"] = {n: grads[n].detach().clone() for n in tr.params}
    ret[f"strides{rank}"] = all(q.grad.stride() == q.stride() for q in holder)
    dp.barrier()
    dp.shutdown()


def test_two_rank_step_is_mean_over_shard_gradients():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_shard_worker, args=(world, port, ret), nprocs=world, join=True)
        assert ret["strides0"] and ret["strides1"]
        for n in ret[0]:
            want = (ret["own0"][n] + ret["own1"][n]) / 2
            assert torch.equal(ret[0][n], ret[1][n]), n
            assert torch.allclose(ret[0][n], want, rtol=1e-6, atol=1e-9), n
