#!/usr/bin/env python3
"""One rank of the two-rank data-parallel semantics test (tests/test_dp_two_ranks_gpu.py starts two of these as fresh child
processes BEFORE the parent test process has touched the GPU).  gloo backend, both ranks on cuda:0 (RCCL refuses two ranks on
one device; the exchange code path -- flatten, all-reduce, unflatten, clip, optimizer -- is the same).

Writes <out>/rank<r>.npz: parameters after (a) one eager data-parallel step, (b) one replayed hipGraph data-parallel step
(graph | all-reduce | graph, the gloo form) from the same initial state, (c) a following ragged batch through the eager path.
Not a test module and not product code: it only drives mo-vae_amd's public API."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Args:
    def __init__(self, **kw):
        self.__dict__.update(kw)


CFG = dict(arch="vae", input_size=32, latent_dim=16, hidden_dims=[16, 32, 64], dataset_size=1000, global_batch=16, ragged=5)
#: the overlapped two-bucket form (MOVAE_DP_OVERLAP=1: graph 1 | all-reduce(task-side bucket) under graph 1b | all-reduce(shared
#: bucket) | graph 2) on a model whose task-side / shared split is not trivial: the VQ-VAE's codebook and decoder are task-side
#: (early bucket), the encoder is shared (late bucket)
CFG_VQ = dict(arch="vq_vae", input_size=16, embedding_dim=8, num_embeddings=16, hidden_dims=[8, 16], num_residual_layers=2,
              dataset_size=1000, global_batch=8, agg="aligned_mtl")


def make_inputs_vq():
    import torch

    g = torch.Generator().manual_seed(91)
    return torch.rand(CFG_VQ["global_batch"], 3, CFG_VQ["input_size"], CFG_VQ["input_size"], generator=g)


def make_inputs():
    import torch

    B, D = CFG["global_batch"], CFG["latent_dim"]
    g = torch.Generator().manual_seed(77)
    x = torch.rand(B, 3, CFG["input_size"], CFG["input_size"], generator=g)
    eps = torch.randn(B, D, generator=g)
    x2 = torch.rand(2 * CFG["ragged"], 3, CFG["input_size"], CFG["input_size"], generator=g)
    eps2 = torch.randn(2 * CFG["ragged"], D, generator=g)
    return x, eps, x2, eps2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--out", type=str, required=True)
    ap.add_argument("--agg", type=str, default="upgrad")
    o = ap.parse_args()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(o.port), RANK=str(o.rank), WORLD_SIZE=str(o.world),
                      LOCAL_RANK=str(o.rank), MOVAE_DIST_BACKEND="gloo", MOVAE_NO_REBUILD="1")
    import numpy as np
    import torch

    import movae_amd  # noqa: F401
    from movae_amd import aggregation
    from movae_amd.models import get_network
    from movae_amd.parallel import DataParallelGrads
    from movae_amd.train import GraphedTrainStep, make_optimizer, train_step

    dev = torch.device("cuda:0")
    dp = DataParallelGrads.from_env(backend="gloo")
    assert dp is not None and dp.world_size == o.world and dp.rank == o.rank
    x, eps, x2, eps2 = make_inputs()
    per = CFG["global_batch"] // o.world
    sl = slice(o.rank * per, (o.rank + 1) * per)
    rg = slice(o.rank * CFG["ragged"], (o.rank + 1) * CFG["ragged"])

    def make(capturable, clip):
        a = Args(arch=CFG["arch"], batch_size=CFG["global_batch"], dataset_size=CFG["dataset_size"], recons_objective="mse",
                 recons_activation=None, loss_weights=None, agg_norm_eps=1e-4, agg_reg_eps=1e-4, mgda_epsilon=1e-5,
                 mgda_max_iters=250, pref_weights=None, optimizer="adam", lr=1e-3, wd=0, momentum=0.9, latent_dim=CFG["latent_dim"],
                 hidden_dims=CFG["hidden_dims"], aggregator=o.agg, max_grad_norm=clip)
        torch.manual_seed(5 + 100 * o.rank)  # replicas start DIFFERENT on purpose: attach() must make them rank 0's
        net = get_network(CFG["input_size"], 3, a, dev).to(dev).train()
        dp.attach(net)
        return net, a, make_optimizer(net, a, capturable=capturable), aggregation.make_aggregator(a)

    res = {}
    # (a) eager data-parallel step
    net, a, opt, agg = make(False, None)
    res["init"] = {n: p.detach().cpu().numpy().copy() for n, p in net.named_parameters()}
    net.eps_override = eps[sl].to(dev)
    train_step(net, x[sl].to(dev), opt, agg, a, dp)
    torch.cuda.synchronize()
    res["eager"] = {n: p.detach().cpu().numpy().copy() for n, p in net.named_parameters()}
    # (b) the replayed form from the same initial state, then (c) a ragged batch through the eager path
    net, a, opt, agg = make(True, None)
    static_eps = eps[sl].to(dev).clone()
    net.eps_override = static_eps
    gs = GraphedTrainStep(net, opt, agg, a, x[sl].to(dev), dp=dp, preserve_state=True)
    gs.step(x[sl].to(dev))
    torch.cuda.synchronize()
    res["graph"] = {n: p.detach().cpu().numpy().copy() for n, p in net.named_parameters()}
    res["graph_form"] = gs.dp_form
    net.eps_override = eps2[rg].to(dev)
    train_step(net, x2[rg].to(dev), opt, agg, a, dp)
    torch.cuda.synchronize()
    res["ragged"] = {n: p.detach().cpu().numpy().copy() for n, p in net.named_parameters()}
    # (d) the overlapped two-bucket form, VQ-VAE with Aligned-MTL: one replayed step from rank 0's init
    xv = make_inputs_vq()
    perv = CFG_VQ["global_batch"] // o.world
    av = Args(arch=CFG_VQ["arch"], batch_size=CFG_VQ["global_batch"], dataset_size=CFG_VQ["dataset_size"], recons_objective="mse",
              recons_activation=None, loss_weights=None, agg_norm_eps=1e-4, agg_reg_eps=1e-4, mgda_epsilon=1e-5, mgda_max_iters=250,
              pref_weights=None, optimizer="adam", lr=1e-3, wd=0, momentum=0.9, embedding_dim=CFG_VQ["embedding_dim"],
              num_embeddings=CFG_VQ["num_embeddings"], hidden_dims=CFG_VQ["hidden_dims"],
              num_residual_layers=CFG_VQ["num_residual_layers"], aggregator=CFG_VQ["agg"], max_grad_norm=None)
    torch.manual_seed(9 + 100 * o.rank)
    netv = get_network(CFG_VQ["input_size"], 3, av, dev).to(dev).train()
    dp.attach(netv)
    res["vq_init"] = {n: p.detach().cpu().numpy().copy() for n, p in netv.named_parameters()}
    os.environ["MOVAE_DP_OVERLAP"] = "1"
    try:
        gv = GraphedTrainStep(netv, make_optimizer(netv, av, capturable=True), aggregation.make_aggregator(av), av,
                              xv[o.rank * perv: (o.rank + 1) * perv].to(dev), dp=dp, preserve_state=True)
    finally:
        del os.environ["MOVAE_DP_OVERLAP"]
    gv.step(xv[o.rank * perv: (o.rank + 1) * perv].to(dev))
    torch.cuda.synchronize()
    res["vq_overlap"] = {n: p.detach().cpu().numpy().copy() for n, p in netv.named_parameters()}
    flat = {}
    for case in ("init", "eager", "graph", "ragged", "vq_init", "vq_overlap"):
        for n, v in res[case].items():
            flat[f"{case}/{n}"] = v
    flat["graph_form"] = np.array(res["graph_form"])
    flat["vq_form"] = np.array(gv.dp_form)
    flat["vq_buckets"] = np.array([gv.flat_a.numel(), gv.flat_b.numel() if gv.flat_b is not None else 0])
    np.savez(os.path.join(o.out, f"rank{o.rank}.npz"), **flat)
    dp.barrier()
    dp.shutdown()
    print(f"rank {o.rank}: done ({res['graph_form']})", flush=True)


if __name__ == "__main__":
    main()
