import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import movae_amd
from movae_amd import _lib as L, aggregation, autojac, ops
from movae_amd.models import get_network
from conftest import cfg_from_meta
from test_hip_models import Args, _full_case

dev = torch.device("cuda:0")
tag, batch = sys.argv[1], int(sys.argv[2])
fx, m = _full_case(tag)
c = cfg_from_meta(m)
size = int(m["input_size"])
for on in [bool(int(ch)) for ch in sys.argv[3]]:
    ops.DEFER_REDUCE = on
    args = Args(arch=c["arch"], batch_size=batch, dataset_size=c["dataset_size"], recons_objective="mse", recons_activation=None,
                loss_weights=None, **{k: v for k, v in c.items() if k in ("latent_dim", "hidden_dims", "embedding_dim", "num_embeddings", "num_residual_layers", "anneal_steps")})
    torch.manual_seed(3)
    net = get_network(size, num_channels=3, args=args, device=dev).to(dev).train()
    x = torch.rand(batch, 3, size, size, generator=torch.Generator().manual_seed(4)).to(dev)
    out = net(x)
    ld = net.loss_function(x, args=out)
    comp = [v for k, v in ld.items() if k != "total_loss"]
    use_hook = len(sys.argv) > 4
    hook = {}
    agg = aggregation.UPGrad()
    if use_hook:
        agg.register_forward_hook(lambda mod, inp, w: hook.update(J=inp[0].detach().norm(dim=1).cpu().numpy()))
    autojac.mtl_backward(losses=comp, features=[out[f] for f in net.features], aggregator=agg)
    torch.cuda.synchronize()
    fg = None
    jb = list(autojac._J_CACHE.values())[-1]
    print("   J row norms", jb.J.norm(dim=1).cpu().numpy(), "finite", bool(torch.isfinite(jb.J).all()))
    bad = [n for n, p in net.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print("DEFER", on, {k: float(v.detach()) for k, v in ld.items()})
    print("   feature cotangents", fg)
    print("   hook", hook)
    print("   non-finite grads:", len(bad), bad[:4])
    del net, out, ld, comp
