"""Debug aid: every torch.empty* float32 buffer is filled with a NaN whose payload names the allocation site; a NaN in a result
then tells which buffer was read before it was written."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

SITES = {}
_orig = {n: getattr(torch, n) for n in ("empty", "empty_like", "empty_strided")}


def _site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "mo-vae_amd" in fr.filename or "movae_amd" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return "?"


def _wrap(name):
    def f(*a, **k):
        t = _orig[name](*a, **k)
        if t.dtype == torch.float32 and t.is_cuda and t.numel() > 0:
            sid = SITES.setdefault(_site(), len(SITES) + 1)
            base = t if name != "empty_strided" else t
            try:
                flat = torch.as_strided(t, (t.numel(),), (1,)) if not t.is_contiguous() else t.view(-1)
                flat.view(torch.int32).fill_(0x7FC00000 | sid)
            except Exception as e:  # noqa: BLE001
                print("poison failed", name, e)
        return t
    return f


for n in _orig:
    setattr(torch, n, _wrap(n))

import movae_amd  # noqa: E402
from movae_amd import _lib as L, aggregation, autojac, ops  # noqa: E402
from movae_amd.models import get_network  # noqa: E402
from conftest import cfg_from_meta  # noqa: E402
from test_hip_models import Args, _full_case  # noqa: E402

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
    if "latent_dim" in c:
        net.eps_override = torch.randn(batch, c["latent_dim"], generator=torch.Generator().manual_seed(5)).to(dev)
    out = net(x)
    ld = net.loss_function(x, args=out)
    comp = [v for k, v in ld.items() if k != "total_loss"]
    autojac.mtl_backward(losses=comp, features=[out[f] for f in net.features], aggregator=aggregation.UPGrad())
    torch.cuda.synchronize()
    print("DEFER", on, {k: float(v.detach()) for k, v in ld.items()})
    inv = {v: k for k, v in SITES.items()}
    for n, p in net.named_parameters():
        if p.grad is None:
            continue
        g = torch.as_strided(p.grad, (p.grad.numel(),), (1,)) if not p.grad.is_contiguous() else p.grad.reshape(-1)
        bad = ~torch.isfinite(g)
        if bad.any():
            pay = (g[bad].view(torch.int32) & 0x3FFFFF).unique().cpu().tolist()
            print("   NaN in", n, int(bad.sum()), "of", g.numel(), "payload sites", [inv.get(q, q) for q in pay[:6]])
    del net, out, ld, comp
