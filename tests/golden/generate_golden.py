#!/usr/bin/env python3
"""Golden-vector generator (runs ONLY in the build container, never on the GPU box).

Imports the reference's own in-tree arithmetic from /root/reference (read-only):
  * utils/objectives.py                       (recon losses, KL)
  * models/{vae,vq_vae,vq_vae2,betatc_vae}.py  (forward + loss_function)
  * utils/torchmoo/{mgda,aligned_mtl}.py       (Frank-Wolfe / eigh-balance weightings)
and writes small .npz fixtures (inputs + expected outputs only; no reference source text)
next to this script.  The third-party pieces the reference relies on but does not ship
(torchsummary, torchjd base classes) are replaced by in-memory placeholder modules that
contain no arithmetic of their own beyond `G = J @ J.T` / `w @ J` / mean weights, so every
number stored here was produced by the reference's code or by plain torch.autograd on the
reference's modules.

Usage:  python tests/golden/generate_golden.py            (tiny fixtures, seconds)
        python tests/golden/generate_golden.py --full     (adds C1..C5 full-size checksums)
"""
import argparse
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------------------
# In-memory placeholders for the absent third-party modules
# --------------------------------------------------------------------------------------
def _install_placeholders():
    ts = types.ModuleType("torchsummary")
    ts.summary = lambda *a, **k: None
    sys.modules["torchsummary"] = ts

    class Weighting(nn.Module):
        def __class_getitem__(cls, item):
            return cls

    class MeanWeighting(Weighting):
        def forward(self, m):
            k = m.shape[0]
            return torch.full((k,), 1.0 / k, dtype=m.dtype, device=m.device)

    class GramianWeightedAggregator(nn.Module):
        def __init__(self, weighting):
            super().__init__()
            self.gramian_weighting = weighting

        def forward(self, J):
            return self.gramian_weighting(J @ J.T) @ J

    def _absent(*a, **k):
        raise RuntimeError("torchjd arithmetic is not available in this container")

    names = {
        "torchjd": {},
        "torchjd.aggregation": {"UPGrad": _absent},
        "torchjd.aggregation._aggregator_bases": {"GramianWeightedAggregator": GramianWeightedAggregator},
        "torchjd.aggregation._weighting_bases": {"PSDMatrix": torch.Tensor, "Weighting": Weighting},
        "torchjd.aggregation._mean": {"MeanWeighting": MeanWeighting},
        "torchjd.aggregation._utils": {},
        "torchjd.aggregation._utils.pref_vector": {
            "pref_vector_to_str_suffix": lambda p: "",
            "pref_vector_to_weighting": lambda p, default: default,
        },
        "torchjd.aggregation._utils.dual_cone": {"project_weights": _absent},
        "torchjd.aggregation._utils.non_differentiable": {"raise_non_differentiable_error": _absent},
    }
    for name, attrs in names.items():
        m = types.ModuleType(name)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m


def _np(t):
    return t.detach().cpu().numpy().copy()


class _Capture:
    """Records every torch.randn_like draw so the fixture can hold the reparameterisation eps."""

    def __init__(self):
        self.draws = []
        self._orig = torch.randn_like

    def __enter__(self):
        def wrapped(t, *a, **k):
            e = self._orig(t, *a, **k)
            self.draws.append(e.clone())
            return e

        torch.randn_like = wrapped
        return self

    def __exit__(self, *exc):
        torch.randn_like = self._orig


class _Replay:
    def __init__(self, draws):
        self.draws = list(draws)
        self._orig = torch.randn_like

    def __enter__(self):
        it = iter(self.draws)
        torch.randn_like = lambda t, *a, **k: next(it).clone()
        return self

    def __exit__(self, *exc):
        torch.randn_like = self._orig


class _Args:
    def __init__(self, **kw):
        self.__dict__.update(kw)


# --------------------------------------------------------------------------------------
def objectives_fixture():
    from utils import objectives as O

    g = torch.Generator().manual_seed(1234)
    out = {}
    x = torch.rand(3, 3, 8, 8, generator=g)
    r_tanh = torch.tanh(torch.randn(3, 3, 8, 8, generator=g))
    r_sig = torch.sigmoid(torch.randn(3, 3, 8, 8, generator=g) * 3)
    # force a few saturated probabilities so the BCE log clamp (>= -100) is exercised
    r_sig.view(-1)[0] = 0.0
    r_sig.view(-1)[1] = 1.0
    x.view(-1)[0] = 1.0
    x.view(-1)[1] = 0.0
    out["x"], out["r_tanh"], out["r_sig"] = _np(x), _np(r_tanh), _np(r_sig)
    for name, fn, r in [
        ("mse", O.mse_per_pixel_mean, r_tanh),
        ("l1", O.laplacian_per_pixel_mean, r_tanh),
        ("smooth_l1", O.smooth_l1_per_pixel_mean, r_tanh * 3),
        ("bce", O.bce_per_pixel_mean, r_sig),
    ]:
        rr = r.clone().requires_grad_(True)
        v = fn(x, rr)
        (gr,) = torch.autograd.grad(v, rr)
        out[f"{name}.value"], out[f"{name}.grad"] = _np(v), _np(gr)
        if name == "smooth_l1":
            out["r_sl1"] = _np(r)
    mu = torch.randn(5, 7, generator=g).requires_grad_(True)
    lv = (torch.randn(5, 7, generator=g) * 0.7).requires_grad_(True)
    v = O.kl_divergence(mu, lv)
    gmu, glv = torch.autograd.grad(v, [mu, lv])
    out.update({"kl.mu": _np(mu), "kl.log_var": _np(lv), "kl.value": _np(v), "kl.gmu": _np(gmu), "kl.glv": _np(glv)})
    # activation defaults chosen by get_recon_obj_and_activation
    acts = {}
    for obj in ["mse", "bce", "l1", "smooth_l1"]:
        for act in [None, "tanh", "sigmoid", "none"]:
            _, a = O.get_recon_obj_and_activation(obj, recons_activation=act)
            acts[f"{obj}|{act}"] = a
    out["activation_table"] = np.array([f"{k}={v}" for k, v in acts.items()])
    np.savez_compressed(os.path.join(HERE, "objectives.npz"), **out)
    print("objectives.npz", len(out))


def weightings_fixture():
    from utils.torchmoo.mgda import MGDA, MGDAWeighting
    from utils.torchmoo.aligned_mtl import AlignedMTLWeighting

    out = {}
    g = torch.Generator().manual_seed(77)
    cases = {
        "kat": torch.tensor([[-4.0, 1.0, 1.0], [6.0, 1.0, 1.0]]),
        "k2": torch.randn(2, 50, generator=g),
        "k3": torch.randn(3, 40, generator=g) * torch.tensor([[1.0], [10.0], [0.1]]),
        "k3_zero_row": torch.cat([torch.randn(2, 30, generator=g), torch.zeros(1, 30)]),
        "k4": torch.randn(4, 64, generator=g),
        "k4_conflict": torch.randn(4, 16, generator=g) - 0.8 * torch.randn(1, 16, generator=g),
        "k5_rankdef": (torch.randn(5, 2, generator=g) @ torch.randn(2, 33, generator=g)),
        "k2_parallel": torch.stack([torch.arange(1.0, 9.0), -2 * torch.arange(1.0, 9.0)]),
    }
    for name, J in cases.items():
        K = J.shape[0]
        G = J @ J.T
        losses = torch.rand(K, generator=g) + 0.1
        if name == "kat":
            losses = torch.tensor([0.5, 2.0])
        out[f"{name}.J"], out[f"{name}.G"], out[f"{name}.losses"] = _np(J), _np(G), _np(losses)
        for nt in ["none", "l2", "loss", "loss+"]:
            w = MGDAWeighting(norm_type=nt)
            w.set_losses(losses)
            alpha = w(G)
            out[f"{name}.mgda.{nt}"] = _np(alpha)
            out[f"{name}.mgda.{nt}.iters"] = np.array(w.convergence_count)
        ws = MGDAWeighting(norm_type="none", stable=True)
        out[f"{name}.mgda.stable"] = _np(ws(G))
        for sm in ["min", "median", "rmse"]:
            out[f"{name}.amtl.{sm}"] = _np(AlignedMTLWeighting(None, scale_mode=sm)(G))
    # the aggregator-level docstring KATs (utils/torchmoo/mgda.py:54-86)
    J = cases["kat"]
    for nt in ["none", "l2", "loss", "loss+"]:
        A = MGDA(norm_type=nt)
        A.set_losses(torch.tensor([0.5, 2.0]))
        out[f"kat.mgda_agg.{nt}"] = _np(A(J))
    np.savez_compressed(os.path.join(HERE, "weightings.npz"), **out)
    print("weightings.npz", len(out))


def agg_variants_fixture():
    """In-tree pieces of NUPGrad / PNUPGrad / COMFORT: the two Gramian normalisations, the
    regulariser and COMFORT's beta schedule.  Their QP (torchjd `project_weights`) and torchjd's UPGrad are
    third-party and absent, so the aggregators as a whole stay "parity unpinned" like UPGrad."""
    from utils.torchmoo import nupgrad as NU
    from utils.torchmoo import pnupgrad as PN
    from utils.torchmoo.comfort import beta_schedule

    fx = np.load(os.path.join(HERE, "weightings.npz"))
    out = {}
    names = sorted({k.split(".")[0] for k in fx.files if k.endswith(".G")})
    out["cases"] = np.array(names)
    for name in names:
        G = torch.from_numpy(fx[f"{name}.G"])
        for eps in (1e-4, 1e-2):
            out[f"{name}.min_l2.{eps}"] = _np(NU.normalize_by_min_l2_norm(G, eps))
            out[f"{name}.min_l2_p.{eps}"] = _np(PN.normalize_by_min_l2_norm(G, eps))
            out[f"{name}.cosine.{eps}"] = _np(PN.normalize(G, eps))
            out[f"{name}.reg.{eps}"] = _np(NU.regularize(NU.normalize_by_min_l2_norm(G, eps), eps))
    zero = torch.zeros(3, 3)
    out["zero.min_l2"] = _np(NU.normalize_by_min_l2_norm(zero, 1e-4))
    grid = []
    for total in (1, 2, 10, 50):
        for epoch in (1, 2, 5, 10, 50, 60):
            for (k, a, l, u) in ((1.0, 1.0, 0.01, 1.0), (0.0, 1.0, 0.1, 0.9), (5.0, 2.0, 0.0, 1.0), (-1.0, 0.5, 0.2, 0.7)):
                grid.append((epoch, total, k, a, l, u, beta_schedule(epoch, total, k=k, a=a, l=l, u=u)))
    out["beta_schedule"] = np.array(grid, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "agg_variants.npz"), **out)
    print("agg_variants.npz", len(out))


def _model_fixture(tag, arch, seed, B, input_size, args_kw, objective="mse"):
    from models import get_network
    from models.betatc_vae import BetaTCVAE

    out = {}
    args = _Args(arch=arch, batch_size=B, dataset_size=1000, recons_objective=objective,
                 recons_activation=None, loss_weights=None, **args_kw)
    torch.manual_seed(seed)
    np.random.seed(seed)
    BetaTCVAE.num_iter = 0
    net = get_network(input_size, num_channels=3, args=args, device=torch.device("cpu"))
    net.train()
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    for k, v in sd0.items():
        out[f"sd0.{k}"] = _np(v)
    gen = torch.Generator().manual_seed(seed + 1)
    x = torch.rand(B, 3, input_size, input_size, generator=gen)
    out["x"] = _np(x)
    out["meta"] = np.array([f"arch={arch}", f"seed={seed}", f"B={B}", f"input_size={input_size}",
                            f"objective={objective}", "dataset_size=1000"]
                           + [f"{k}={v}" for k, v in args_kw.items()])
    out["lambda_weights"] = np.array([f"{k}={v!r}" for k, v in net.lambda_weights.items()])
    out["features"] = np.array(list(net.features))
    out["objectives"] = np.array(list(net.objectives.keys()))

    # ---- forward + losses -----------------------------------------------------------
    with _Capture() as cap:
        outputs = net(x)
    for i, e in enumerate(cap.draws):
        out[f"eps.{i}"] = _np(e)
    loss_dict = net.loss_function(x, args=outputs)
    for k, v in outputs.items():
        if isinstance(v, torch.Tensor):
            out[f"out.{k}"] = _np(v)
        elif isinstance(v, float):
            out[f"out.{k}"] = np.array(v)
    for k, v in loss_dict.items():
        out[f"loss.{k}"] = _np(v)
    names = [n for n, _ in net.named_parameters()]
    params = [p for _, p in net.named_parameters()]
    # de-duplicate aliases (VQVAE2 registers vq_top/vq_bottom twice): named_parameters already does.
    comp = [(k, v) for k, v in loss_dict.items() if k != "total_loss"]
    # ---- per-loss gradients by plain autograd (total derivatives) ----------------------
    for i, (k, v) in enumerate(comp):
        gs = torch.autograd.grad(v, params, retain_graph=True, allow_unused=True)
        for n, p, gq in zip(names, params, gs):
            out[f"gloss.{i}.{n}"] = _np(gq if gq is not None else torch.zeros_like(p))
        feats = [outputs[f] for f in net.features]
        fg = torch.autograd.grad(v, feats, retain_graph=True, allow_unused=True)
        for f, ft, gq in zip(net.features, feats, fg):
            out[f"gfeat.{i}.{f}"] = _np(gq if gq is not None else torch.zeros_like(ft))
    # ---- sum path: total_loss.backward() + one Adam step --------------------------------
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    opt.zero_grad()
    loss_dict["total_loss"].backward()
    for n, p in zip(names, params):
        out[f"gsum.{n}"] = _np(p.grad if p.grad is not None else torch.zeros_like(p))
    opt.step()
    for k, v in net.state_dict().items():
        out[f"sd1.{k}"] = _np(v)
    # ---- second step (same batch, same eps) pins BN running stats / num_iter behaviour ----
    with _Replay(cap.draws):
        outputs2 = net(x)
    ld2 = net.loss_function(x, args=outputs2)
    for k, v in ld2.items():
        out[f"loss2.{k}"] = _np(v)
    # ---- eval-mode forward on the updated model (BN uses running stats) ------------------
    net.eval()
    with torch.no_grad(), _Replay(cap.draws):
        oe = net(x)
    out["eval.recons"] = _np(oe["recons"])
    np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **out)
    size = os.path.getsize(os.path.join(HERE, f"{tag}.npz"))
    print(f"{tag}.npz keys={len(out)} bytes={size}")


def model_fixtures():
    _model_fixture("vae_tiny", "vae", 42, 4, 16, dict(latent_dim=8, hidden_dims=[8, 16]))
    _model_fixture("vae_tiny_bce", "vae", 123123, 6, 32, dict(latent_dim=6, hidden_dims=[4, 8, 16]), objective="bce")
    _model_fixture("vae_1x1", "vae", 7, 5, 8, dict(latent_dim=4, hidden_dims=[4, 8, 8]))
    _model_fixture("vq_vae_tiny", "vq_vae", 42, 3, 16,
                   dict(embedding_dim=8, num_embeddings=16, hidden_dims=[8, 16], num_residual_layers=2))
    _model_fixture("vq_vae2_tiny", "vq_vae2", 12341234, 2, 32,
                   dict(embedding_dim=8, num_embeddings=16, hidden_dims=[16, 32], num_residual_layers=2))
    _model_fixture("betatc_vae_tiny", "betatc_vae", 42, 5, 16,
                   dict(latent_dim=6, hidden_dims=[8, 16], anneal_steps=200), objective="mse")
    gg_vae_fixture()


def gg_vae_fixture():
    """the gradient-guided VAE (models/gg_vae.py), edge matching version 1."""
    _model_fixture("gg_vae_tiny", "gg_vae", 31, 4, 16, dict(latent_dim=8, hidden_dims=[8, 16]))
    _model_fixture("gg_vq_vae_tiny", "gg_vq_vae", 57, 3, 16,
                   dict(embedding_dim=8, num_embeddings=16, hidden_dims=[8, 16], num_residual_layers=2))
    _model_fixture("gg_vq_vae2_tiny", "gg_vq_vae2", 77, 2, 32,
                   dict(embedding_dim=8, num_embeddings=16, hidden_dims=[16, 32], num_residual_layers=2))
    # two versioned archs end to end (the objective order / K = 5 plumbing); every variant's arithmetic is in edge_variants.npz
    _model_fixture("gg_vq_vae_v4_tiny", "gg_vq_vae_v4", 91, 3, 16,
                   dict(embedding_dim=8, num_embeddings=16, hidden_dims=[8, 16], num_residual_layers=2))
    _model_fixture("gg_vae_v5_tiny", "gg_vae_v5", 93, 4, 16, dict(latent_dim=8, hidden_dims=[8, 16]))
    edge_variants_fixture()


def edge_variants_fixture():
    """Every edge-matching variant of models/gg_vae.py and models/gg_vq_vae.py on one random (inputs, recons) pair: the loss and
    its gradient w.r.t. recons, computed by the reference's own methods."""
    from models.gg_vae import GGVAE
    from models.gg_vq_vae import GGVQVAE

    torch.manual_seed(5)
    vq = GGVQVAE(in_channels=3, embedding_dim=4, num_embeddings=8, hidden_dims=[4, 8], num_residual_layers=1, input_size=16,
                 version="v2")
    va = GGVAE(latent_dim=4, input_size=16, in_channels=3, hidden_dims=[4, 8])
    gen = torch.Generator().manual_seed(6)
    x = torch.rand(3, 3, 12, 10, generator=gen)
    out = {"x": _np(x)}
    # "flat" recons has constant patches: exactly-zero Sobel responses (the clamp branches of normalize / cosine_similarity)
    cases = {"rand": torch.rand(3, 3, 12, 10, generator=gen) * 1.4 - 0.2, "far": torch.rand(3, 3, 12, 10, generator=gen) * 6 - 3}
    flat = cases["rand"].clone()
    flat[:, :, 2:8, 3:9] = 0.25
    methods = {"signed_mse": vq.edge_matching_loss_v1, "mag": vq.edge_matching_loss_v2, "maxnorm": vq.edge_matching_loss_v3,
               "angle": vq.edge_matching_loss_v4, "masked": vq.edge_matching_loss_v5, "cosine": vq.edge_matching_loss_v6,
               "gg_vae.mag": va.edge_matching_loss, "gg_vae.maxnorm": va.edge_matching_loss_v2, "gg_vae.angle": va.edge_matching_loss_v3,
               "gg_vae.cosine": va.edge_matching_loss_v5}
    for cname, r in list(cases.items()) + [("flat", flat)]:
        out[f"{cname}.recons"] = _np(r)
        for mname, fn in methods.items():
            if cname == "flat" and mname.split(".")[-1] in ("angle", "cosine"):
                continue  # atan2 / the 1e-12 clamp give NaN / 1e20-scale gradients there: not a numerical fixture
            rr = r.clone().requires_grad_(True)
            loss = fn(x, rr)
            (g,) = torch.autograd.grad(loss, rr)
            out[f"{cname}.{mname}.loss"] = _np(loss)
            out[f"{cname}.{mname}.grad"] = _np(g)
    np.savez_compressed(os.path.join(HERE, "edge_variants.npz"), **out)
    print("edge_variants.npz", len(out))


# --------------------------------------------------------------------------------------
def pixelcnn_fixture():
    """models/pixelcnn_prior.py (imports torch only): PixelCNN and HierarchicalPixelCNN at tiny sizes -- init state_dict (incl. the
    mask buffers), logits, cross-entropy (main.py:1003-1006 / loss_function), every parameter's gradient, and the parameters
    after one step of the prior loop main.py:995-1011 (clip_grad_norm_ 1.0, Adam lr 3e-4) followed by a second forward (which
    re-applies the in-place weight mask)."""
    import torch.nn.functional as F
    from models.pixelcnn_prior import HierarchicalPixelCNN, PixelCNN

    out = {}
    K, D, hid, L, seed = 16, 8, 16, 2, 7
    g = torch.Generator().manual_seed(70)
    z = torch.randint(0, K, (3, 8, 8), generator=g)
    z_top = torch.randint(0, K, (2, 4, 4), generator=g)
    z_bot = torch.randint(0, K, (2, 8, 8), generator=g)
    out["meta"] = np.array([f"num_embeddings={K}", f"embedding_dim={D}", f"hidden_channels={hid}", f"num_layers={L}", f"seed={seed}",
                            "lr=0.0003"])
    out["z"], out["z_top"], out["z_bottom"] = _np(z), _np(z_top), _np(z_bot)
    for tag, hier in (("flat", False), ("hier", True)):
        torch.manual_seed(seed)
        net = (HierarchicalPixelCNN(K, D, hid, L) if hier else PixelCNN(K, D, hid, L)).train()
        for k, v in net.state_dict().items():
            out[f"{tag}.sd0.{k}"] = _np(v)
        opt = torch.optim.Adam(net.parameters(), lr=3e-4, weight_decay=0.0)
        opt.zero_grad()
        if hier:
            o = net(z_top, z_bot)
            ld = net.loss_function(z_top, z_bot)
            out[f"{tag}.logits_top"], out[f"{tag}.logits_bottom"] = _np(o["logits_top"]), _np(o["logits_bottom"])
        else:
            logits = net(z)
            ld = {"total_loss": F.cross_entropy(logits.permute(0, 2, 3, 1).reshape(-1, K), z.reshape(-1))}
            out[f"{tag}.logits"] = _np(logits)
        for k, v in ld.items():
            out[f"{tag}.loss.{k}"] = _np(v)
        ld["total_loss"].backward()
        for n, p in net.named_parameters():
            out[f"{tag}.g.{n}"] = _np(p.grad if p.grad is not None else torch.zeros_like(p))
        out[f"{tag}.gnorm"] = _np(torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0))
        opt.step()
        if hier:
            ld2 = net.loss_function(z_top, z_bot)
        else:
            ld2 = {"total_loss": F.cross_entropy(net(z).permute(0, 2, 3, 1).reshape(-1, K), z.reshape(-1))}
        for k, v in ld2.items():
            out[f"{tag}.loss2.{k}"] = _np(v)
        for k, v in net.state_dict().items():
            out[f"{tag}.sd1.{k}"] = _np(v)
        print("pixelcnn", tag, {k: float(v) for k, v in ld.items()}, {k: float(v) for k, v in ld2.items()})
    np.savez_compressed(os.path.join(HERE, "pixelcnn_tiny.npz"), **out)


def _checksum(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.norm().item()])


def full_fixture():
    """Step-0 scalars + per-parameter (sum, L2) gradient checksums at BASELINE.json's shapes.

    Inputs and parameters are regenerated from the seed by the tests (torch's CPU generator is
    platform independent), so only scalars are stored.
    """
    from models import get_network
    from models.betatc_vae import BetaTCVAE

    cfgs = {
        "C1": ("vae", 42, 128, 32, 50000, dict(latent_dim=128, hidden_dims=[32, 64, 128, 256, 512])),
        "C2": ("vae", 42, 256, 32, 50000, dict(latent_dim=128, hidden_dims=[32, 64, 128, 256, 512])),
        "C3": ("vq_vae", 42, 16, 64, 162770, dict(embedding_dim=64, num_embeddings=512, hidden_dims=[128, 256], num_residual_layers=2)),
        "C4": ("vq_vae2", 42, 2, 256, 30000, dict(embedding_dim=64, num_embeddings=512, hidden_dims=[128, 256], num_residual_layers=2)),
        "C5": ("betatc_vae", 42, 8, 256, 1281167, dict(latent_dim=128, hidden_dims=[32, 64, 128, 256, 512], anneal_steps=200)),
    }
    out = {}
    for tag, (arch, seed, B, size, dsz, kw) in cfgs.items():
        args = _Args(arch=arch, batch_size=B, dataset_size=dsz, recons_objective="mse",
                     recons_activation=None, loss_weights=None, **kw)
        torch.manual_seed(seed)
        BetaTCVAE.num_iter = 0
        net = get_network(size, num_channels=3, args=args, device=torch.device("cpu"))
        net.train()
        gen = torch.Generator().manual_seed(seed + 1)
        x = torch.rand(B, 3, size, size, generator=gen)
        egen = torch.Generator().manual_seed(seed + 2)
        orig = torch.randn_like
        torch.randn_like = lambda t, *a, **k: torch.randn(t.shape, generator=egen)
        try:
            o = net(x)
        finally:
            torch.randn_like = orig
        ld = net.loss_function(x, args=o)
        ld["total_loss"].backward()
        out[f"{tag}.meta"] = np.array([f"arch={arch}", f"seed={seed}", f"B={B}", f"input_size={size}", f"dataset_size={dsz}"]
                                      + [f"{k}={v}" for k, v in kw.items()])
        for k, v in ld.items():
            out[f"{tag}.loss.{k}"] = _np(v)
        out[f"{tag}.recons"] = _checksum(o["recons"])
        for n, p in net.named_parameters():
            out[f"{tag}.p.{n}"] = _checksum(p)
            out[f"{tag}.g.{n}"] = _checksum(p.grad if p.grad is not None else torch.zeros_like(p))
        for k in o:
            if k.startswith("encoding_inds") and o[k] is not None:
                out[f"{tag}.hist.{k}"] = np.bincount(_np(o[k]).reshape(-1), minlength=kw["num_embeddings"])
        print(tag, {k: float(v) for k, v in ld.items()})
    np.savez_compressed(os.path.join(HERE, "full_configs.npz"), **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--only-agg-variants", action="store_true", help="regenerate agg_variants.npz only")
    ap.add_argument("--only-gg-vae", action="store_true", help="regenerate gg_vae_tiny.npz only")
    ap.add_argument("--only-pixelcnn", action="store_true", help="regenerate pixelcnn_tiny.npz only")
    a = ap.parse_args()
    _install_placeholders()
    sys.path.insert(0, REF)
    torch.set_num_threads(8)
    if a.only_agg_variants:
        agg_variants_fixture()
        sys.exit(0)
    if a.only_gg_vae:
        gg_vae_fixture()
        sys.exit(0)
    if a.only_pixelcnn:
        pixelcnn_fixture()
        sys.exit(0)
    objectives_fixture()
    weightings_fixture()
    agg_variants_fixture()
    model_fixtures()
    pixelcnn_fixture()
    if a.full:
        full_fixture()
