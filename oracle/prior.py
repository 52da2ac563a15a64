"""Oracle: PixelCNN prior over VQ code grids on CPU (TEST INFRASTRUCTURE, see oracle/__init__.py).

Plain PyTorch restatement of the reference's models/pixelcnn_prior.py -- MaskedConv2d :25-54, GatedResBlock :57-92,
PixelCNN :262-321, HierarchicalPixelCNN :352-399 -- and of one step of its training loop main.py:995-1011 (cross-entropy,
clip_grad_norm_ 1.0, Adam).  Pinned by tests/golden/pixelcnn_tiny.npz, which tests/golden/generate_golden.py produced by
importing the reference module itself.  Functional: parameters live in an ordered dict keyed like the reference's state_dict."""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


def _mask(kind, w):
    """pixelcnn_prior.py:40-50."""
    m = torch.zeros_like(w)
    kh, kw = w.shape[2], w.shape[3]
    m[:, :, : kh // 2, :] = 1.0
    m[:, :, kh // 2, : kw // 2] = 1.0
    if kind == "B":
        m[:, :, kh // 2, kw // 2] = 1.0
    return m


def _conv_init(sd, name, cin, cout, k, mask=None):
    c = nn.Conv2d(cin, cout, k)
    sd[name + ".weight"], sd[name + ".bias"] = c.weight.detach().clone(), c.bias.detach().clone()
    if mask is not None:
        sd[name + ".mask"] = _mask(mask, c.weight.detach())


def init_pixelcnn(sd, prefix, K, D, hidden, layers, kernel=7, cond=0):
    """Creation order of PixelCNN.__init__ (pixelcnn_prior.py:283-305): the RNG draws replay the reference's."""
    sd[prefix + "embedding.weight"] = nn.Embedding(K, D).weight.detach().clone()
    _conv_init(sd, prefix + "conv_in", D + cond, hidden, kernel, "A")
    for i in range(layers):
        b = f"{prefix}res_blocks.{i}."
        _conv_init(sd, b + "conv1", hidden, hidden // 2, 1)
        _conv_init(sd, b + "conv2", hidden // 2, hidden // 2, 3, "B")
        _conv_init(sd, b + "conv_gate", hidden // 2, hidden, 1)
        _conv_init(sd, b + "conv_feature", hidden // 2, hidden, 1)
    _conv_init(sd, prefix + "conv_out.1", hidden, hidden, 1)
    _conv_init(sd, prefix + "conv_out.3", hidden, K, 1)


def init_state(cfg, seed):
    torch.manual_seed(seed)
    sd = OrderedDict()
    K, D, hid, L = cfg["num_embeddings"], cfg["embedding_dim"], cfg["hidden_channels"], cfg["num_layers"]
    if cfg.get("hierarchical"):
        init_pixelcnn(sd, "prior_top.", K, D, hid, L)
        sd["embedding_top.weight"] = nn.Embedding(K, D).weight.detach().clone()
        up = nn.ConvTranspose2d(D, D, kernel_size=4, stride=2, padding=1)
        sd["upsample_top.weight"], sd["upsample_top.bias"] = up.weight.detach().clone(), up.bias.detach().clone()
        init_pixelcnn(sd, "prior_bottom.", K, D, hid, L, cond=D)
    else:
        init_pixelcnn(sd, "", K, D, hid, L)
    return sd


def parameter_names(sd):
    return [k for k in sd if not k.endswith(".mask")]


def _masked_conv(sd, name, x, pad):
    w = sd[name + ".weight"]
    with torch.no_grad():
        w.mul_(sd[name + ".mask"])  # pixelcnn_prior.py:52: in place, outside the tape
    return F.conv2d(x, w, sd[name + ".bias"], padding=pad)


def pixelcnn_logits(sd, prefix, z, layers, cond=None):
    """pixelcnn_prior.py:307-321."""
    x = F.embedding(z, sd[prefix + "embedding.weight"]).permute(0, 3, 1, 2).contiguous()
    if cond is not None:
        x = torch.cat([x, cond], dim=1)
    k = sd[prefix + "conv_in.weight"].shape[-1]
    x = _masked_conv(sd, prefix + "conv_in", x, k // 2)
    for i in range(layers):
        b = f"{prefix}res_blocks.{i}."
        out = F.relu(F.conv2d(x, sd[b + "conv1.weight"], sd[b + "conv1.bias"]))
        out = F.relu(_masked_conv(sd, b + "conv2", out, 1))
        gate = torch.sigmoid(F.conv2d(out, sd[b + "conv_gate.weight"], sd[b + "conv_gate.bias"]))
        feat = torch.tanh(F.conv2d(out, sd[b + "conv_feature.weight"], sd[b + "conv_feature.bias"]))
        x = x + gate * feat
    x = F.relu(F.conv2d(F.relu(x), sd[prefix + "conv_out.1.weight"], sd[prefix + "conv_out.1.bias"]))
    return F.conv2d(x, sd[prefix + "conv_out.3.weight"], sd[prefix + "conv_out.3.bias"])


def xent(logits, z):
    """main.py:1003-1006."""
    return F.cross_entropy(logits.permute(0, 2, 3, 1).reshape(-1, logits.shape[1]), z.reshape(-1))


def losses(sd, cfg, z_top, z_bottom=None):
    L = cfg["num_layers"]
    if not cfg.get("hierarchical"):
        logits = pixelcnn_logits(sd, "", z_top, L)
        return {"total_loss": xent(logits, z_top)}, {"logits": logits}
    lt = pixelcnn_logits(sd, "prior_top.", z_top, L)
    emb = F.embedding(z_top, sd["embedding_top.weight"]).permute(0, 3, 1, 2).contiguous()
    up = F.conv_transpose2d(emb, sd["upsample_top.weight"], sd["upsample_top.bias"], stride=2, padding=1)
    lb = pixelcnn_logits(sd, "prior_bottom.", z_bottom, L, cond=up)
    a, b = xent(lt, z_top), xent(lb, z_bottom)
    return {"loss_top": a, "loss_bottom": b, "total_loss": a + b}, {"logits_top": lt, "logits_bottom": lb}


class PriorTrainer:
    """One optimisation step of main.py:995-1011: zero_grad, loss, backward, clip_grad_norm_(1.0), Adam(lr) step."""

    def __init__(self, cfg, seed, lr=3e-4):
        self.cfg = dict(cfg)
        self.sd = init_state(cfg, seed)
        self.params = OrderedDict((n, self.sd[n].requires_grad_(True)) for n in parameter_names(self.sd))
        self.opt = torch.optim.Adam(list(self.params.values()), lr=lr, weight_decay=0.0)

    def grads(self, z_top, z_bottom=None):
        ld, out = losses(self.sd, self.cfg, z_top, z_bottom)
        gs = torch.autograd.grad(ld["total_loss"], list(self.params.values()), allow_unused=True)
        return ld, out, OrderedDict((n, torch.zeros_like(p) if g is None else g) for (n, p), g in zip(self.params.items(), gs))

    def step(self, z_top, z_bottom=None):
        ld, out, g = self.grads(z_top, z_bottom)
        for n, p in self.params.items():
            p.grad = g[n].detach().clone()
        torch.nn.utils.clip_grad_norm_(list(self.params.values()), 1.0)
        self.opt.step()
        return {k: float(v.detach()) for k, v in ld.items()}
