"""Oracle: the four hot-path networks as pure functions of a parameter dict (TEST
INFRASTRUCTURE, see oracle/__init__.py).

Each network is described by (a) ``init_<arch>`` -- replays the reference constructor's layer
*creation* order on torch's CPU generator so the same seed yields the same tensors, then returns
them in the reference's ``state_dict`` order and under its key names -- and (b)
``forward_<arch>`` / ``losses_<arch>`` -- the forward pass and loss dictionary written with
torch.nn.functional primitives on that dict.  NCHW fp32 throughout, like the reference.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import objectives as O

LRELU = 0.01  # nn.LeakyReLU() default slope, models/vae.py:125


# ---------------------------------------------------------------------------------------
# parameter drawing (torch.nn layers are used only for their reset_parameters RNG sequence)
# ---------------------------------------------------------------------------------------
class _Draw:
    def __init__(self):
        self.sd = OrderedDict()

    def conv(self, key, cin, cout, k, bias=True):
        m = nn.Conv2d(cin, cout, k, bias=bias)
        self.sd[key + ".weight"] = m.weight.detach().clone()
        if bias:
            self.sd[key + ".bias"] = m.bias.detach().clone()

    def convT(self, key, cin, cout, k):
        m = nn.ConvTranspose2d(cin, cout, k)
        self.sd[key + ".weight"] = m.weight.detach().clone()
        self.sd[key + ".bias"] = m.bias.detach().clone()

    def linear(self, key, fin, fout):
        m = nn.Linear(fin, fout)
        self.sd[key + ".weight"] = m.weight.detach().clone()
        self.sd[key + ".bias"] = m.bias.detach().clone()

    def bn(self, key, c):
        self.sd[key + ".weight"] = torch.ones(c)
        self.sd[key + ".bias"] = torch.zeros(c)
        self.sd[key + ".running_mean"] = torch.zeros(c)
        self.sd[key + ".running_var"] = torch.ones(c)
        self.sd[key + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    def codebook(self, key, K, D):
        # models/vq_vae.py:24-25 -- nn.Embedding draws normal_ first, then is overwritten uniformly
        m = nn.Embedding(K, D)
        m.weight.data.uniform_(-1.0 / K, 1.0 / K)
        self.sd[key + ".embedding.weight"] = m.weight.detach().clone()


BUFFER_SUFFIXES = (".running_mean", ".running_var", ".num_batches_tracked")


def is_buffer(key):
    return key.endswith(BUFFER_SUFFIXES) or key in ("sobel_x", "sobel_y")


def parameter_names(sd, arch=None):
    """Names in ``named_parameters()`` order (aliases de-duplicated like nn.Module does)."""
    out = []
    for k in sd:
        if is_buffer(k) or k.startswith(("vq_top.", "vq_bottom.")):
            continue
        out.append(k)
    return out


def batch_norm(sd, key, h, train, eps=1e-5, momentum=0.1):
    """nn.BatchNorm2d semantics (models/vae.py:123): batch statistics in training mode, one
    running-stat update per call with the unbiased variance."""
    g, b = sd[key + ".weight"], sd[key + ".bias"]
    if train:
        n = h.numel() // h.shape[1]
        mean = h.mean(dim=(0, 2, 3))
        var = ((h - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
        with torch.no_grad():
            sd[key + ".running_mean"].mul_(1 - momentum).add_(momentum * mean.detach())
            sd[key + ".running_var"].mul_(1 - momentum).add_(momentum * var.detach() * (n / max(n - 1, 1)))
            sd[key + ".num_batches_tracked"].add_(1)
    else:
        mean, var = sd[key + ".running_mean"], sd[key + ".running_var"]
    xhat = (h - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + eps)
    return xhat * g[None, :, None, None] + b[None, :, None, None]


# ---------------------------------------------------------------------------------------
# VAE  (models/vae.py)
# ---------------------------------------------------------------------------------------
def init_vae(cfg):
    """Creation order models/vae.py:117-173; registration order puts final_layer before decoder."""
    hd, C, L = list(cfg["hidden_dims"]), cfg.get("in_channels", 3), cfg["latent_dim"]
    sp = cfg["input_size"] // 2 ** len(hd)
    d = _Draw()
    cin = C
    for i, h in enumerate(hd):
        d.conv(f"encoder.{i}.0", cin, h, 3)
        d.bn(f"encoder.{i}.1", h)
        cin = h
    feat = hd[-1] * sp * sp
    d.linear("mu", feat, L)
    d.linear("log_var", feat, L)
    d.linear("decoder_input", L, feat)
    rev = hd[::-1]
    for i in range(len(rev) - 1):
        d.convT(f"decoder.{i + 1}.0", rev[i], rev[i + 1], 3)
        d.bn(f"decoder.{i + 1}.1", rev[i + 1])
    d.convT("final_layer.0", rev[-1], rev[-1], 3)
    d.bn("final_layer.1", rev[-1])
    d.conv("final_layer.3", rev[-1], C, 3)
    sd = OrderedDict()
    for k, v in d.sd.items():
        if not k.startswith("decoder."):
            sd[k] = v
    for k, v in d.sd.items():
        if k.startswith("decoder."):
            sd[k] = v
    return sd


def forward_vae(sd, x, cfg, eps, train=True):
    """models/vae.py:181-206."""
    hd = list(cfg["hidden_dims"])
    sp = cfg["input_size"] // 2 ** len(hd)
    h = x
    for i in range(len(hd)):
        h = F.conv2d(h, sd[f"encoder.{i}.0.weight"], sd[f"encoder.{i}.0.bias"], stride=2, padding=1)
        h = F.leaky_relu(batch_norm(sd, f"encoder.{i}.1", h, train), LRELU)
    h = h.flatten(1)
    mu = F.linear(h, sd["mu.weight"], sd["mu.bias"])
    log_var = F.linear(h, sd["log_var.weight"], sd["log_var.bias"])
    z = mu + eps * torch.exp(0.5 * log_var)  # models/vae.py:187-192
    h = F.linear(z, sd["decoder_input.weight"], sd["decoder_input.bias"]).unflatten(1, (hd[-1], sp, sp))
    for i in range(len(hd) - 1):
        k = f"decoder.{i + 1}"
        h = F.conv_transpose2d(h, sd[k + ".0.weight"], sd[k + ".0.bias"], stride=2, padding=1, output_padding=1)
        h = F.leaky_relu(batch_norm(sd, k + ".1", h, train), LRELU)
    h = F.conv_transpose2d(h, sd["final_layer.0.weight"], sd["final_layer.0.bias"], stride=2, padding=1, output_padding=1)
    h = F.leaky_relu(batch_norm(sd, "final_layer.1", h, train), LRELU)
    h = F.conv2d(h, sd["final_layer.3.weight"], sd["final_layer.3.bias"], padding=1)
    _, act = O.resolve_objective(cfg.get("recons_objective", "mse"), cfg.get("recons_activation"))
    return {"recons": O.ACTIVATIONS[act](h), "mu": mu, "log_var": log_var, "z": z}


def losses_vae(x, out, cfg):
    """models/vae.py:211-228 with the factory's kld weight batch_size/dataset_size
    (models/__init__.py:49-55: the CLI batch size, not the actual batch)."""
    fn, _ = O.resolve_objective(cfg.get("recons_objective", "mse"), cfg.get("recons_activation"))
    lw = cfg["lambda_weights"]
    r = lw["reconstruction_loss"] * fn(x, out["recons"])
    k = lw["kld_loss"] * O.kl_divergence(out["mu"], out["log_var"])
    return OrderedDict(reconstruction_loss=r, kld_loss=k, total_loss=r + k)


# ---------------------------------------------------------------------------------------
# Gradient-guided VAE (models/gg_vae.py) -- the VAE plus two Sobel edge losses
# ---------------------------------------------------------------------------------------
GG_EPS = 1e-8  # models/gg_vae.py:8


def _sobel():
    """models/gg_vae.py:44-53 -- (3,1,3,3) depthwise filters, registered as buffers BEFORE the sub-modules in state_dict order."""
    sx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]])
    sy = torch.tensor([[-1.0, -2.0, -1.0], [0.0, 0.0, 0.0], [1.0, 2.0, 1.0]])
    return sx.expand(3, 1, 3, 3).clone(), sy.expand(3, 1, 3, 3).clone()


def _with_sobel(sd):
    out = OrderedDict()
    out["sobel_x"], out["sobel_y"] = _sobel()
    out.update(sd)
    return out


def init_gg_vae(cfg):
    sd = OrderedDict()
    sd["sobel_x"], sd["sobel_y"] = _sobel()
    sd.update(init_vae(cfg))  # the Sobel tensors draw nothing from the RNG
    return sd


def edge_weighted_pixel_loss(x, recons):
    """models/gg_vae.py:125-137."""
    sx, sy = _sobel()
    gx = F.conv2d(x, sx, padding=1, groups=x.size(1))
    gy = F.conv2d(x, sy, padding=1, groups=x.size(1))
    w = torch.sqrt(gx ** 2 + gy ** 2 + GG_EPS).max(dim=1)[0]
    w = w / (w.max() + GG_EPS)
    return (w.unsqueeze(1) * F.mse_loss(recons, x, reduction="none")).mean()


def edge_matching_loss(x, recons):
    """models/gg_vae.py:139-156 (edge_matching_version 1)."""
    sx, sy = _sobel()
    g = x.size(1)
    gp = torch.sqrt(F.conv2d(recons, sx, padding=1, groups=g) ** 2 + F.conv2d(recons, sy, padding=1, groups=g) ** 2 + GG_EPS)
    gt = torch.sqrt(F.conv2d(x, sx, padding=1, groups=g) ** 2 + F.conv2d(x, sy, padding=1, groups=g) ** 2 + GG_EPS)
    return F.smooth_l1_loss(gp, gt)


def _sobel_pair(x, recons):
    sx, sy = _sobel()
    g = x.size(1)
    return (F.conv2d(recons, sx, padding=1, groups=g), F.conv2d(recons, sy, padding=1, groups=g),
            F.conv2d(x, sx, padding=1, groups=g), F.conv2d(x, sy, padding=1, groups=g))


def edge_matching_variant(x, recons, mode):
    """The edge-matching family, one restatement per variant (the names are include/movae.h's enum movae_edge_match):
    mag         models/gg_vae.py:139-156 == gg_vq_vae.py:184-199 (v2) == gg_vq_vae2.py:118-129
    signed_mse  models/gg_vq_vae.py:172-182 (v1)
    maxnorm     models/gg_vae.py:158-173 (v2) == gg_vq_vae.py:201-216 (v3)
    angle       models/gg_vae.py:176-190 (v3) == gg_vq_vae.py:219-232 (v4)
    masked      models/gg_vq_vae.py:234-247 (v5)
    cosine      models/gg_vae.py:192-208 (v5) == gg_vq_vae.py:249-264 (v6)"""
    if mode == "mag":
        return edge_matching_loss(x, recons)
    rx, ry, tx, ty = _sobel_pair(x, recons)
    if mode == "signed_mse":
        return F.mse_loss(rx, tx) + F.mse_loss(ry, ty)
    if mode == "angle":
        return F.smooth_l1_loss(torch.atan2(ry, rx), torch.atan2(ty, tx))
    if mode == "cosine":
        gt = F.normalize(torch.stack([tx, ty], dim=1), p=2, dim=1)
        gp = F.normalize(torch.stack([rx, ry], dim=1), p=2, dim=1)
        return 1 - F.cosine_similarity(gp, gt).mean()
    gp = torch.sqrt(rx ** 2 + ry ** 2 + GG_EPS)
    gt = torch.sqrt(tx ** 2 + ty ** 2 + GG_EPS)
    if mode == "maxnorm":
        return F.smooth_l1_loss(gp / (gp.max() + GG_EPS), gt / (gt.max() + GG_EPS))
    if mode == "masked":
        mask = (gt > gt.mean()).float()
        return F.smooth_l1_loss(gp * mask, gt * mask)
    raise ValueError(mode)


GG_VAE_EDGE_MODE = {1: "mag", 2: "maxnorm", 3: "angle", 5: "cosine"}  # models/gg_vae.py:57-63
GG_VQ_VAE_EDGE_MODE = {"v1": None, "v2": "signed_mse", "v3": "mag", "v4": "maxnorm", "v5": "angle", "v6": "masked",
                       "v7": "cosine"}  # models/gg_vq_vae.py:65-88


def losses_gg_vae(x, out, cfg):
    """models/gg_vae.py:222-251 -- order reconstruction, gradient_guided, edge_matching, kld."""
    fn, _ = O.resolve_objective(cfg.get("recons_objective", "mse"), cfg.get("recons_activation"))
    lw = cfg["lambda_weights"]
    r = lw["reconstruction_loss"] * fn(x, out["recons"])
    gg = lw["gradient_guided_loss"] * edge_weighted_pixel_loss(x, out["recons"])
    mode = GG_VAE_EDGE_MODE.get(cfg.get("edge_matching_version", 1), "mag")
    em = lw["edge_matching_loss"] * edge_matching_variant(x, out["recons"], mode)
    k = lw["kld_loss"] * O.kl_divergence(out["mu"], out["log_var"])
    return OrderedDict(reconstruction_loss=r, gradient_guided_loss=gg, edge_matching_loss=em, kld_loss=k, total_loss=r + gg + em + k)


def losses_gg_vq_vae(x, out, cfg):
    """models/gg_vq_vae.py through VQVAE.loss_function (models/vq_vae.py:367-391): objectives order reconstruction,
    embedding, commitment, gradient_guided[, edge_matching for versions v2..v7]."""
    fn, _ = O.resolve_objective(cfg.get("recons_objective", "mse"), cfg.get("recons_activation"))
    lw = cfg["lambda_weights"]
    ld = OrderedDict()
    ld["reconstruction_loss"] = lw["reconstruction_loss"] * fn(x, out["recons"])
    ld["embedding_loss"] = lw["embedding_loss"] * out["embedding_loss"]
    ld["commitment_loss"] = lw["commitment_loss"] * out["commitment_loss"]
    ld["gradient_guided_loss"] = lw["gradient_guided_loss"] * edge_weighted_pixel_loss(x, out["recons"])
    mode = GG_VQ_VAE_EDGE_MODE[cfg.get("version", "v1")]
    if mode is not None:
        ld["edge_matching_loss"] = lw["edge_matching_loss"] * edge_matching_variant(x, out["recons"], mode)
    ld["total_loss"] = sum(ld.values())
    return ld


def losses_gg_vq_vae2(x, out, cfg):
    """models/gg_vq_vae2.py:131-161 -- order reconstruction, commitment, embedding, gradient_guided, edge_matching
    (edge_matching_loss_v2 :118-129 is the arithmetic of models/gg_vae.py:139-156)."""
    fn, _ = O.resolve_objective(cfg.get("recons_objective", "mse"), cfg.get("recons_activation"))
    lw = cfg["lambda_weights"]
    r = lw["reconstruction_loss"] * fn(x, out["recons"])
    c = lw["commitment_loss"] * out["commitment_loss"]
    e = lw["embedding_loss"] * out["embedding_loss"]
    gg = lw["gradient_guided_loss"] * edge_weighted_pixel_loss(x, out["recons"])
    em = lw["edge_matching_loss"] * edge_matching_loss(x, out["recons"])
    return OrderedDict(reconstruction_loss=r, commitment_loss=c, embedding_loss=e, gradient_guided_loss=gg, edge_matching_loss=em,
                       total_loss=r + c + e + gg + em)


# ---------------------------------------------------------------------------------------
# Vector quantiser (models/vq_vae.py:27-64)
# ---------------------------------------------------------------------------------------
def vector_quantize(latents_nchw, E):
    x = latents_nchw.permute(0, 2, 3, 1).contiguous()
    flat = x.reshape(-1, E.shape[1])
    dist = (flat ** 2).sum(dim=1, keepdim=True) + (E ** 2).sum(dim=1) - 2.0 * flat @ E.t()
    idx = torch.argmin(dist, dim=1)  # first minimum on ties
    q = E[idx].reshape(x.shape)  # == one_hot @ E, row selection is exact
    commitment = ((q.detach() - x) ** 2).sum() / x.numel()
    embedding = ((q - x.detach()) ** 2).sum() / x.numel()
    q_st = x + (q - x).detach()
    return q_st.permute(0, 3, 1, 2).contiguous(), commitment, embedding, idx


def usage_percent(idx, K):
    """models/vq_vae.py:110-124."""
    return float(torch.unique(idx).numel() / K * 100.0)


# ---------------------------------------------------------------------------------------
# VQ-VAE (models/vq_vae.py:148-391)
# ---------------------------------------------------------------------------------------
def init_vq_vae(cfg):
    hd, C = list(cfg["hidden_dims"]), cfg.get("in_channels", 3)
    D, K, n = cfg["embedding_dim"], cfg["num_embeddings"], cfg["num_residual_layers"]
    d = _Draw()
    cin, i = C, 0
    for h in hd:
        d.conv(f"encoder.{i}.0", cin, h, 4)
        cin, i = h, i + 1
    d.conv(f"encoder.{i}.0", cin, cin, 3)
    i += 1
    for _ in range(n):
        d.conv(f"encoder.{i}.resblock.0", cin, cin, 3, bias=False)
        d.conv(f"encoder.{i}.resblock.2", cin, cin, 1, bias=False)
        i += 1
    i += 1  # LeakyReLU slot
    d.conv(f"encoder.{i}.0", cin, D, 1)
    d.codebook("vq_layer", K, D)
    j = 0
    d.conv(f"decoder.{j}.0", D, hd[-1], 3)
    j += 1
    for _ in range(n):
        d.conv(f"decoder.{j}.resblock.0", hd[-1], hd[-1], 3, bias=False)
        d.conv(f"decoder.{j}.resblock.2", hd[-1], hd[-1], 1, bias=False)
        j += 1
    j += 1
    rev = hd[::-1]
    for t in range(len(rev) - 1):
        d.convT(f"decoder.{j}.0", rev[t], rev[t + 1], 4)
        j += 1
    d.convT(f"decoder.{j}.0", rev[-1], C, 4)
    return d.sd


def _residual(sd, key, h):
    """models/vq_vae.py:127-145 -- x + conv1x1(relu(conv3x3(x))), bias-free."""
    r = F.conv2d(h, sd[key + ".resblock.0.weight"], None, padding=1)
    r = F.conv2d(F.relu(r), sd[key + ".resblock.2.weight"], None)
    return h + r


def encode_vq_vae(sd, x, cfg):
    hd, n = list(cfg["hidden_dims"]), cfg["num_residual_layers"]
    h, i = x, 0
    for _ in hd:
        h = F.leaky_relu(F.conv2d(h, sd[f"encoder.{i}.0.weight"], sd[f"encoder.{i}.0.bias"], stride=2, padding=1), LRELU)
        i += 1
    h = F.leaky_relu(F.conv2d(h, sd[f"encoder.{i}.0.weight"], sd[f"encoder.{i}.0.bias"], padding=1), LRELU)
    i += 1
    for _ in range(n):
        h = _residual(sd, f"encoder.{i}", h)
        i += 1
    h = F.leaky_relu(h, LRELU)
    i += 1
    return F.leaky_relu(F.conv2d(h, sd[f"encoder.{i}.0.weight"], sd[f"encoder.{i}.0.bias"]), LRELU)


def decode_vq_vae(sd, q, cfg):
    hd, n = list(cfg["hidden_dims"]), cfg["num_residual_layers"]
    j = 0
    h = F.leaky_relu(F.conv2d(q, sd[f"decoder.{j}.0.weight"], sd[f"decoder.{j}.0.bias"], padding=1), LRELU)
    j += 1
    for _ in range(n):
        h = _residual(sd, f"decoder.{j}", h)
        j += 1
    h = F.leaky_relu(h, LRELU)
    j += 1
    for _ in range(len(hd) - 1):
        h = F.leaky_relu(F.conv_transpose2d(h, sd[f"decoder.{j}.0.weight"], sd[f"decoder.{j}.0.bias"], stride=2, padding=1), LRELU)
        j += 1
    h = F.conv_transpose2d(h, sd[f"decoder.{j}.0.weight"], sd[f"decoder.{j}.0.bias"], stride=2, padding=1)
    _, act = O.resolve_objective(cfg.get("recons_objective", "mse"), cfg.get("recons_activation"))
    return O.ACTIVATIONS[act](h)


def forward_vq_vae(sd, x, cfg, eps=None, train=True):
    """models/vq_vae.py:327-365."""
    enc = encode_vq_vae(sd, x, cfg)
    q, commit, embed, idx = vector_quantize(enc, sd["vq_layer.embedding.weight"])
    return {"recons": decode_vq_vae(sd, q, cfg), "quantized_inputs": q, "encoding": enc,
            "commitment_loss": commit, "embedding_loss": embed,
            "codebook_usage_percentage": usage_percent(idx, cfg["num_embeddings"]), "encoding_inds": idx}


def losses_vq_vae(x, out, cfg):
    """models/vq_vae.py:367-391 -- order reconstruction, embedding, commitment."""
    fn, _ = O.resolve_objective(cfg.get("recons_objective", "mse"), cfg.get("recons_activation"))
    lw = cfg["lambda_weights"]
    ld = OrderedDict()
    ld["reconstruction_loss"] = lw["reconstruction_loss"] * fn(x, out["recons"])
    ld["embedding_loss"] = lw["embedding_loss"] * out["embedding_loss"]
    ld["commitment_loss"] = lw["commitment_loss"] * out["commitment_loss"]
    ld["total_loss"] = sum(ld.values())
    return ld


# ---------------------------------------------------------------------------------------
# VQ-VAE-2 (models/vq_vae2.py)
# ---------------------------------------------------------------------------------------
RES_CH = 32  # models/vq_vae2.py:190-212 hard-codes n_res_channel = 32


def _init_enc2(d, key, cin, ch, n, stride):
    """models/vq_vae2.py:30-55."""
    if stride == 4:
        d.conv(f"{key}.blocks.0", cin, ch // 2, 4)
        d.conv(f"{key}.blocks.2", ch // 2, ch, 4)
        d.conv(f"{key}.blocks.4", ch, ch, 3)
        i = 5
    else:
        d.conv(f"{key}.blocks.0", cin, ch // 2, 4)
        d.conv(f"{key}.blocks.2", ch // 2, ch, 3)
        i = 3
    for _ in range(n):
        d.conv(f"{key}.blocks.{i}.conv.1", ch, RES_CH, 3)
        d.conv(f"{key}.blocks.{i}.conv.3", RES_CH, ch, 1)
        i += 1


def _init_dec2(d, key, cin, cout, ch, n, stride):
    """models/vq_vae2.py:61-100."""
    d.conv(f"{key}.blocks.0", cin, ch, 3)
    i = 1
    for _ in range(n):
        d.conv(f"{key}.blocks.{i}.conv.1", ch, RES_CH, 3)
        d.conv(f"{key}.blocks.{i}.conv.3", RES_CH, ch, 1)
        i += 1
    i += 1
    if stride == 4:
        d.convT(f"{key}.blocks.{i}", ch, ch // 2, 4)
        d.convT(f"{key}.blocks.{i + 2}", ch // 2, cout, 4)
    else:
        d.convT(f"{key}.blocks.{i}", ch, cout, 4)


def init_vq_vae2(cfg):
    ch, C = cfg["hidden_dims"][0], cfg.get("in_channels", 3)
    D, K, n = cfg["embedding_dim"], cfg["num_embeddings"], cfg["num_residual_layers"]
    d = _Draw()
    _init_enc2(d, "enc_b", C, ch, n, 4)
    _init_enc2(d, "enc_t", ch, ch, n, 2)
    d.conv("quantize_conv_t", ch, D, 1)
    d.codebook("quantize_t", K, D)
    _init_dec2(d, "dec_t", D, D, ch, n, 2)
    d.conv("quantize_conv_b", D + ch, D, 1)
    d.codebook("quantize_b", K, D)
    d.convT("upsample_t", D, D, 4)
    _init_dec2(d, "dec", 2 * D, C, ch, n, 4)
    sd = OrderedDict()
    for k, v in d.sd.items():
        sd[k] = v
        if k == "quantize_b.embedding.weight":  # aliases registered at models/vq_vae2.py:199-200
            sd["vq_top.embedding.weight"] = sd["quantize_t.embedding.weight"]
            sd["vq_bottom.embedding.weight"] = sd["quantize_b.embedding.weight"]
    return sd


def _conv(sd, key, h, stride=1, padding=0):
    return F.conv2d(h, sd[key + ".weight"], sd[key + ".bias"], stride=stride, padding=padding)


def _convT(sd, key, h):
    return F.conv_transpose2d(h, sd[key + ".weight"], sd[key + ".bias"], stride=2, padding=1)


def _resblock2(sd, key, h):
    """models/vq_vae2.py:13-28."""
    r = _conv(sd, key + ".conv.1", F.relu(h), padding=1)
    r = _conv(sd, key + ".conv.3", F.relu(r))
    return r + h


def _enc2(sd, key, h, n, stride):
    if stride == 4:
        h = F.relu(_conv(sd, f"{key}.blocks.0", h, 2, 1))
        h = F.relu(_conv(sd, f"{key}.blocks.2", h, 2, 1))
        h = _conv(sd, f"{key}.blocks.4", h, 1, 1)
        i = 5
    else:
        h = F.relu(_conv(sd, f"{key}.blocks.0", h, 2, 1))
        h = _conv(sd, f"{key}.blocks.2", h, 1, 1)
        i = 3
    for _ in range(n):
        h = _resblock2(sd, f"{key}.blocks.{i}", h)
        i += 1
    return F.relu(h)


def _dec2(sd, key, h, n, stride):
    h = _conv(sd, f"{key}.blocks.0", h, 1, 1)
    i = 1
    for _ in range(n):
        h = _resblock2(sd, f"{key}.blocks.{i}", h)
        i += 1
    h = F.relu(h)
    i += 1
    if stride == 4:
        h = F.relu(_convT(sd, f"{key}.blocks.{i}", h))
        h = _convT(sd, f"{key}.blocks.{i + 2}", h)
    else:
        h = _convT(sd, f"{key}.blocks.{i}", h)
    return h


def forward_vq_vae2(sd, x, cfg, eps=None, train=True):
    """models/vq_vae2.py:218-282."""
    n, K = cfg["num_residual_layers"], cfg["num_embeddings"]
    enc_b = _enc2(sd, "enc_b", x, n, 4)
    enc_t = _enc2(sd, "enc_t", enc_b, n, 2)
    q_t, c_t, e_t, i_t = vector_quantize(_conv(sd, "quantize_conv_t", enc_t), sd["quantize_t.embedding.weight"])
    dec_t = _dec2(sd, "dec_t", q_t, n, 2)
    q_b, c_b, e_b, i_b = vector_quantize(_conv(sd, "quantize_conv_b", torch.cat([dec_t, enc_b], 1)),
                                         sd["quantize_b.embedding.weight"])
    up = _convT(sd, "upsample_t", q_t)
    h = _dec2(sd, "dec", torch.cat([up, q_b], 1), n, 4)
    _, act = O.resolve_objective(cfg.get("recons_objective", "mse"), cfg.get("recons_activation"))
    return {"recons": O.ACTIVATIONS[act](h), "encoding_top": enc_t, "encoding_bottom": enc_b,
            "quantized_top": q_t, "quantized_bottom": q_b,
            "commitment_loss": c_t + c_b, "embedding_loss": e_t + e_b,
            "codebook_usage_percentage": (usage_percent(i_t, K) + usage_percent(i_b, K)) / 2.0,
            "encoding_inds_top": i_t, "encoding_inds_bottom": i_b}


def losses_vq_vae2(x, out, cfg):
    """models/vq_vae2.py:313-334 -- order reconstruction, commitment, embedding."""
    fn, _ = O.resolve_objective(cfg.get("recons_objective", "mse"), cfg.get("recons_activation"))
    lw = cfg["lambda_weights"]
    ld = OrderedDict()
    ld["reconstruction_loss"] = lw["reconstruction_loss"] * fn(x, out["recons"])
    ld["commitment_loss"] = lw["commitment_loss"] * out["commitment_loss"]
    ld["embedding_loss"] = lw["embedding_loss"] * out["embedding_loss"]
    ld["total_loss"] = ld["reconstruction_loss"] + ld["commitment_loss"] + ld["embedding_loss"]
    return ld


# ---------------------------------------------------------------------------------------
# Beta-TC-VAE (models/betatc_vae.py)
# ---------------------------------------------------------------------------------------
def init_betatc_vae(cfg):
    hd, C, L = list(cfg["hidden_dims"]), cfg.get("in_channels", 3), cfg["latent_dim"]
    sp = cfg["input_size"] // 2 ** len(hd)
    feat = hd[-1] * sp * sp
    d = _Draw()
    cin = C
    for i, h in enumerate(hd):
        d.conv(f"encoder.{i}.0", cin, h, 4)
        cin = h
    d.linear("fc", feat, 256)
    d.linear("fc_mu", 256, L)
    d.linear("fc_var", 256, L)
    d.linear("decoder_input", L, feat)
    rev = hd[::-1]
    for i in range(len(rev) - 1):
        d.convT(f"decoder.{i}.0", rev[i], rev[i + 1], 3)
    d.convT("final_layer.0", rev[-1], rev[-1], 3)
    d.conv("final_layer.2", rev[-1], C, 3)
    return d.sd


def forward_betatc_vae(sd, x, cfg, eps, train=True):
    """models/betatc_vae.py:170-222."""
    hd = list(cfg["hidden_dims"])
    sp = cfg["input_size"] // 2 ** len(hd)
    h = x
    for i in range(len(hd)):
        h = F.leaky_relu(_conv(sd, f"encoder.{i}.0", h, 2, 1), LRELU)
    h = F.linear(h.flatten(1), sd["fc.weight"], sd["fc.bias"])  # no activation after fc
    mu = F.linear(h, sd["fc_mu.weight"], sd["fc_mu.bias"])
    log_var = F.linear(h, sd["fc_var.weight"], sd["fc_var.bias"])
    z = eps * torch.exp(0.5 * log_var) + mu
    h = F.linear(z, sd["decoder_input.weight"], sd["decoder_input.bias"]).view(-1, hd[-1], sp, sp)
    for i in range(len(hd) - 1):
        k = f"decoder.{i}.0"
        h = F.leaky_relu(F.conv_transpose2d(h, sd[k + ".weight"], sd[k + ".bias"], stride=2, padding=1, output_padding=1), LRELU)
    h = F.leaky_relu(F.conv_transpose2d(h, sd["final_layer.0.weight"], sd["final_layer.0.bias"], stride=2, padding=1, output_padding=1), LRELU)
    h = _conv(sd, "final_layer.2", h, 1, 1)
    _, act = O.resolve_objective(cfg.get("recons_objective", "mse"), cfg.get("recons_activation"))
    return {"recons": O.ACTIVATIONS[act](h), "input": x, "mu": mu, "log_var": log_var, "z": z}


def _log_density_gaussian(x, mu, logvar):
    """models/betatc_vae.py:224-234."""
    return -0.5 * (math.log(2 * math.pi) + logvar) - 0.5 * ((x - mu) ** 2 * torch.exp(-logvar))


def log_importance_weights(B, dataset_size):
    """models/betatc_vae.py:273-289 -- fp32 matrix, Python-float scalars."""
    M_N = B / dataset_size
    ds = (1 / M_N) * B
    strat = (ds - B + 1) / (ds * (B - 1))
    W = torch.full((B, B), 1 / (B - 1), dtype=torch.float32)
    W.view(-1)[::B] = 1 / ds
    W.view(-1)[1::B] = strat
    W[B - 2, 0] = strat
    return W.log()


def losses_betatc_vae(x, out, cfg, train=True):
    """models/betatc_vae.py:236-324.  ``cfg['num_iter']`` plays the class-level counter
    (betatc_vae.py:13,298-300)."""
    fn, _ = O.resolve_objective(cfg.get("recons_objective", "mse"), cfg.get("recons_activation"))
    mu, lv, z = out["mu"], out["log_var"], out["z"]
    B, D = z.shape
    rec = fn(x, out["recons"])
    log_q_zx = _log_density_gaussian(z, mu, lv).sum(dim=1)
    zeros = torch.zeros_like(z)
    log_p_z = _log_density_gaussian(z, zeros, zeros).sum(dim=1)
    mat = _log_density_gaussian(z.view(B, 1, D), mu.view(1, B, D), lv.view(1, B, D))
    mat = mat + log_importance_weights(B, cfg["dataset_size"]).view(B, B, 1)
    log_q_z = torch.logsumexp(mat.sum(2), dim=1)
    log_prod_q_z = torch.logsumexp(mat, dim=1).sum(1)
    mi = (log_q_zx - log_q_z).mean()
    tc = (log_q_z - log_prod_q_z).mean()
    kld = (log_prod_q_z - log_p_z).mean()
    if train:
        cfg["num_iter"] = cfg.get("num_iter", 0) + 1
        anneal = min(0 + 1 * cfg["num_iter"] / cfg["anneal_steps"], 1)
    else:
        anneal = 1.0
    lw = cfg["lambda_weights"]
    ld = OrderedDict()
    ld["reconstruction_loss"] = lw["reconstruction_loss"] * rec
    ld["mi_loss"] = lw["mi_loss"] * mi
    ld["tc_loss"] = lw["tc_loss"] * 1 * tc
    ld["kld"] = lw["kld"] * 1 * anneal * kld
    ld["total_loss"] = ld["reconstruction_loss"] + ld["mi_loss"] + ld["tc_loss"] + ld["kld"]
    return ld


# ---------------------------------------------------------------------------------------
# registry + factory defaults (models/__init__.py:18-211)
# ---------------------------------------------------------------------------------------
ARCHS = {
    "vae": dict(init=init_vae, forward=forward_vae, losses=losses_vae, features=["mu", "log_var"],
                needs_eps=True, eps_dim="latent_dim"),
    "gg_vae": dict(init=init_gg_vae, forward=forward_vae, losses=losses_gg_vae, features=["mu", "log_var"],
                   needs_eps=True, eps_dim="latent_dim"),
    "vq_vae": dict(init=init_vq_vae, forward=forward_vq_vae, losses=losses_vq_vae, features=["encoding"],
                   needs_eps=False),
    "gg_vq_vae": dict(init=lambda cfg: _with_sobel(init_vq_vae(cfg)), forward=lambda *a, **k: forward_vq_vae(*a, **k),
                      losses=losses_gg_vq_vae, features=["encoding"], needs_eps=False),
    "vq_vae2": dict(init=init_vq_vae2, forward=forward_vq_vae2, losses=losses_vq_vae2,
                    features=["encoding_top", "encoding_bottom"], needs_eps=False),
    "gg_vq_vae2": dict(init=lambda cfg: _with_sobel(init_vq_vae2(cfg)), forward=lambda *a, **k: forward_vq_vae2(*a, **k),
                       losses=losses_gg_vq_vae2, features=["encoding_top", "encoding_bottom"], needs_eps=False),
    "betatc_vae": dict(init=init_betatc_vae, forward=forward_betatc_vae, losses=losses_betatc_vae,
                       features=["mu", "log_var"], needs_eps=True, eps_dim="latent_dim"),
}


def default_lambda_weights(arch, batch_size, dataset_size):
    r = batch_size / dataset_size
    return {
        "vae": {"reconstruction_loss": 1.0, "kld_loss": r},
        "gg_vae": {"reconstruction_loss": 1.0, "kld_loss": r, "gradient_guided_loss": 1.0, "edge_matching_loss": 1.0},
        "vq_vae": {"reconstruction_loss": 1.0, "embedding_loss": 1.0, "commitment_loss": 0.25},
        "gg_vq_vae": {"reconstruction_loss": 1.0, "gradient_guided_loss": 1.0, "embedding_loss": 1.0, "commitment_loss": 0.25},
        "vq_vae2": {"reconstruction_loss": 1.0, "commitment_loss": 1.0, "embedding_loss": 0.25},
        "gg_vq_vae2": {"reconstruction_loss": 1.0, "commitment_loss": 1.0, "embedding_loss": 0.25, "gradient_guided_loss": 1.0,
                       "edge_matching_loss": 1.0},
        "betatc_vae": {"reconstruction_loss": 1.0, "mi_loss": 1.0, "tc_loss": 1.0, "kld": r},
    }[arch]


def canonical_arch(arch):
    """models/__init__.py:155-178: `gg_vae_vN` / `gg_vq_vae_vN` select one class plus a version argument."""
    if arch.startswith("gg_vae_v"):
        return "gg_vae", dict(edge_matching_version=int(arch[len("gg_vae_v"):]))
    if arch.startswith("gg_vq_vae_v"):
        return "gg_vq_vae", dict(version=arch[len("gg_vq_vae_"):])
    return arch, {}


def make_cfg(arch, input_size, batch_size, dataset_size, **kw):
    arch, extra = canonical_arch(arch)
    kw = dict(extra, **kw)
    if arch == "gg_vq_vae" and kw.get("version", "v1") != "v1" and "lambda_weights" not in kw:
        kw["lambda_weights"] = dict(default_lambda_weights(arch, batch_size, dataset_size), edge_matching_loss=1.0)
    cfg = dict(arch=arch, input_size=input_size, in_channels=3, batch_size=batch_size,
               dataset_size=dataset_size, recons_objective="mse", recons_activation=None)
    cfg.update(kw)
    cfg.setdefault("lambda_weights", default_lambda_weights(arch, batch_size, dataset_size))
    if arch == "betatc_vae":
        cfg.setdefault("anneal_steps", 200)
        cfg.setdefault("num_iter", 0)
    return cfg


def init_state(cfg, seed=None):
    if seed is not None:
        torch.manual_seed(seed)
    return ARCHS[cfg["arch"]]["init"](cfg)
