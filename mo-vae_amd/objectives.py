"""Host-side mirror of the reference's utils/objectives.py (same function names, argument order
`fn(inputs, recons)`, error behaviour) on the HIP loss kernels (csrc/losses.hip)."""
from . import ops

_VALID = {"mse", "bce", "l1", "smooth_l1", "perceptual"}


def _pair(inputs, recons):
    """Both operands in one memory order (NHWC) without copies when they already are."""
    if inputs.dim() == 4:
        return ops.to_nhwc(inputs), ops.to_nhwc(recons)
    return inputs, recons


def _make(kind):
    def fn(inputs, recons, scale=1.0, out_act=False):
        """out_act: the caller vouches that this loss is the ONE reader of `recons` on the tape (the plain VQ-VAE / VQ-VAE-2 / BetaTC
        models): when `recons` is the output of a conv + tanh / sigmoid pair, the loss's backward kernel applies the activation's
        derivative itself (ops.ReconLoss, ops.OUT_ACT_LINKS)."""
        x, r = _pair(inputs, recons)
        return ops.recon_loss(r, x, kind, scale, ops.out_act_link(r) if out_act else None)

    fn.kind = kind
    return fn


mse_per_pixel_mean = _make("mse")            # utils/objectives.py:95-97
bce_per_pixel_mean = _make("bce")            # utils/objectives.py:108-110
laplacian_per_pixel_mean = _make("l1")       # utils/objectives.py:129-131
smooth_l1_per_pixel_mean = _make("smooth_l1")  # utils/objectives.py:134-136


def edge_weighted_pixel_loss(inputs, recons, scale=1.0):
    """models/gg_vae.py:125-137 (GGVAE.edge_weighted_pixel_loss)."""
    x, r = _pair(inputs, recons)
    return ops.edge_weighted_pixel_loss(r, x, scale)


def edge_matching_loss(inputs, recons, scale=1.0, mode="mag"):
    """The edge-matching family of models/gg_vae.py:139-208 and models/gg_vq_vae.py:172-264; `mode` names the variant
    (mag = GGVAE version 1 / GGVQVAE edge_matching_loss_v2, the default)."""
    x, r = _pair(inputs, recons)
    return ops.edge_matching_loss(r, x, scale, mode)


def make_edge_matching(mode):
    """An objective `fn(inputs, recons, scale=1.0)` bound to one edge-matching variant."""
    def fn(inputs, recons, scale=1.0):
        return edge_matching_loss(inputs, recons, scale, mode)

    fn.mode = mode
    return fn


def kl_divergence(mu, log_var, scale=1.0):
    """utils/objectives.py:141-144."""
    return ops.kl_divergence(mu, log_var, scale)


def get_recon_obj_and_activation(recons_objective, recons_activation="tanh", model=None, use_logits=False):
    """utils/objectives.py:6-43 -> (recon_fn, recons_activation)."""
    recons_objective = recons_objective.lower()
    if recons_objective not in _VALID:
        raise ValueError(f"recons_objective must be one of {_VALID}, got {recons_objective}")
    if recons_objective == "mse":
        return mse_per_pixel_mean, recons_activation or "tanh"
    if recons_objective == "bce":
        if use_logits:
            raise NotImplementedError("bce-with-logits is not on the MI355X hot path (no reference arch enables it)")
        return bce_per_pixel_mean, "sigmoid"
    if recons_objective == "l1":
        return laplacian_per_pixel_mean, recons_activation or "tanh"
    if recons_objective == "smooth_l1":
        return smooth_l1_per_pixel_mean, recons_activation or "tanh"
    raise NotImplementedError(
        "recons_objective='perceptual' needs pretrained VGG16 weights (network fetch) and is outside the hot path")
