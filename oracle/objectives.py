"""Oracle: reconstruction objectives and KL term (TEST INFRASTRUCTURE, see oracle/__init__.py).

Written out element-wise instead of through torch.nn.functional so that the arithmetic being
pinned is visible.  Follows utils/objectives.py.
"""
import torch


def resolve_objective(recons_objective, recons_activation=None):
    """utils/objectives.py:6-43 -- objective name -> (loss fn, output activation name)."""
    name = recons_objective.lower()
    if name == "mse":
        return mse, recons_activation or "tanh"
    if name == "bce":
        return bce, "sigmoid"  # forced, whatever was asked (objectives.py:26-27)
    if name == "l1":
        return l1, recons_activation or "tanh"
    if name == "smooth_l1":
        return smooth_l1, recons_activation or "tanh"
    raise ValueError(f"recons_objective {recons_objective!r} not supported by the oracle")


def mse(inputs, recons):
    """utils/objectives.py:95-97 -- mean over every element of (r - x)^2."""
    d = recons - inputs
    return (d * d).sum() / d.numel()


def l1(inputs, recons):
    """utils/objectives.py:129-131."""
    return (recons - inputs).abs().sum() / inputs.numel()


def smooth_l1(inputs, recons):
    """utils/objectives.py:134-136 -- Huber with beta = 1."""
    d = (recons - inputs).abs()
    v = torch.where(d < 1.0, 0.5 * d * d, d - 0.5)
    return v.sum() / v.numel()


def bce(inputs, recons):
    """utils/objectives.py:108-110 -- ATen clamps both logs at -100."""
    lg = torch.log(recons).clamp(min=-100.0)
    lg1 = torch.log(1.0 - recons).clamp(min=-100.0)
    v = -(inputs * lg + (1.0 - inputs) * lg1)
    return v.sum() / v.numel()


def kl_divergence(mu, log_var):
    """utils/objectives.py:141-144 -- mean_b( -1/2 sum_d (1 + lv - mu^2 - e^lv) )."""
    per_sample = -0.5 * (1.0 + log_var - mu * mu - torch.exp(log_var)).sum(dim=1)
    return per_sample.sum() / per_sample.numel()


ACTIVATIONS = {
    "tanh": torch.tanh,
    "sigmoid": torch.sigmoid,
    "none": lambda t: t,
}
