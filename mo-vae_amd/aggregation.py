"""Gradient aggregators with torchjd's call protocol (what main.py:1191-1250 constructs and hooks):

    aggregator(J: Tensor[K, m]) -> Tensor[m]
    aggregator.weighting        -> nn.Module, forward(J) -> w[K]   (forward hooks see (J,), w)
    MGDA.set_losses(Tensor[K])

Everything numeric runs in libmovae_hip.so on the current stream: Gramian (one HBM pass over J),
the K x K solve in a single-wave fp64 kernel, combine (second pass).  No host synchronisation --
the reference's UPGrad round trip to numpy/quadprog and MGDA's per-iteration `.item()` are gone.
"""
import os

import torch
import torch.nn as tnn

from . import _lib as L


def _st(t):
    return L.stream_ptr(t.device)


def _check_matrix(J):
    if J.dim() != 2:
        raise ValueError(f"Parameter `matrix` should be a tensor of dimension 2. Found `matrix.shape = {tuple(J.shape)}`.")
    if J.shape[0] > L.MAX_K:
        raise ValueError(f"at most {L.MAX_K} objectives are supported on device, got {J.shape[0]}")
    L.require_gpu(J)
    if J.dtype != torch.float32 or J.stride(1) != 1:
        raise ValueError("the Jacobian must be fp32 with unit column stride")


#: MOVAE_FUSE_GRAM=0: the Gramian is finished by its own kernel before the weighting runs (A/B knob)
FUSE_GRAM = os.environ.get("MOVAE_FUSE_GRAM", "1") != "0"


def compute_gramian(J):
    """G = J J^T (torchjd GramianWeightedAggregator), fp32 [K, K]."""
    _check_matrix(J)
    k, m = J.shape
    G = torch.empty((k, k), dtype=torch.float32, device=J.device)
    ws = L.workspace(J.device)
    L.call("movae_gram", J.data_ptr(), J.stride(0) if k > 1 else max(J.stride(0), m), k, m, G.data_ptr(),
                                ws.data_ptr(), ws.numel(), _st(J))
    return G


def combine(J, w):
    """g = w @ J."""
    _check_matrix(J)
    k, m = J.shape
    g = torch.empty(m, dtype=torch.float32, device=J.device)
    L.call("movae_combine", J.data_ptr(), J.stride(0) if k > 1 else max(J.stride(0), m), k, m, w.data_ptr(),
                                   g.data_ptr(), 0, _st(J))
    return g


def gd_similarity(J, w):
    """cos(J.T @ w, J.mean(0)) as a device scalar (the quantity main.py:108-119 logs)."""
    _check_matrix(J)
    k, m = J.shape
    out = torch.empty((), dtype=torch.float32, device=J.device)
    ws = L.workspace(J.device)
    L.call("movae_gd_similarity", J.data_ptr(), J.stride(0) if k > 1 else max(J.stride(0), m), k, m, w.data_ptr(),
                                         out.data_ptr(), ws.data_ptr(), ws.numel(), _st(J))
    return out


def _pref_tensor(pref, device):
    if pref is None:
        return None
    return torch.as_tensor(pref, dtype=torch.float32).to(device).contiguous()


class Weighting(tnn.Module):
    """Maps a PSD Gramian [K, K] to weights [K]."""

    def forward(self, gramian):  # pragma: no cover - abstract
        raise NotImplementedError


class _FromJacobian(tnn.Module):
    """`weighting << compute_gramian`: what torchjd exposes as ``aggregator.weighting``."""

    def __init__(self, gramian_weighting):
        super().__init__()
        self.gramian_weighting = gramian_weighting

    def forward(self, J):
        gw = self.gramian_weighting
        fused = getattr(gw, "from_jacobian", None)
        if fused is not None and FUSE_GRAM and not gw._forward_hooks and not gw._forward_pre_hooks:
            # the weighting's kernel folds the Gramian's partial sums itself (one launch fewer; same G, same weights)
            self.last_gramian, w = fused(J)
            return w
        self.last_gramian = compute_gramian(J)  # kept for aggregators that derive a second weighting from it (COMFORT)
        return gw(self.last_gramian)


class GramianWeightedAggregator(tnn.Module):
    def __init__(self, gramian_weighting):
        super().__init__()
        self.gramian_weighting = gramian_weighting
        self.weighting = _FromJacobian(gramian_weighting)

    def forward(self, J):
        w = self.weighting(J)
        return combine(J, w)


# ---- UPGrad --------------------------------------------------------------------------------------
class UPGradWeighting(Weighting):
    """Projection of every weighted row onto the dual cone of all rows.  `norm` selects the Gramian normalisation:
    "trace" (torchjd UPGrad), "min_l2" (NUPGrad), "cosine" (PNUPGrad's second branch)."""

    def __init__(self, pref_vector=None, norm_eps=1e-4, reg_eps=1e-4, norm="trace"):
        super().__init__()
        self.pref_vector, self.norm_eps, self.reg_eps, self.norm = pref_vector, norm_eps, reg_eps, norm

    def _mode(self):
        return self.norm

    def forward(self, G):
        k = G.shape[0]
        w = torch.empty(k, dtype=torch.float32, device=G.device)
        pref = _pref_tensor(self.pref_vector, G.device)
        L.call("movae_weights_upgrad_norm", G.data_ptr(), k, L.UPGRAD_NORM[self._mode()], float(self.norm_eps),
               float(self.reg_eps), L.ptr(pref), w.data_ptr(), _st(G))
        return w

    def from_jacobian(self, J):
        """(G, w) = (J J^T, forward(G)) in two launches (movae_gram_upgrad): the solver kernel finishes the Gramian itself."""
        _check_matrix(J)
        k, m = J.shape
        G = torch.empty((k, k), dtype=torch.float32, device=J.device)
        w = torch.empty(k, dtype=torch.float32, device=J.device)
        pref = _pref_tensor(self.pref_vector, J.device)
        ws = L.workspace(J.device)
        L.call("movae_gram_upgrad", J.data_ptr(), J.stride(0) if k > 1 else max(J.stride(0), m), k, m, G.data_ptr(),
               L.UPGRAD_NORM[self._mode()], float(self.norm_eps), float(self.reg_eps), L.ptr(pref), w.data_ptr(), 0, ws.data_ptr(),
               ws.numel(), _st(J))
        return G, w


class _PNUPGradWeighting(UPGradWeighting):
    """utils/torchmoo/pnupgrad.py:127-134: with probability `prob` the cosine-normalised Gramian, else the min-L2 one.
    The coin is torch's CPU generator, as in the reference (`torch.rand(1).item()`), so a seeded run makes the same
    sequence of choices; it is drawn on the host per call and therefore not replayable from a captured hipGraph."""

    def __init__(self, pref_vector=None, prob=0.5, norm_eps=1e-4, reg_eps=1e-4):
        super().__init__(pref_vector, norm_eps, reg_eps, norm="min_l2")
        self.prob = prob

    def _mode(self):
        return "cosine" if torch.rand(1).item() < self.prob else "min_l2"


class UPGrad(GramianWeightedAggregator):
    """torchjd.aggregation.UPGrad(pref_vector, norm_eps, reg_eps) as constructed at main.py:1195."""

    def __init__(self, pref_vector=None, norm_eps=0.0001, reg_eps=0.0001, solver="quadprog"):
        super().__init__(UPGradWeighting(pref_vector, norm_eps, reg_eps))
        self._pref_vector, self._norm_eps, self._reg_eps = pref_vector, norm_eps, reg_eps

    def __repr__(self):
        return f"UPGrad(pref_vector={self._pref_vector!r}, norm_eps={self._norm_eps}, reg_eps={self._reg_eps})"


class NUPGrad(GramianWeightedAggregator):
    """utils/torchmoo/nupgrad.py:37-89 as constructed at main.py:1226."""

    def __init__(self, pref_vector=None, norm_eps=0.0001, reg_eps=0.0001, solver="quadprog"):
        super().__init__(UPGradWeighting(pref_vector, norm_eps, reg_eps, norm="min_l2"))
        self._pref_vector, self._norm_eps, self._reg_eps = pref_vector, norm_eps, reg_eps

    def __repr__(self):
        return f"NUPGrad(pref_vector={self._pref_vector!r}, norm_eps={self._norm_eps}, reg_eps={self._reg_eps}, solver='quadprog')"


class PNUPGrad(GramianWeightedAggregator):
    """utils/torchmoo/pnupgrad.py:36-95 as constructed at main.py:1228."""

    def __init__(self, pref_vector=None, prob=0.5, norm_eps=0.0001, reg_eps=0.0001, solver="quadprog"):
        super().__init__(_PNUPGradWeighting(pref_vector, prob, norm_eps, reg_eps))
        self._pref_vector, self._prob, self._norm_eps, self._reg_eps = pref_vector, prob, norm_eps, reg_eps

    def __repr__(self):
        return (f"PNUPGrad(pref_vector={self._pref_vector!r}, prob={self._prob}, norm_eps={self._norm_eps}, "
                f"reg_eps={self._reg_eps}, solver='quadprog')")


# ---- MGDA (utils/torchmoo/mgda.py) -------------------------------------------------------------------
class MGDAWeighting(Weighting):
    def __init__(self, norm_type="none", epsilon=1e-5, max_iters=250, stable=False, min_eigenvalue_eps=1e-10):
        super().__init__()
        if norm_type not in ("none", "l2", "loss", "loss+"):
            raise ValueError("Parameter `norm_type` should be 'none', 'l2', 'loss', or 'loss+'. Found "
                             f"`norm_type = {norm_type!r}`.")
        self.norm_type, self.epsilon, self.max_iters = norm_type, epsilon, max_iters
        self.stable, self.min_eigenvalue_eps = stable, min_eigenvalue_eps
        self._losses = None
        self._info = None

    def set_losses(self, losses):
        if losses.dim() != 1:
            raise ValueError(f"Parameter `losses` should be a 1D tensor. Found `losses.shape = {losses.shape}`.")
        self._losses = losses.detach()

    @property
    def convergence_count(self):
        """iterations used by the last solve (host read on demand only)."""
        return None if self._info is None else int(self._info.item())

    def forward(self, G):
        k = G.shape[0]
        losses = None
        if self.norm_type in ("loss", "loss+"):
            if self._losses is None:
                raise RuntimeError(f"Losses must be set before calling forward() when using norm_type='{self.norm_type}'. "
                                   "Call set_losses() first.")
            if self._losses.shape[0] != k:
                raise ValueError(f"Number of losses ({self._losses.shape[0]}) must match the number of rows in the gramian ({k}).")
            losses = self._losses.to(device=G.device, dtype=torch.float32).contiguous()
        w = torch.empty(k, dtype=torch.float32, device=G.device)
        self._info = torch.empty(1, dtype=torch.int32, device=G.device)
        if self.stable:  # StableMGDA: eigenvalues clamped from below before the Frank-Wolfe iteration (mgda.py:286-317)
            L.call("movae_weights_mgda_stable", G.data_ptr(), k, L.MGDA_NORM[self.norm_type], L.ptr(losses), float(self.epsilon),
                   int(self.max_iters), float(self.min_eigenvalue_eps), w.data_ptr(), self._info.data_ptr(), _st(G))
        else:
            L.call("movae_weights_mgda", G.data_ptr(), k, L.MGDA_NORM[self.norm_type], L.ptr(losses), float(self.epsilon),
                   int(self.max_iters), w.data_ptr(), self._info.data_ptr(), _st(G))
        return w


class StableMGDA(GramianWeightedAggregator):
    """utils/torchmoo/mgda.py:139-153: MGDA with eigen regularisation always on."""

    def __init__(self, norm_type="none", epsilon=1e-5, max_iters=250, min_eigenvalue_eps=1e-10):
        w = MGDAWeighting(norm_type, epsilon, max_iters, True, min_eigenvalue_eps)
        super().__init__(w)
        self._mgda_weighting = w

    @property
    def mgda_weighting(self):
        return self._mgda_weighting

    def set_losses(self, losses):
        self._mgda_weighting.set_losses(losses)


class MGDA(GramianWeightedAggregator):
    def __init__(self, norm_type="none", epsilon=1e-5, max_iters=250, stable=False, min_eigenvalue_eps=1e-10):
        w = MGDAWeighting(norm_type, epsilon, max_iters, stable, min_eigenvalue_eps)
        super().__init__(w)
        self._mgda_weighting = w
        self._norm_type, self._epsilon, self._max_iters, self._stable = norm_type, epsilon, max_iters, stable

    @property
    def mgda_weighting(self):
        return self._mgda_weighting

    def set_losses(self, losses):
        self._mgda_weighting.set_losses(losses)

    def __repr__(self):
        return (f"MGDA(norm_type={self._norm_type!r}, epsilon={self._epsilon}, max_iters={self._max_iters}, "
                f"stable={self._stable})")


# ---- Aligned-MTL (utils/torchmoo/aligned_mtl.py) --------------------------------------------------------
class AlignedMTLWeighting(Weighting):
    def __init__(self, pref_vector=None, scale_mode="min"):
        super().__init__()
        self._pref_vector, self._scale_mode = pref_vector, scale_mode

    def forward(self, G):
        if self._scale_mode not in L.AMTL_SCALE:
            raise ValueError(f"Invalid scale_mode={self._scale_mode!r}. Expected 'min', 'median', or 'rmse'.")
        k = G.shape[0]
        w = torch.empty(k, dtype=torch.float32, device=G.device)
        pref = _pref_tensor(self._pref_vector, G.device)
        L.call("movae_weights_amtl", G.data_ptr(), k, L.AMTL_SCALE[self._scale_mode], L.ptr(pref), w.data_ptr(), _st(G))
        return w


class AlignedMTL(GramianWeightedAggregator):
    def __init__(self, pref_vector=None, scale_mode="min"):
        super().__init__(AlignedMTLWeighting(pref_vector, scale_mode))
        self._pref_vector, self._scale_mode = pref_vector, scale_mode

    def __repr__(self):
        return f"AlignedMTL(pref_vector={self._pref_vector!r}, scale_mode={self._scale_mode!r})"


# ---- constant weightings (torchjd Mean / Sum) ---------------------------------------------------------------
class _ConstWeighting(Weighting):
    def __init__(self, mean):
        super().__init__()
        self.mean = mean

    def forward(self, G):
        k = G.shape[0]
        w = torch.empty(k, dtype=torch.float32, device=G.device)
        L.call("movae_weights_const", k, 1.0 / k if self.mean else 1.0, w.data_ptr(), _st(G))
        return w


class Mean(GramianWeightedAggregator):
    def __init__(self):
        super().__init__(_ConstWeighting(True))


class Sum(GramianWeightedAggregator):
    def __init__(self):
        super().__init__(_ConstWeighting(False))


# ---- torchjd's other Gramian weightings offered by main.py:1196-1222 ---------------------------------------------------
# torchjd is a third-party dependency pinned to "main" (requirements.txt:58) and absent from the reference tree; these
# follow the published algorithms (oracle/aggregation.py restates them too) -- parity unpinned.
class DualProjWeighting(Weighting):
    def __init__(self, pref_vector=None, norm_eps=1e-4, reg_eps=1e-4):
        super().__init__()
        self.pref_vector, self.norm_eps, self.reg_eps = pref_vector, norm_eps, reg_eps

    def forward(self, G):
        k = G.shape[0]
        w = torch.empty(k, dtype=torch.float32, device=G.device)
        pref = _pref_tensor(self.pref_vector, G.device)
        L.call("movae_weights_dualproj", G.data_ptr(), k, float(self.norm_eps), float(self.reg_eps), L.ptr(pref), w.data_ptr(), _st(G))
        return w


class DualProj(GramianWeightedAggregator):
    """torchjd.aggregation.DualProj(pref_vector, norm_eps, reg_eps) as constructed at main.py:1220-1221."""

    def __init__(self, pref_vector=None, norm_eps=0.0001, reg_eps=0.0001, solver="quadprog"):
        super().__init__(DualProjWeighting(pref_vector, norm_eps, reg_eps))
        self._pref_vector, self._norm_eps, self._reg_eps = pref_vector, norm_eps, reg_eps

    def __repr__(self):
        return f"DualProj(pref_vector={self._pref_vector!r}, norm_eps={self._norm_eps}, reg_eps={self._reg_eps}, solver='quadprog')"


class PCGradWeighting(Weighting):
    """The K task orders are drawn with torch.randperm on the host generator, one per task per call, exactly where the
    reference draws them -- a seeded run sees the same orders; like PNUPGrad it is therefore not hipGraph-replayable."""

    def forward(self, G):
        k = G.shape[0]
        perm = torch.stack([torch.randperm(k) for _ in range(k)]).to(dtype=torch.int32).to(G.device)
        w = torch.empty(k, dtype=torch.float32, device=G.device)
        L.call("movae_weights_pcgrad", G.data_ptr(), k, perm.data_ptr(), w.data_ptr(), _st(G))
        return w


class PCGrad(GramianWeightedAggregator):
    """torchjd.aggregation.PCGrad() as constructed at main.py:1196-1197."""

    def __init__(self):
        super().__init__(PCGradWeighting())

    def __repr__(self):
        return "PCGrad()"


class IMTLGWeighting(Weighting):
    def forward(self, G):
        k = G.shape[0]
        w = torch.empty(k, dtype=torch.float32, device=G.device)
        L.call("movae_weights_imtlg", G.data_ptr(), k, w.data_ptr(), _st(G))
        return w


class IMTLG(GramianWeightedAggregator):
    """torchjd.aggregation.IMTLG() as constructed at main.py:1207-1208."""

    def __init__(self):
        super().__init__(IMTLGWeighting())

    def __repr__(self):
        return "IMTLG()"


class CAGradWeighting(Weighting):
    def __init__(self, c, norm_eps=1e-4):
        super().__init__()
        if c < 0.0:
            raise ValueError(f"Parameter `c` should be a non-negative float. Found `c = {c}`.")
        self.c, self.norm_eps = c, norm_eps

    def forward(self, G):
        k = G.shape[0]
        w = torch.empty(k, dtype=torch.float32, device=G.device)
        L.call("movae_weights_cagrad", G.data_ptr(), k, float(self.c), float(self.norm_eps), w.data_ptr(), _st(G))
        return w


class CAGrad(GramianWeightedAggregator):
    """torchjd.aggregation.CAGrad(c, norm_eps) as constructed at main.py:1216-1217 (cvxpy / CLARABEL there; a closed-form
    KKT enumeration on the device here -- see csrc/agg.hip)."""

    def __init__(self, c, norm_eps=0.0001):
        super().__init__(CAGradWeighting(c, norm_eps))
        self._c, self._norm_eps = c, norm_eps

    def __repr__(self):
        return f"CAGrad(c={self._c}, norm_eps={self._norm_eps})"


def beta_schedule(epoch, total_epochs, k=1.0, a=1.0, l=0.01, u=1.0):
    """utils/torchmoo/comfort.py:20-66."""
    import math

    if total_epochs <= 1:
        return u
    progress = (epoch - 1) / (total_epochs - 1)
    progress = min(1.0, max(0.0, progress)) ** a
    f = progress if k <= 0 else (1.0 - math.exp(-k * progress)) / (1.0 - math.exp(-k))
    beta = l + (u - l) * f
    return float(min(u, max(l, beta)))


class COMFORT:
    """utils/torchmoo/comfort.py:68-165: g = (1 - beta) g_MGDA + beta g_UPGrad with beta following `beta_schedule` per
    epoch (`set_epoch`, main.py:1290-1291).  Both weight vectors come from ONE Gramian and the blend happens on the K
    weights, so the Jacobian is read twice (Gram, combine) instead of four times; `weighting` is the MGDA weighting, as
    in the reference (hooks see J and the MGDA weights)."""

    def __init__(self, mgda_norm_type="none", mgda_stable=False, mgda_epsilon=1e-5, mgda_max_iters=250,
                 mgda_min_eigenvalue_eps=1.0, beta_k=1.0, beta_a=1.0, beta_l=0.01, beta_u=1.0):
        self._mgda = MGDA(norm_type=mgda_norm_type, epsilon=mgda_epsilon, max_iters=mgda_max_iters, stable=mgda_stable,
                          min_eigenvalue_eps=mgda_min_eigenvalue_eps)
        self._upgrad = UPGrad()
        self._beta_k, self._beta_a, self._beta_l, self._beta_u = beta_k, beta_a, beta_l, beta_u
        self._current_epoch, self._total_epochs = 1, 1
        self._norm_type = mgda_norm_type
        self.weighting = self._mgda.weighting

    def set_epoch(self, epoch, total_epochs):
        self._current_epoch, self._total_epochs = epoch, total_epochs

    def set_losses(self, losses):
        self._mgda.set_losses(losses)

    def _get_beta(self):
        return beta_schedule(self._current_epoch, self._total_epochs, k=self._beta_k, a=self._beta_a, l=self._beta_l, u=self._beta_u)

    def __call__(self, J):
        w_m = self.weighting(J)                                        # through the module: forward hooks fire
        w_u = self._upgrad.gramian_weighting(self.weighting.last_gramian)
        beta = self._get_beta()
        w = torch.empty_like(w_m)
        L.call("movae_axpby", 1.0 - beta, w_m.data_ptr(), beta, w_u.data_ptr(), w.data_ptr(), w.numel(), _st(w))
        return combine(J, w)

    def __repr__(self):
        return (f"COMFORT(mgda_norm_type={self._norm_type!r}, mgda_stable={self._mgda._stable}, beta_k={self._beta_k}, "
                f"beta_a={self._beta_a}, beta_l={self._beta_l}, beta_u={self._beta_u})")


OUT_OF_SCOPE = ("nashmtl",)  # stateful, solves a cvxpy problem (ECOS) every `update_weights_every` steps


def make_aggregator(args):
    """Aggregator factory with the reference's names and flags (main.py:1191-1246).
    Returns None / "sum" / an aggregator object, and normalises args.aggregator like the reference."""
    if args.aggregator is None:
        args.aggregator = "sum"
        return None
    name = args.aggregator.lower()
    pref = getattr(args, "pref_weights", None)
    if name == "upgrad":
        return UPGrad(norm_eps=args.agg_norm_eps, reg_eps=args.agg_reg_eps, pref_vector=pref)
    if name == "mean":
        return Mean()
    if name in ("aligned_mtl", "aligned_mtl_min", "amtl", "amtl_min"):
        args.aggregator = "aligned_mtl"
        return AlignedMTL(pref_vector=pref)
    if name == "aligned_mtl_median":
        return AlignedMTL(scale_mode="median", pref_vector=pref)
    if name == "aligned_mtl_rmse":
        return AlignedMTL(scale_mode="rmse", pref_vector=pref)
    mg = {"mgda": "none", "mgda_ln": "l2", "mgda_gn": "loss", "mgda_lgn": "loss+"}
    if name in mg:
        return MGDA(epsilon=args.mgda_epsilon, max_iters=args.mgda_max_iters, norm_type=mg[name])
    if name == "jd_sum":
        return Sum()
    if name == "nupgrad":
        return NUPGrad(norm_eps=args.agg_norm_eps, reg_eps=args.agg_reg_eps)
    if name == "pnupgrad":
        return PNUPGrad(norm_eps=args.agg_norm_eps, reg_eps=args.agg_reg_eps)
    if name == "comfort":
        return COMFORT(mgda_norm_type=getattr(args, "comfort_mgda_norm_type", "none"),
                       mgda_stable=getattr(args, "comfort_mgda_stable", False), mgda_epsilon=args.mgda_epsilon,
                       mgda_max_iters=args.mgda_max_iters, mgda_min_eigenvalue_eps=getattr(args, "mgda_min_eigenvalue_eps", 1e-10),
                       beta_k=getattr(args, "comfort_beta_k", 1.0), beta_a=getattr(args, "comfort_beta_a", 1.0),
                       beta_l=getattr(args, "comfort_beta_l", 0.01), beta_u=getattr(args, "comfort_beta_u", 1.0))
    if name == "pcgrad":
        return PCGrad()
    if name == "imtlg":
        return IMTLG()
    if name == "cagrad":
        return CAGrad(c=1.0, norm_eps=args.agg_norm_eps)
    if name == "dualproj":
        return DualProj(norm_eps=args.agg_norm_eps, reg_eps=args.agg_reg_eps)
    if name == "sum":
        return "sum"
    if name in OUT_OF_SCOPE:
        raise NotImplementedError(f"Aggregator {args.aggregator} exists in the reference but is outside this build's scope "
                                  "(sum, upgrad, nupgrad, pnupgrad, comfort, mgda*, aligned_mtl*, mean, jd_sum, pcgrad, imtlg, dualproj, cagrad); see DESIGN.md")
    raise ValueError(f"Aggregator {args.aggregator} not supported")
