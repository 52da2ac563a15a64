"""Gradient aggregators with torchjd's call protocol (what main.py:1191-1250 constructs and hooks):

    aggregator(J: Tensor[K, m]) -> Tensor[m]
    aggregator.weighting        -> nn.Module, forward(J) -> w[K]   (forward hooks see (J,), w)
    MGDA.set_losses(Tensor[K])

Everything numeric runs in libmovae_hip.so on the current stream: Gramian (one HBM pass over J),
the K x K solve in a single-wave fp64 kernel, combine (second pass).  No host synchronisation --
the reference's UPGrad round trip to numpy/quadprog and MGDA's per-iteration `.item()` are gone.
"""
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
        return self.gramian_weighting(compute_gramian(J))


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
    def __init__(self, pref_vector=None, norm_eps=1e-4, reg_eps=1e-4):
        super().__init__()
        self.pref_vector, self.norm_eps, self.reg_eps = pref_vector, norm_eps, reg_eps

    def forward(self, G):
        k = G.shape[0]
        w = torch.empty(k, dtype=torch.float32, device=G.device)
        pref = _pref_tensor(self.pref_vector, G.device)
        L.call("movae_weights_upgrad", G.data_ptr(), k, float(self.norm_eps), float(self.reg_eps), L.ptr(pref),
                                              w.data_ptr(), _st(G))
        return w


class UPGrad(GramianWeightedAggregator):
    """torchjd.aggregation.UPGrad(pref_vector, norm_eps, reg_eps) as constructed at main.py:1195."""

    def __init__(self, pref_vector=None, norm_eps=0.0001, reg_eps=0.0001, solver="quadprog"):
        super().__init__(UPGradWeighting(pref_vector, norm_eps, reg_eps))
        self._pref_vector, self._norm_eps, self._reg_eps = pref_vector, norm_eps, reg_eps

    def __repr__(self):
        return f"UPGrad(pref_vector={self._pref_vector!r}, norm_eps={self._norm_eps}, reg_eps={self._reg_eps})"


# ---- MGDA (utils/torchmoo/mgda.py) -------------------------------------------------------------------
class MGDAWeighting(Weighting):
    def __init__(self, norm_type="none", epsilon=1e-5, max_iters=250, stable=False, min_eigenvalue_eps=1e-10):
        super().__init__()
        if norm_type not in ("none", "l2", "loss", "loss+"):
            raise ValueError("Parameter `norm_type` should be 'none', 'l2', 'loss', or 'loss+'. Found "
                             f"`norm_type = {norm_type!r}`.")
        if stable:
            raise NotImplementedError("StableMGDA (eigen regularisation) is only reached through COMFORT, which is out of scope")
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
        L.call("movae_weights_mgda", G.data_ptr(), k, L.MGDA_NORM[self.norm_type], L.ptr(losses), float(self.epsilon),
                                            int(self.max_iters), w.data_ptr(), self._info.data_ptr(), _st(G))
        return w


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


OUT_OF_SCOPE = ("pcgrad", "imtlg", "cagrad", "nashmtl", "dualproj", "nupgrad", "pnupgrad", "comfort")


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
    if name == "sum":
        return "sum"
    if name in OUT_OF_SCOPE:
        raise NotImplementedError(f"Aggregator {args.aggregator} exists in the reference but is outside this build's scope "
                                  "(sum, upgrad, mgda*, aligned_mtl*, mean, jd_sum); see DESIGN.md")
    raise ValueError(f"Aggregator {args.aggregator} not supported")
