"""Oracle: Gramian, weightings and combine for the K-loss gradient aggregation (TEST
INFRASTRUCTURE, see oracle/__init__.py).

MGDA / Aligned-MTL follow the reference's in-tree code and are pinned by
tests/golden/weightings.npz (generated from that code).  UPGrad / Mean / Sum follow torchjd's
published algorithm (third-party, absent: parity unpinned beyond the docstring KAT
utils/torchmoo/nupgrad.py:58-62, ``UPGrad()([[-4,1,1],[6,1,1]]) = [0.2929, 1.9004, 1.9004]``).
"""
import itertools

import numpy as np
import torch


def gramian(J):
    """torchjd GramianWeightedAggregator: G = J J^T in the Jacobian's dtype (fp32)."""
    return J @ J.T


# ---- UPGrad -----------------------------------------------------------------------------
def _qp_lower_bounded(G, u):
    """argmin_w 1/2 w^T G w  s.t.  w >= u   (G symmetric positive definite, float64).

    Exact active-set enumeration: with v = w - u >= 0 the KKT system is
        v_F = -G_FF^{-1} (G u)_F,  v_A = 0,  (G (u + v))_A >= 0,  v_F >= 0
    for exactly one partition (F free, A active) because the problem is strictly convex.
    quadprog's Goldfarb-Idnani iteration (what torchjd calls through qpsolvers) converges to the
    same unique point.
    """
    K = len(u)
    c = G @ u
    best, best_viol = None, np.inf
    for r in range(K + 1):
        for F in itertools.combinations(range(K), r):
            F = list(F)
            v = np.zeros(K)
            if F:
                v[F] = np.linalg.solve(G[np.ix_(F, F)], -c[F])
            grad = G @ v + c
            A = [i for i in range(K) if i not in F]
            viol = max([0.0] + [-v[i] for i in F] + [-grad[i] for i in A])
            if viol < best_viol:
                best, best_viol = u + v, viol
            if viol <= 1e-12 * max(1.0, np.abs(c).max()):
                return u + v
    return best


def normalize_min_l2(G, eps):
    """utils/torchmoo/nupgrad.py:122-158 (identical copy in pnupgrad.py:137-166): scale gradient k by a_min / a_k, a_k the
    L2 norms sqrt(clamp(G_kk, eps)), a_min the smallest norm above eps (zero matrix when there is none).  float32 like the
    reference.  Pinned by tests/golden/agg_variants.npz."""
    G = np.asarray(G, dtype=np.float32)
    l2 = np.sqrt(np.maximum(np.diagonal(G), np.float32(eps))).astype(np.float32)
    mask = l2 > np.float32(eps)
    if not mask.any():
        return np.zeros_like(G)
    amin = l2[mask].min()
    sf = np.where(mask, amin / l2, np.float32(0.0)).astype(np.float32)
    return (G * (sf[:, None] * sf[None, :])).astype(np.float32)


def normalize_cosine(G, eps):
    """utils/torchmoo/pnupgrad.py:13-24: G_ij / (||g_i|| ||g_j||) with ||g_k|| = sqrt(clamp(G_kk, eps)).  float32."""
    G = np.asarray(G, dtype=np.float32)
    gn = np.sqrt(np.maximum(np.diagonal(G), np.float32(eps))).astype(np.float32)
    return (G / (gn[:, None] * gn[None, :])).astype(np.float32)


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


def upgrad_weights(G, norm_eps=1e-4, reg_eps=1e-4, pref=None, norm="trace"):
    """torchjd UPGrad weighting (constructed at main.py:1195): trace-normalise, regularise,
    project every row of U = diag(pref or 1/K) onto the dual cone, sum the rows.
    Arithmetic in float64 on the host like torchjd's numpy path; result cast to G's dtype.
    norm = "min_l2" is NUPGrad (nupgrad.py:115-120), "cosine" PNUPGrad's other branch (pnupgrad.py:127-134): the same
    projection on a differently normalised Gramian."""
    Gd = np.asarray(G.detach().cpu().numpy() if isinstance(G, torch.Tensor) else G, dtype=np.float64)
    K = Gd.shape[0]
    if norm == "trace":
        tr = np.trace(Gd)
        Gn = np.zeros_like(Gd) if tr < norm_eps else Gd / tr
    elif norm == "min_l2":
        Gn = normalize_min_l2(Gd, norm_eps).astype(np.float64)
    elif norm == "cosine":
        Gn = normalize_cosine(Gd, norm_eps).astype(np.float64)
    else:
        raise ValueError(norm)
    Gn = Gn + reg_eps * np.eye(K)
    u_diag = np.full(K, 1.0 / K) if pref is None else np.asarray(pref, dtype=np.float64)
    W = np.zeros((K, K))
    for i in range(K):
        u = np.zeros(K)
        u[i] = u_diag[i]
        W[i] = _qp_lower_bounded(Gn, u)
    return W.sum(axis=0)


# ---- torchjd DualProj / PCGrad / IMTL-G (main.py:1196-1222) -----------------------------------------------------------
# third-party torchjd @ main (requirements.txt:58), absent from the reference tree: restated from the published algorithms
# (DualProj: torchjd's `project_weights` applied once to the mean weights; PCGrad: Yu et al. 2020 on the Gramian with
# torch.randperm task orders; IMTL-G: Liu et al. 2021, pinv(G) d normalised to sum 1).  parity unpinned.
def dualproj_weights(G, norm_eps=1e-4, reg_eps=1e-4, pref=None):
    Gd = np.asarray(G.detach().cpu().numpy() if isinstance(G, torch.Tensor) else G, dtype=np.float64)
    K = Gd.shape[0]
    tr = np.trace(Gd)
    Gn = (np.zeros_like(Gd) if tr < norm_eps else Gd / tr) + reg_eps * np.eye(K)
    u = np.full(K, 1.0 / K) if pref is None else np.asarray(pref, dtype=np.float64)
    return _qp_lower_bounded(Gn, u)


def pcgrad_weights(G):
    """Draws K permutations from torch's global CPU generator, one per task, like torchjd's loop."""
    Gt = torch.as_tensor(np.asarray(G), dtype=torch.float32)
    K = Gt.shape[0]
    weights = torch.zeros(K)
    for i in range(K):
        permutation = torch.randperm(K)
        cur = torch.zeros(K)
        cur[i] = 1.0
        for j in permutation.tolist():
            if j == i:
                continue
            ip = Gt[j] @ cur
            if ip < 0.0:
                cur[j] -= ip / Gt[j, j]
        weights = weights + cur
    return weights.numpy()


def imtlg_weights(G):
    Gt = torch.as_tensor(np.asarray(G), dtype=torch.float32)
    d = torch.sqrt(torch.diagonal(Gt))
    v = torch.linalg.pinv(Gt) @ d
    vs = v.sum()
    return (torch.zeros_like(v) if vs.abs() < 1e-12 else v / vs).numpy()


def cagrad_weights(G, c=1.0, norm_eps=1e-4):
    """torchjd CAGrad (Liu et al. 2021; main.py:1216-1217 builds CAGrad(c=1.0, norm_eps)): with u = 1/K, g0 = sqrt(u'Gu),
    w* = argmin over the simplex of (Gu)'w + c g0 sqrt(w'Gw); weights = u + (c g0 / sqrt(w*'Gw*)) w*, the mean weights when
    c g0 <= norm_eps or sqrt(w*'Gw*) <= norm_eps.  torchjd hands the cone problem to cvxpy / CLARABEL; here the KKT system
    is solved in closed form on every support S (G_S w = (t / kappa)(lambda 1 - b_S), a quadratic in lambda) and the
    support with the least KKT violation wins -- checked against torchjd's documented example and scipy SLSQP.
    parity unpinned."""
    Gd = np.asarray(G.detach().cpu().numpy() if isinstance(G, torch.Tensor) else G, dtype=np.float64)
    K = Gd.shape[0]
    u = np.full(K, 1.0 / K)
    kappa = c * np.sqrt(max(u @ Gd @ u, 0.0))
    if kappa <= norm_eps:
        return u
    b = Gd @ u
    best, best_viol = None, np.inf
    jitter = 1e-14 * np.trace(Gd)
    for r in range(1, K + 1):
        for S in itertools.combinations(range(K), r):
            S = list(S)
            GS = Gd[np.ix_(S, S)] + jitter * np.eye(r)
            try:
                np.linalg.cholesky(GS)
            except np.linalg.LinAlgError:
                continue
            p, q = np.linalg.solve(GS, np.ones(r)), np.linalg.solve(GS, b[S])
            A, B, C = p.sum(), q.sum(), b[S] @ q
            disc = B * B - A * (C - kappa ** 2)
            if disc < 0 or A <= 0:
                continue
            lam = (B + np.sqrt(disc)) / A
            den = lam * A - B
            if den <= 0:
                continue
            w = np.zeros(K)
            w[S] = (lam * p - q) / den
            t = np.sqrt(max(w @ Gd @ w, 0.0))
            if t <= 0:
                continue
            grad = b + kappa * (Gd @ w) / t - lam
            viol = max([0.0] + [-w[i] for i in S] + [-grad[i] for i in range(K) if i not in S])
            if viol < best_viol:
                best, best_viol = w, viol
    if best is None:
        return u
    gw = np.sqrt(max(best @ Gd @ best, 0.0))
    if gw <= norm_eps:
        return u
    return u + (kappa / gw) * best


# ---- MGDA (utils/torchmoo/mgda.py:221-367) ---------------------------------------------
def mgda_weights(G, norm_type="none", losses=None, epsilon=1e-5, max_iters=250,
                 stable=False, min_eigenvalue_eps=1e-10, return_iters=False):
    G = np.asarray(G, dtype=np.float32)
    K = G.shape[0]
    if norm_type in ("l2", "loss+"):
        n = np.sqrt(np.maximum(np.diag(G), np.float32(1e-20))).astype(np.float32)
    if norm_type in ("loss", "loss+"):
        if losses is None:
            raise RuntimeError("Losses must be set before calling forward()")
        ls = np.maximum(np.asarray(losses, dtype=np.float32), np.float32(1e-20))
    if norm_type == "l2":
        G = G / np.outer(n, n)
    elif norm_type == "loss":
        G = G / np.outer(ls, ls)
    elif norm_type == "loss+":
        c = (ls * n).astype(np.float32)
        G = G / np.outer(c, c)
    G = G.astype(np.float32)
    if stable:  # mgda.py:287-317
        lam, V = np.linalg.eigh(G)
        lam = np.maximum(lam, np.float32(min_eigenvalue_eps))
        G = (V @ (lam[:, None] * V.T)).astype(np.float32)
    alpha = np.full(K, 1.0 / K, dtype=np.float32)
    it = 0
    for it in range(max_iters):
        Ga = G @ alpha
        t = int(np.argmin(Ga))
        a = np.float32(alpha @ G[:, t])
        b = np.float32(alpha @ Ga)
        c = G[t, t]
        if c <= a:
            gamma = np.float32(1.0)
        elif b <= a:
            gamma = np.float32(0.0)
        else:
            gamma = np.float32((b - a) / (b + c - 2 * a))
        alpha = ((1 - gamma) * alpha).astype(np.float32)
        alpha[t] += gamma
        if gamma < epsilon:
            break
    return (alpha, it + 1) if return_iters else alpha


# ---- Aligned-MTL (utils/torchmoo/aligned_mtl.py:97-133) ------------------------------------
def aligned_mtl_weights(G, scale_mode="min", pref=None):
    G = np.asarray(G, dtype=np.float32)
    K = G.shape[0]
    w0 = np.full(K, 1.0 / K, dtype=np.float32) if pref is None else np.asarray(pref, dtype=np.float32)
    lam, V = np.linalg.eigh(G, UPLO="U")
    tol = lam.max() * K * np.finfo(np.float32).eps
    rank = int((lam > tol).sum())
    if rank == 0:
        return w0
    order = np.argsort(-lam, kind="stable")
    lam, V = lam[order][:rank], V[:, order][:, :rank]
    if scale_mode == "min":
        scale = lam[-1]
    elif scale_mode == "median":
        scale = np.sort(lam)[(rank - 1) // 2]  # torch.median returns the lower middle
    elif scale_mode == "rmse":
        scale = lam.mean()
    else:
        raise ValueError(f"Invalid scale_mode={scale_mode!r}")
    B = np.sqrt(scale) * (V @ np.diag(1 / np.sqrt(lam)) @ V.T)
    return (B @ w0).astype(np.float32)


# ---- factory mirroring main.py:1191-1246 --------------------------------------------------------
def make_weighting(name, **kw):
    """Returns f(G, losses) -> weights (numpy)."""
    n = name.lower()
    if n == "upgrad":
        return lambda G, losses=None: upgrad_weights(G, kw.get("norm_eps", 1e-4), kw.get("reg_eps", 1e-4))
    if n in ("aligned_mtl", "aligned_mtl_min", "amtl", "amtl_min"):
        return lambda G, losses=None: aligned_mtl_weights(G, "min")
    if n == "aligned_mtl_median":
        return lambda G, losses=None: aligned_mtl_weights(G, "median")
    if n == "aligned_mtl_rmse":
        return lambda G, losses=None: aligned_mtl_weights(G, "rmse")
    mg = {"mgda": "none", "mgda_ln": "l2", "mgda_gn": "loss", "mgda_lgn": "loss+"}
    if n in mg:
        return lambda G, losses=None: mgda_weights(G, mg[n], losses, kw.get("epsilon", 1e-5), kw.get("max_iters", 250))
    if n == "nupgrad":
        return lambda G, losses=None: upgrad_weights(G, kw.get("norm_eps", 1e-4), kw.get("reg_eps", 1e-4), norm="min_l2")
    if n in ("pnupgrad_cosine", "pnupgrad_min_l2"):  # the two branches of PNUPGrad's coin flip (pnupgrad.py:129-132)
        return lambda G, losses=None: upgrad_weights(G, kw.get("norm_eps", 1e-4), kw.get("reg_eps", 1e-4), norm=n.split("_", 1)[1])
    if n == "comfort":  # (1 - beta) MGDA + beta UPGrad (comfort.py:146-157); UPGrad() with its default eps
        beta = beta_schedule(kw.get("epoch", 1), kw.get("total_epochs", 1), kw.get("beta_k", 1.0), kw.get("beta_a", 1.0),
                             kw.get("beta_l", 0.01), kw.get("beta_u", 1.0))
        nt = kw.get("mgda_norm_type", "none")
        return lambda G, losses=None: ((1.0 - beta) * np.asarray(mgda_weights(G, nt, losses, kw.get("epsilon", 1e-5), kw.get("max_iters", 250)),
                                                                  dtype=np.float64) + beta * upgrad_weights(G))
    if n == "dualproj":
        return lambda G, losses=None: dualproj_weights(G, kw.get("norm_eps", 1e-4), kw.get("reg_eps", 1e-4), kw.get("pref"))
    if n == "pcgrad":
        return lambda G, losses=None: pcgrad_weights(G)
    if n == "imtlg":
        return lambda G, losses=None: imtlg_weights(G)
    if n == "cagrad":
        return lambda G, losses=None: cagrad_weights(G, kw.get("c", 1.0), kw.get("norm_eps", 1e-4))
    if n == "mean":
        return lambda G, losses=None: np.full(len(G), 1.0 / len(G))
    if n == "jd_sum":
        return lambda G, losses=None: np.ones(len(G))
    raise ValueError(f"Aggregator {name} not supported")


def aggregate(J, weighting, losses=None):
    """aggregator(J) = weighting(J J^T) @ J; returns (g, w, G)."""
    G = gramian(J)
    w = torch.as_tensor(np.asarray(weighting(G.detach().cpu().numpy(), losses)), dtype=J.dtype)
    return w @ J, w, G
