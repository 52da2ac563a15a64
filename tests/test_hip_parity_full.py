"""Parity at the BASELINE.json shapes, element by element, against the CPU oracle
run on the same seeded inputs in the same process (the oracle itself is pinned by tests/test_oracle_golden.py):

  * the `--agg sum` gradient of every parameter (not its norm);
  * the AGGREGATED step each config names -- C2 upgrad, C3 aligned_mtl, C4 mgda_ln, C5 upgrad -- through
    autojac.mtl_backward: Gramian, weights and every parameter's gradient (batched pull-back, grouped wgrad with split-K,
    paired launches at the real tile counts);
  * the north-star target "ELBO within 1e-3 relative of the reference after 1 epoch": one CIFAR-10 epoch at bs 256 is 196
    steps; HIP eager and HIP hipGraph replay are stepped side by side with OracleTrainer.step on the same batches and the
    same CPU-drawn eps, every loss component compared at the end and the maximum drift along the way reported.

GPU only.  The UPGrad / mtl_backward arithmetic of the oracle restates third-party torchjd (absent): parity unpinned beyond
the docstring KAT and the unit-weights invariant."""
import os

import numpy as np
import pytest
import torch

from conftest import cfg_from_meta, load_golden

pytestmark = pytest.mark.gpu

#: batch sizes: the per-GPU batch of the config (C4 = 64 over 8 GPUs, C5 = 32 per GPU); MOVAE_TEST_SMALL=1 falls back to the
#: reduced batches of tests/golden/full_configs.npz
FULL_B = {"C1": 128, "C2": 256, "C3": 128, "C4": 8, "C5": 32}
AGG = {"C2": "upgrad", "C3": "aligned_mtl", "C4": "mgda_ln", "C5": "upgrad"}


class Args:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _case(tag):
    fx = load_golden("full_configs")
    m = {}
    for s in fx[f"{tag}.meta"]:
        k, v = str(s).split("=", 1)
        m[k] = v
    m["objective"] = "mse"
    c = cfg_from_meta(m)
    if os.environ.get("MOVAE_TEST_SMALL") != "1":
        c["batch_size"] = FULL_B[tag]
    return c, int(m["seed"])


def _build_pair(tag, agg, device):
    """(HIP net, OracleTrainer, x, eps) with identical parameters (both replay the reference's init sequence from the seed)."""
    import movae_amd  # noqa: F401
    from movae_amd.models import get_network
    from movae_amd.models.betatc_vae import BetaTCVAE
    from oracle import nets
    from oracle.step import OracleTrainer

    c, seed = _case(tag)
    B, size = c["batch_size"], c["input_size"]
    kw = {k: v for k, v in c.items() if k in ("latent_dim", "hidden_dims", "embedding_dim", "num_embeddings", "num_residual_layers",
                                              "anneal_steps")}
    args = Args(arch=c["arch"], batch_size=B, dataset_size=c["dataset_size"], recons_objective="mse", recons_activation=None,
                loss_weights=None, **kw)
    torch.manual_seed(seed)
    BetaTCVAE.num_iter = 0
    net = get_network(size, num_channels=3, args=args, device=device).to(device).train()
    tr = OracleTrainer(nets.make_cfg(**c), seed=seed, agg=agg)
    for n, p in net.named_parameters():  # same init on both sides, bit for bit
        assert torch.equal(p.detach().cpu(), tr.params[n].detach()), f"init differs: {n}"
    x = torch.rand(B, 3, size, size, generator=torch.Generator().manual_seed(seed + 1))
    eps = torch.randn(B, c["latent_dim"], generator=torch.Generator().manual_seed(seed + 2)) if "latent_dim" in c else None
    if eps is not None:
        net.eps_override = eps.to(device)
    return net, tr, x, eps, c


def _fp64_twin(tr):
    """The same oracle in float64 (same initial values): the yardstick that tells rounding noise from error.  A gradient that
    passes through ten training-mode BatchNorms is a difference of nearly equal sums; fp32 implementations that add in
    different orders legitimately differ by 1e-4..1e-3 of the largest entry -- the fp32 oracle itself does, against this twin."""
    from collections import OrderedDict

    from oracle.step import OracleTrainer

    t64 = OracleTrainer(tr.cfg, seed=0, agg=tr.agg)
    with torch.no_grad():
        for k, v in tr.sd.items():
            t64.sd[k] = v.detach().double().requires_grad_(v.requires_grad) if v.is_floating_point() else v.detach().clone()
    t64.params = OrderedDict((n, t64.sd[n]) for n in tr.params)
    return t64


class KinkRecorder:
    """LeakyReLU' jumps at 0.  Of the 1.6e7 pre-activations of a C2 step a handful lie within fp32 rounding distance (1e-7)
    of it, and two correct fp32 implementations may put such an element on different sides.  ONE such flip moves the gradient
    of every layer upstream of it by ~1e-3 relative (the BatchNorm backward spreads it over the channel; measured on the fp64
    oracle by flipping the smallest pre-activation of encoder.1: encoder.1.1.bias 1.1e-3, encoder.0.0.weight 7.8e-4, nothing
    downstream) -- as large as any error this test is meant to find.  So the side every pre-activation fell on is RECORDED on
    the HIP path (in call order: fused BatchNorm + activation, conv / linear epilogue activations, stand-alone activations) and
    the float64 oracle is evaluated with those sides (`masked`): it then differentiates exactly the piecewise-linear function
    the HIP step evaluated, and the comparison is held to rounding-level tolerances again."""

    def __init__(self, monkeypatch):
        from movae_amd import ops

        self.masks = []
        o_lazy, o_bna, o_c, o_ct, o_lin, o_act = (ops.batch_norm_lazy, ops.batch_norm_act, ops.conv2d, ops.conv_transpose2d, ops.linear,
                                                  ops.activation)

        def rec(out):
            self.masks.append((out.detach() > 0).cpu())
            return out

        def lazy(y, gamma, beta, rm, rv, nbt, eps, momentum, act, slope, fusion):
            out = o_lazy(y, gamma, beta, rm, rv, nbt, eps, momentum, act, slope, fusion)
            if act == "lrelu" and isinstance(out, ops.LazyBN):  # (a tensor came through the patched batch_norm_act: recorded there)
                rec(ops.scale_shift_act(out.y.detach(), out.scale, out.shift, out.slope))  # the kernels' own fmaf(y, scale, shift)
            return out

        def bna(*a, **k):
            act = k.get("act", a[8] if len(a) > 8 else None)
            out = o_bna(*a, **k)
            return rec(out) if act == "lrelu" else out

        def wrap(fn, pos):
            def f(*a, **k):
                act = k.get("act", a[pos] if len(a) > pos else None)
                out = fn(*a, **k)
                return rec(out) if act == "lrelu" else out
            return f

        monkeypatch.setattr(ops, "batch_norm_lazy", lazy)
        monkeypatch.setattr(ops, "batch_norm_act", bna)
        monkeypatch.setattr(ops, "conv2d", wrap(o_c, 5))
        monkeypatch.setattr(ops, "conv_transpose2d", wrap(o_ct, 6))
        monkeypatch.setattr(ops, "linear", wrap(o_lin, 3))
        monkeypatch.setattr(ops, "activation", wrap(o_act, 1))

    def nchw(self):
        return [m.permute(0, 3, 1, 2) if m.dim() == 4 else m for m in self.masks]


class masked:
    """Context: torch.nn.functional.leaky_relu takes the side of the kink from `masks` (call order) instead of from its input;
    with masks=None it records the sides it takes (the fp32 oracle's own)."""

    def __init__(self, masks=None):
        self.masks, self.seen, self.k = masks, [], 0

    def __enter__(self):
        import torch.nn.functional as F

        self.orig = F.leaky_relu

        def lrelu(z, negative_slope=0.01, inplace=False):
            if self.masks is None:
                self.seen.append(z.detach() > 0)
                return self.orig(z, negative_slope)
            assert self.k < len(self.masks), f"the oracle applies more LeakyReLUs than the HIP path recorded ({len(self.masks)})"
            m = self.masks[self.k]
            self.k += 1
            assert m.shape == z.shape, f"LeakyReLU #{self.k - 1}: HIP recorded {tuple(m.shape)}, oracle has {tuple(z.shape)}"
            return torch.where(m, z, z * negative_slope)

        F.leaky_relu = lrelu
        return self

    def __exit__(self, *exc):
        import torch.nn.functional as F

        F.leaky_relu = self.orig
        if self.masks is not None and exc[0] is None:
            assert self.k == len(self.masks), f"the oracle applied {self.k} LeakyReLUs, the HIP path recorded {len(self.masks)}"


#: Two correct fp32 evaluations put a handful of the ~1.6e7 pre-activations of a step on different sides of the LeakyReLU kink
#: (measured: C1 3, C2 5, C5 6 -- values within fp32 rounding distance of zero).  The fp64 yardstick is evaluated on the HIP step's
#: sides, so the COUNT has to be bounded too: a kernel bug that flipped thousands would otherwise be absorbed by the mask.
MAX_KINK_FLIPS = 32


def _kink_report(hip_masks, ora_masks, what):
    n = sum(int((a != b).sum()) for a, b in zip(hip_masks, ora_masks))
    tot = sum(a.numel() for a in hip_masks)
    print(f"[{what}] {n} of {tot} pre-activations fell on different sides of the LeakyReLU kink in the HIP step and the fp32 oracle")
    assert n <= max(MAX_KINK_FLIPS, int(2e-6 * tot)), (
        f"{what}: {n} of {tot} pre-activations on different sides of the LeakyReLU kink (fp32 rounding explains <= {MAX_KINK_FLIPS})")
    return n


def _cmp_vs_fp64(net, g32, g64, what, factor=8.0, floor=2e-4, g64_ora=None):
    """Per parameter: rel-L2 error of the HIP gradient against the fp64 oracle must stay within `factor` x the fp32 oracle's own
    error against it (or `floor`, whichever is larger), and the same for the worst single entry (as a fraction of the largest
    entry).  Returns the table's worst rows for the log."""
    rows, bad = [], []
    for n, p in net.named_parameters():
        ref = g64[n].detach().numpy()
        ref_o = (g64_ora if g64_ora is not None else g64)[n].detach().numpy()  # the fp64 oracle on the fp32 oracle's kink sides
        o32 = g32[n].detach().double().numpy()
        got = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().cpu().double().numpy()
        scale, nrm = np.abs(ref).max(), np.linalg.norm(ref)
        if scale < 1e-7:  # analytically zero (conv bias in front of BatchNorm)
            if np.abs(got).max() > 1e-6:
                bad.append((n, "expected ~0", float(np.abs(got).max())))
            continue
        e_hip, e_ora = np.linalg.norm(got - ref) / nrm, np.linalg.norm(o32 - ref_o) / nrm
        m_hip, m_ora = np.abs(got - ref).max() / scale, np.abs(o32 - ref_o).max() / scale
        rows.append((n, e_hip, e_ora, m_hip, m_ora))
        if e_hip > max(factor * e_ora, floor) or m_hip > max(factor * m_ora, 4 * floor):
            bad.append((n, f"relL2 hip {e_hip:.2e} vs fp32-oracle {e_ora:.2e}", f"max-entry hip {m_hip:.2e} vs {m_ora:.2e}"))
    rows.sort(key=lambda r: -r[1])
    print(f"[{what}] worst parameters (relL2 HIP / fp32 oracle, max-entry HIP / fp32 oracle, all against the fp64 oracle):")
    for r in rows[:5]:
        print(f"    {r[0]:34s} {r[1]:.2e} / {r[2]:.2e}   {r[3]:.2e} / {r[4]:.2e}")
    assert not bad, f"{what}: {len(bad)} parameters off: {bad[:6]}"
    return rows[0][1] if rows else 0.0


def _cmp_grads(net, ograds, rtol, atol_rel, what):
    """Element-wise: |got - want| <= rtol * |want| + atol_rel * max|want| per parameter; returns the worst global rel-L2."""
    worst, bad = 0.0, []
    for n, p in net.named_parameters():
        want = ograds[n].detach().double().numpy()
        got = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().cpu().double().numpy()
        scale = np.abs(want).max()
        err = np.abs(got - want)
        lim = rtol * np.abs(want) + atol_rel * max(scale, 1e-30)
        if scale < 1e-7:  # an analytically zero gradient (conv bias in front of BatchNorm): ~1e-9 rounding noise on the oracle side
            if np.abs(got).max() > 1e-6:
                bad.append((n, "expected ~0", float(np.abs(got).max())))
            continue
        if (err > lim).any():
            i = int(np.argmax(err - lim))
            bad.append((n, float(got.reshape(-1)[i]), float(want.reshape(-1)[i]), float(scale)))
        rel = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30)
        worst = max(worst, rel)
    assert not bad, f"{what}: {len(bad)} parameters differ element-wise, first: {bad[:4]}"
    return worst


# BN-free (C5) and the small VAEs (C1/C2) hold fp32 tolerance element-wise; the VQ configs route some rows through the
# codebook (argmin near-ties can move single rows between codes, which the loss tolerance above already bounds)
# (rtol on the element, atol as a fraction of the parameter's largest gradient entry: the fp32 noise floor of a sum over
# 10^4..10^6 products sits near 1e-4 of the largest entry, so entries far below that carry no significant digits)
SUM_TOL = {"C1": (2e-3, 2e-4), "C2": (2e-3, 2e-4), "C3": (5e-3, 5e-4), "C4": (5e-3, 5e-4), "C5": (2e-3, 2e-4)}


@pytest.mark.parametrize("tag", ["C1", "C2", "C3", "C4", "C5"])
def test_full_size_sum_gradients_elementwise(tag, gpu_device, monkeypatch):
    net, tr, x, eps, c = _build_pair(tag, "sum", gpu_device)
    t64 = _fp64_twin(tr) if tag not in ("C3", "C4") else None  # (before tr runs: BetaTC's annealing counter lives in the cfg)
    t64o = _fp64_twin(tr) if t64 is not None else None
    with masked() as own:  # the fp32 oracle, recording its own kink sides
        _, old, ograds, _ = tr.grads(x, eps)
    kinks = KinkRecorder(monkeypatch)
    xg = x.to(gpu_device)
    out = net(xg)
    ld = net.loss_function(xg, args=out)
    assert list(ld.keys()) == list(old.keys())
    for k, v in ld.items():
        np.testing.assert_allclose(v.item(), float(old[k].detach()), rtol=5e-4, atol=1e-5 + 2e-6 * abs(float(old["total_loss"].detach())), err_msg=k)
    ld["total_loss"].backward()
    if tag in ("C3", "C4"):  # VQ: a float64 run quantises near-tied rows differently, so the fp32 oracle is the reference
        rtol, atol_rel = SUM_TOL[tag]
        worst = _cmp_grads(net, ograds, rtol, atol_rel, f"{tag} sum")
        assert worst < 2e-3, f"{tag}: global rel-L2 {worst:.2e}"
    else:
        x64, e64 = x.double(), eps.double() if eps is not None else None
        _kink_report(kinks.nchw(), own.seen, f"{tag} sum")
        with masked(kinks.nchw()):      # float64 on the sides the HIP step took ...
            _, _, g64, _ = t64.grads(x64, e64)
        with masked(own.seen):          # ... and on the sides the fp32 oracle took (its own error: the yardstick)
            _, _, g64o, _ = t64o.grads(x64, e64)
        worst = _cmp_vs_fp64(net, ograds, g64, f"{tag} sum B={c['batch_size']}", g64_ora=g64o)
    print(f"[{tag} sum B={c['batch_size']}] worst per-parameter rel-L2 = {worst:.2e}")


@pytest.mark.parametrize("tag", ["C2", "C3", "C4", "C5"])
def test_full_size_aggregated_step_matches_oracle(tag, gpu_device, monkeypatch):
    import movae_amd  # noqa: F401
    from movae_amd import aggregation, autojac

    agg = AGG[tag]
    net, tr, x, eps, c = _build_pair(tag, agg, gpu_device)
    vq = tag in ("C3", "C4")
    t64 = _fp64_twin(tr) if not vq else None
    t64o = _fp64_twin(tr) if not vq else None
    with masked() as own:
        _, old, ograds, oinfo = tr.grads(x, eps)
    kinks = KinkRecorder(monkeypatch)
    a = Args(aggregator=agg, agg_norm_eps=1e-4, agg_reg_eps=1e-4, mgda_epsilon=1e-5, mgda_max_iters=250, pref_weights=None)
    A = aggregation.make_aggregator(a)
    seen = {}
    # G and w are taken from the device inside the hook (fp64 Gramian of the Jacobian the HIP step really built)
    A.weighting.register_forward_hook(lambda mod, inp, o: seen.update(G=(inp[0].double() @ inp[0].double().T).cpu(), w=o.detach().cpu(),
                                                                       m=inp[0].shape[1]))
    xg = x.to(gpu_device)
    out = net(xg)
    ld = net.loss_function(xg, args=out)
    comp = [v for k, v in ld.items() if k != "total_loss"]
    if isinstance(A, aggregation.MGDA):
        A.set_losses(torch.stack(comp))
    net.zero_grad(set_to_none=True)
    autojac.mtl_backward(losses=comp, features=[out[f] for f in net.features], aggregator=A, retain_graph=True)
    torch.cuda.synchronize()
    i64 = g64 = g64o = None
    if not vq:  # float64 oracle on the kink sides of the HIP step / of the fp32 oracle (see KinkRecorder)
        x64, e64 = x.double(), eps.double() if eps is not None else None
        _kink_report(kinks.nchw(), own.seen, f"{tag} {agg}")
        with masked(kinks.nchw()):
            _, _, g64, i64 = t64.grads(x64, e64)
        with masked(own.seen):
            _, _, g64o, i64o = t64o.grads(x64, e64)
    Go = (i64["G"] if i64 is not None else oinfo["G"]).double().numpy()
    assert seen["m"] == oinfo["J"].shape[1], "shared-parameter Jacobian width"
    # an off-diagonal entry is an inner product of two long vectors that may nearly cancel: its noise scales with the two norms
    dg = np.sqrt(np.abs(np.diag(Go)))
    gerr = np.abs(seen["G"].numpy() - Go) / np.maximum(np.outer(dg, dg), 1e-30)
    glim = 2e-3 if vq else 5e-4
    if not vq:  # the fp32 oracle's own Gramian error against float64 is the yardstick where a row is ill-conditioned in fp32 (C5's
        # total-correlation row: a log-sum-exp over the batch's pairwise densities; the fp32 oracle is 2e-3 off there)
        Goo = i64o["G"].double().numpy()
        gora = np.abs(oinfo["G"].double().numpy() - Goo) / np.maximum(np.outer(dg, dg), 1e-30)
        glim = np.maximum(glim, 4.0 * gora)
        print(f"[{tag} {agg}] Gramian error / sqrt(G_ii G_jj): HIP worst {gerr.max():.2e}, fp32 oracle worst {gora.max():.2e}")
    assert (gerr <= glim).all(), f"Gramian: worst |dG_ij| / sqrt(G_ii G_jj) = {gerr.max():.2e}\n{seen['G'].numpy()}\n{Go}"
    w_o = np.asarray(oinfo["w"], dtype=np.float64)
    # Aligned-MTL / MGDA weights are ill-conditioned functions of G (eigen-decomposition, a vertex search): looser
    cond = agg.startswith(("aligned", "mgda"))
    np.testing.assert_allclose(seen["w"].double().numpy(), w_o, rtol=2e-2 if cond else 1e-3, atol=1e-4 * max(1.0, np.abs(w_o).max()),
                               err_msg="weights")
    if vq:
        rtol, atol_rel = (3e-2, 1e-3) if cond else (5e-3, 5e-4)
        worst = _cmp_grads(net, ograds, rtol, atol_rel, f"{tag} {agg}")
    else:  # the aggregated gradient against the fp64 oracle, with the fp32 oracle's own error as the yardstick
        np.testing.assert_allclose(seen["w"].double().numpy(), np.asarray(i64["w"], dtype=np.float64), rtol=2e-3, atol=1e-4,
                                   err_msg="weights vs fp64 oracle")
        worst = _cmp_vs_fp64(net, ograds, g64, f"{tag} {agg} B={c['batch_size']}", g64_ora=g64o)
    print(f"[{tag} {agg} B={c['batch_size']}] w = {seen['w'].tolist()}, worst per-parameter rel-L2 = {worst:.2e}")


EPOCH_STEPS = int(os.environ.get("MOVAE_TEST_EPOCH_STEPS", "196"))  # ceil(50000 / 256): one CIFAR-10 epoch at bs 256

#: the oracle trajectories (fp32 and fp64) are the same for both launch modes: computed once per process
_ORACLE_TRAJ = {}


def _fp64_stepper(tr):
    """_fp64_twin with an optimizer bound to the float64 parameters (the twin's constructor made one for its own fp32 init)."""
    t64 = _fp64_twin(tr)
    t64.opt = torch.optim.Adam(list(t64.params.values()), lr=1e-3)
    return t64


def _oracle_trajectories(tr, xs, eps_list, steps):
    """(keys, fp32 oracle history [steps, K+1], fp64 oracle history): OracleTrainer.step on the same batches and noise."""
    key = (steps, len(xs))
    if key not in _ORACLE_TRAJ:
        t64 = _fp64_stepper(tr)
        keys, h32, h64 = None, [], []
        for i in range(steps):
            ol = tr.step(xs[i % len(xs)], eps_list[i])
            keys = keys or list(ol.keys())
            h32.append([ol[k] for k in keys])
            o64 = t64.step(xs[i % len(xs)].double(), eps_list[i].double())
            assert list(o64.keys()) == keys
            h64.append([o64[k] for k in keys])
        _ORACLE_TRAJ[key] = (keys, np.asarray(h32, dtype=np.float64), np.asarray(h64, dtype=np.float64))
    return _ORACLE_TRAJ[key]


def trajectory_limits(ora32, ora64, floor=1e-3, factor=4.0):
    """Per component: the relative tolerance a fp32 implementation is held to against the fp32 oracle -- 1e-3 (the north-star
    target) or `factor` x the fp32 oracle's own distance from the float64 oracle, whichever is larger.  A fp32 trajectory is
    chaotic in its small components (the weighted KL term of a near-collapsed posterior is ~4e-4 of the ELBO): two correct
    fp32 runs that add in different orders drift apart in it by more than 1e-3 of ITS value while the ELBO stays within 1e-3,
    and the fp32-vs-fp64 distance of the oracle itself measures exactly that."""
    own = np.abs(ora32 - ora64) / np.maximum(np.abs(ora64), 1e-12)
    return np.maximum(floor, factor * own), own


@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_c2_one_epoch_elbo_trajectory_matches_oracle(mode, gpu_device):
    """main.py:154-229 for one epoch at C2 (vae, upgrad, bs 256, Adam 1e-3): the loss dict of the LAST step and the epoch
    averages must agree with the oracle's, component by component (utils/objectives.py:141-144 for the KL term), within
    max(1e-3, 4 x the fp32 oracle's own distance from a float64 oracle stepped alongside) relative."""
    import movae_amd  # noqa: F401
    from movae_amd import aggregation
    from movae_amd.train import GraphedTrainStep, make_optimizer, train_step

    net, tr, _, _, c = _build_pair("C2", "upgrad", gpu_device)
    B, size, D = c["batch_size"], c["input_size"], c["latent_dim"]
    a = Args(aggregator="upgrad", agg_norm_eps=1e-4, agg_reg_eps=1e-4, mgda_epsilon=1e-5, mgda_max_iters=250, pref_weights=None,
             optimizer="adam", lr=1e-3, wd=0, momentum=0.9, max_grad_norm=None)
    opt = make_optimizer(net, a, capturable=(mode == "graph"))
    A = aggregation.make_aggregator(a)
    gx, ge = torch.Generator().manual_seed(1234), torch.Generator().manual_seed(4321)
    pool = 24  # distinct batches, cycled (the epoch's order is the same on both sides)
    xs = [torch.rand(B, 3, size, size, generator=gx) for _ in range(pool)]
    eps_list = [torch.randn(B, D, generator=ge) for _ in range(EPOCH_STEPS)]
    static_eps = torch.zeros(B, D, device=gpu_device)
    net.eps_override = static_eps  # the graph reads the noise from this address; refreshed before every step
    gs = None
    if mode == "graph":
        gs = GraphedTrainStep(net, opt, A, a, xs[0].to(gpu_device), preserve_state=True)  # warm-up steps are rewound
    xs_dev = [t.to(gpu_device) for t in xs]
    keys, hip_hist = None, []
    for i in range(EPOCH_STEPS):
        static_eps.copy_(eps_list[i])
        if gs is not None:
            ld, _ = gs.step(xs_dev[i % pool])
        else:
            ld, _ = train_step(net, xs_dev[i % pool], opt, A, a)
        keys = keys or list(ld.keys())
        hip_hist.append(torch.stack([ld[k].detach().reshape(()) for k in keys]))
    okeys, ora, ora64 = _oracle_trajectories(tr, xs, eps_list, EPOCH_STEPS)
    assert okeys == keys
    hip = torch.stack(hip_hist).double().cpu().numpy()
    assert np.isfinite(hip).all()
    rel = np.abs(hip - ora) / np.maximum(np.abs(ora), 1e-12)
    drift = {k: float(rel[:, j].max()) for j, k in enumerate(keys)}
    lim_last, own_last = trajectory_limits(ora[-1], ora64[-1])
    lim_mean, own_mean = trajectory_limits(ora.mean(0), ora64.mean(0))
    rel_last = np.abs(hip[-1] - ora[-1]) / np.maximum(np.abs(ora[-1]), 1e-12)
    rel_mean = np.abs(hip.mean(0) - ora.mean(0)) / np.maximum(np.abs(ora.mean(0)), 1e-12)
    print(f"[C2 {mode}] {EPOCH_STEPS} steps; three-way table (HIP fp32 | oracle fp32 | oracle fp64), relative distances to the fp32 oracle:")
    for j, k in enumerate(keys):
        print(f"    {k:22s} last {hip[-1, j]:.8f} | {ora[-1, j]:.8f} | {ora64[-1, j]:.8f}   HIP {rel_last[j]:.2e}  fp64 {own_last[j]:.2e}  limit {lim_last[j]:.2e}"
              f"   epoch mean HIP {rel_mean[j]:.2e}  fp64 {own_mean[j]:.2e}  limit {lim_mean[j]:.2e}   max drift {drift[k]:.2e}")
    it = keys.index("total_loss")
    # the north-star target itself: after one epoch the ELBO within 1e-3 relative -- last step and the epoch mean (what main.py logs)
    np.testing.assert_allclose(hip[-1, it], ora[-1, it], rtol=1e-3, err_msg="last-step ELBO")
    np.testing.assert_allclose(hip[:, it].mean(), ora[:, it].mean(), rtol=1e-3, err_msg="epoch-average ELBO")
    # every component, relative to ITSELF, against the fp64 yardstick
    assert (rel_last <= lim_last).all(), f"last-step components {keys}: rel {rel_last} > limits {lim_last} (fp32 oracle vs fp64: {own_last})"
    assert (rel_mean <= lim_mean).all(), f"epoch-average components {keys}: rel {rel_mean} > limits {lim_mean} (fp32 oracle vs fp64: {own_mean})"
    assert drift["total_loss"] < 5e-3 and drift["reconstruction_loss"] < 5e-3, drift  # and never far apart on the way
    assert ora[-1][it] < 0.5 * ora[0][it], "the epoch must actually train"
