"""BatchNorm fused into the neighbouring convolutions (include/movae.h movae_fuse_t): the producer conv's
epilogue / split-K reduce emits the statistics, movae_bn_finalize folds them, the consumer conv (forward and weight gradient)
applies the normalisation + activation while it loads.  Every kernel variant that takes part is compared with the plain
PyTorch fp32 reference of the same chain (conv -> batch_norm(train) -> leaky_relu -> conv) on the CPU.  GPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(got, want, what, rtol=2e-4, atol=2e-5):
    got = got.detach().cpu().float().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = want.detach().cpu().float().numpy() if isinstance(want, torch.Tensor) else np.asarray(want)
    tol = atol * max(1.0, float(np.abs(want).max())) + rtol * np.abs(want)
    bad = np.argwhere(np.abs(got - want) > tol)
    where = ""
    if len(bad):  # which slices of each axis hold the mismatches: a tile-shaped pattern names the kernel and the block
        where = f" | {len(bad)} mismatches; per axis: " + "; ".join(
            f"ax{a}: {np.unique(bad[:, a])[:12].tolist()}{'...' if len(np.unique(bad[:, a])) > 12 else ''}" for a in range(bad.shape[1]))
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol * max(1.0, float(np.abs(want).max())), err_msg=what + where)


# (name, producer, consumer, batch, size, cin, cmid, cout): producer / consumer are ("conv" | "convT", k, stride, pad, out_pad)
CHAINS = [
    # encoder-like: igemm2_fwd producer (no split-K) -> igemm2_fwd consumer
    ("enc_64", ("conv", 3, 2, 1, 0), ("conv", 3, 2, 1, 0), 64, 16, 32, 64, 128),
    # deep encoder: split-K producer (statistics in the reduce) and a 1x1-output consumer (tap window)
    ("enc_deep", ("conv", 3, 2, 1, 0), ("conv", 3, 2, 1, 0), 32, 4, 128, 256, 512),
    # first layer: thin 3-channel producer (LDS-tile statistics) -> igemm consumer
    ("enc_first", ("conv", 3, 2, 1, 0), ("conv", 3, 2, 1, 0), 16, 32, 3, 32, 64),
    # decoder-like: transposed convs (BWD gather form, output-parity classes), unsplit and split
    ("dec_64", ("convT", 3, 2, 1, 1), ("convT", 3, 2, 1, 1), 32, 8, 64, 32, 32),
    ("dec_deep", ("convT", 3, 2, 1, 1), ("convT", 3, 2, 1, 1), 16, 2, 256, 128, 64),
    # last layer: igemm producer -> 3-channel thin consumer (transform while staging the tile; sweep wgrad)
    ("dec_last", ("convT", 3, 2, 1, 1), ("conv", 3, 1, 1, 0), 8, 16, 32, 32, 3),
    # odd sizes: ragged row blocks, channel counts that are multiples of 4 only
    ("ragged", ("conv", 3, 1, 1, 0), ("conv", 3, 2, 1, 0), 5, 7, 12, 20, 36),
]


def _mk(kind, cin, cout, k, gen):
    shape = (cout, cin, k, k) if kind == "conv" else (cin, cout, k, k)
    w = (torch.randn(shape, generator=gen) * (1.0 / (cin * k * k)) ** 0.5)
    return w, torch.randn(cout, generator=gen) * 0.1


def _ref_conv(kind, x, w, b, k, s, p, op):
    return F.conv2d(x, w, b, stride=s, padding=p) if kind == "conv" else F.conv_transpose2d(x, w, b, stride=s, padding=p, output_padding=op)


@pytest.mark.parametrize("case", CHAINS, ids=[c[0] for c in CHAINS])
@pytest.mark.parametrize("fused", [True, False])
def test_conv_bn_act_conv_chain(case, fused, gpu_device, monkeypatch):
    import movae_amd  # noqa: F401
    from movae_amd import nn as mnn, ops

    name, (k1, ks1, s1, p1, op1), (k2, ks2, s2, p2, op2), B, size, cin, cmid, cout = case
    monkeypatch.setattr(mnn, "FUSE_BN", fused)
    import zlib

    seed = zlib.crc32(name.encode()) % 1000  # (not hash(): that is salted per process)
    for attempt in range(50):
        g = torch.Generator().manual_seed(seed + attempt)
        x = torch.randn(B, cin, size, size, generator=g)
        w1, b1 = _mk(k1, cin, cmid, ks1, g)
        w2, b2 = _mk(k2, cmid, cout, ks2, g)
        gamma, beta = torch.rand(cmid, generator=g) + 0.5, torch.randn(cmid, generator=g) * 0.2
        # ---- fp32 reference on the CPU ----
        xr = x.clone().requires_grad_(True)
        pr = [t.clone().requires_grad_(True) for t in (w1, b1, gamma, beta, w2, b2)]
        rm, rv = torch.zeros(cmid), torch.ones(cmid)
        y1 = _ref_conv(k1, xr, pr[0], pr[1], ks1, s1, p1, op1)
        z = F.batch_norm(y1, rm, rv, pr[2], pr[3], training=True, momentum=0.1, eps=1e-5)
        # LeakyReLU' is discontinuous at 0: a pre-activation within rounding distance of zero may legitimately land on either
        # side in two fp32 implementations, and ONE flipped sign moves that element's gradient by 99 % and, through the batch
        # means of the BatchNorm backward, its whole channel a little.  Such draws (about one in forty) are not a parity
        # question: draw again.
        if float(z.detach().abs().min()) > 1e-6:
            break
    h = F.leaky_relu(z, 0.01)
    y2 = _ref_conv(k2, h, pr[4], pr[5], ks2, s2, p2, op2)
    cot = torch.randn(y2.shape, generator=g)
    (y2 * cot).sum().backward()
    # ---- HIP ----
    c1 = (mnn.Conv2d(cin, cmid, ks1, s1, p1) if k1 == "conv" else mnn.ConvTranspose2d(cin, cmid, ks1, s1, p1, op1)).to(gpu_device)
    c2 = (mnn.Conv2d(cmid, cout, ks2, s2, p2) if k2 == "conv" else mnn.ConvTranspose2d(cmid, cout, ks2, s2, p2, op2)).to(gpu_device)
    bn = mnn.BatchNorm2d(cmid).to(gpu_device).train()
    with torch.no_grad():
        c1.weight.copy_(w1), c1.bias.copy_(b1), c2.weight.copy_(w2), c2.bias.copy_(b2), bn.weight.copy_(gamma), bn.bias.copy_(beta)
    stack = mnn.Stack(c1, bn, mnn.LeakyReLU(), c2).to(gpu_device)
    xh = x.to(gpu_device).requires_grad_(True)
    out = stack(ops.to_nhwc(xh))
    assert isinstance(out, torch.Tensor)
    out_nchw = out.permute(0, 3, 1, 2)
    _close(out_nchw, y2, f"{name}: chain output", rtol=5e-4, atol=5e-5)
    (out_nchw * cot.to(gpu_device)).sum().backward()
    _close(bn.running_mean, rm, "running_mean", rtol=1e-4, atol=1e-6)
    _close(bn.running_var, rv, "running_var", rtol=1e-4, atol=1e-6)
    assert int(bn.num_batches_tracked.item()) == 1
    tol = dict(rtol=2e-3, atol=2e-4)
    _close(xh.grad, xr.grad, f"{name}: dx", **tol)
    _close(c2.weight.grad, pr[4].grad, f"{name}: dW2 (weight gradient with the normalised operand formed on load)", **tol)
    _close(c2.bias.grad, pr[5].grad, f"{name}: db2", **tol)
    _close(bn.weight.grad, pr[2].grad, f"{name}: dgamma", **tol)
    _close(bn.bias.grad, pr[3].grad, f"{name}: dbeta", **tol)
    _close(c1.weight.grad, pr[0].grad, f"{name}: dW1", **tol)
    assert float(c1.bias.grad.abs().max()) == 0.0  # a bias in front of a training-mode BatchNorm: identically zero


def test_fused_path_really_runs_and_falls_back(gpu_device, monkeypatch):
    """The fused path must be the one that runs (the *_f entry points are called, no stand-alone BatchNorm launch), a LazyBN
    never reaches a module that is not a conv, and a consumer whose kernel cannot apply the transform gets the materialised
    activation (linear layer: movae_scale_shift_act + the plain call)."""
    import movae_amd  # noqa: F401
    from movae_amd import _lib as L, nn as mnn, ops

    calls = []
    monkeypatch.setattr(L, "TRACE", lambda name, a: calls.append(name))
    torch.manual_seed(0)
    st = mnn.Stack(mnn.Conv2d(8, 16, 3, 2, 1), mnn.BatchNorm2d(16), mnn.LeakyReLU(), mnn.Conv2d(16, 32, 3, 2, 1), mnn.BatchNorm2d(32),
                   mnn.LeakyReLU()).to(gpu_device).train()
    x = torch.randn(4, 8, 8, 8, device=gpu_device)
    out = st(x)
    assert isinstance(out, ops.LazyBN)
    h = ops.flatten_nchw(ops.materialize(out))
    lin = mnn.Linear(h.shape[1], 8).to(gpu_device)
    lin(h).sum().backward()
    assert "movae_conv2d_fwd_f" in calls and "movae_bn_finalize" in calls and "movae_scale_shift_act" in calls
    assert "movae_bn_act_fwd" not in calls and "movae_bn_act_bwd" in calls
    assert any(c.endswith("dgrad_wgrad_grouped_f") or c.endswith("wgrad_grouped_f") for c in calls)
    # eval mode: running statistics, ordinary (materialised) path
    calls.clear()
    st.eval()
    with torch.no_grad():
        o = st(x)
    assert isinstance(o, torch.Tensor) and "movae_bn_act_fwd" in calls and "movae_bn_finalize" not in calls


# (kind, batch, size, cin, cout, k, stride, pad, out_pad): every BatchNorm producer of the C1 / C2 models at the config batch sizes
STAT_SHAPES = [
    ("conv", 256, 32, 3, 32, 3, 2, 1, 0), ("conv", 256, 16, 32, 64, 3, 2, 1, 0), ("conv", 256, 8, 64, 128, 3, 2, 1, 0),
    ("conv", 256, 4, 128, 256, 3, 2, 1, 0), ("conv", 256, 2, 256, 512, 3, 2, 1, 0),
    ("convT", 256, 1, 512, 256, 3, 2, 1, 1), ("convT", 256, 2, 256, 128, 3, 2, 1, 1), ("convT", 256, 4, 128, 64, 3, 2, 1, 1),
    ("convT", 256, 8, 64, 32, 3, 2, 1, 1), ("convT", 256, 16, 32, 32, 3, 2, 1, 1),
    ("conv", 128, 64, 3, 16, 3, 2, 1, 0), ("conv", 128, 32, 16, 32, 3, 2, 1, 0), ("convT", 128, 32, 16, 16, 3, 2, 1, 1),
]


@pytest.mark.parametrize("shape", STAT_SHAPES, ids=[f"{s[0]}{s[1]}x{s[2]}x{s[3]}to{s[4]}" for s in STAT_SHAPES])
def test_fused_statistics_at_config_shapes(shape, gpu_device):
    """The statistics the producer kernels emit (epilogue partials, split-K reduce, thin-channel tile sums, folded when there are
    many) against fp64 statistics of the very y they were taken from: mean and 1/sqrt(var + eps) to fp32 rounding.  (The BatchNorm
    backward amplifies an error in them a thousandfold at these shapes -- tests/test_hip_parity_full.py.)"""
    import movae_amd  # noqa: F401
    from movae_amd import nn as mnn, ops

    kind, B, size, cin, cout, k, s, p, op = shape
    torch.manual_seed(11)
    conv = (mnn.Conv2d(cin, cout, k, s, p) if kind == "conv" else mnn.ConvTranspose2d(cin, cout, k, s, p, op)).to(gpu_device)
    bn = mnn.BatchNorm2d(cout).to(gpu_device).train()
    with torch.no_grad():
        conv.bias.normal_(0.0, 1.0)  # a mean far from zero: E[y^2] - mean^2 must not lose it
    x = torch.rand(B, size, size, cin, device=gpu_device) + 0.5
    out = mnn.Stack(conv, bn, mnn.LeakyReLU()).to(gpu_device)(x)
    assert isinstance(out, ops.LazyBN)
    y = out.y.detach().double().reshape(-1, cout)
    mean = y.mean(0)
    var = ((y - mean) ** 2).mean(0)
    rstd = 1.0 / torch.sqrt(var + bn.eps)
    got_rstd = out.scale.detach().double()            # gamma = 1
    got_mean = -out.shift.detach().double() / got_rstd  # beta = 0
    # sum(y^2)/n - mean^2 in fp32 partials (folded in fp64) carries eps_fp32 * E[y^2] / var of cancellation: the bound is per channel
    bound = (5e-7 * (1.0 + mean ** 2 / var)).cpu().numpy()
    rel = ((got_rstd - rstd).abs() / rstd).cpu().numpy()
    assert (rel <= bound).all(), f"1/sqrt(var + eps): worst {rel.max():.2e} (bound {bound[rel.argmax()]:.2e})"
    merr = ((got_mean - mean).abs() / (mean.abs() + var.sqrt())).cpu().numpy()
    assert merr.max() < 5e-7, f"mean: worst error {merr.max():.2e} of |mean| + std"
