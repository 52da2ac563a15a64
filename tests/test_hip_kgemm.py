"""The block-internal split-K kernels (csrc/kgemm.h: kgemm_k FWD / BWD gather forms, kwgrad_k) against the plain PyTorch fp32
reference of the same op on the CPU, and against the tiled kernels (igemm_v2.h) they stand in for on small problems.

movae_bench_force_kgemm(1) makes every shape the family CAN serve take it (the size heuristic only picks the latency-bound
layers of the CIFAR VAE, models/vae.py:119-126,147-158), so ragged tiles, partial column tiles, tap windows, output-parity
classes, cotangent groups and the fused-BatchNorm epilogues are all exercised here at test sizes; the fused-BatchNorm chains of
tests/test_hip_fused_bn.py are re-run through it as well.  GPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture()
def force(gpu_device):
    import movae_amd  # noqa: F401
    from movae_amd import _lib as L

    lib = L.load()
    prev = lib.movae_bench_force_kgemm(1)
    yield lib
    lib.movae_bench_force_kgemm(prev)


def _close(got, want, what, rtol=2e-4, atol=2e-5):
    got, want = got.detach().cpu().double().numpy(), want.detach().cpu().double().numpy()
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol * max(1.0, float(np.abs(want).max())), err_msg=what)


def _rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


# n, ci, h, w, co, k, s, p
CONV = [
    (8, 64, 4, 4, 128, 3, 2, 1),      # the CIFAR shape family: 4x4 -> 2x2
    (16, 128, 2, 2, 256, 3, 2, 1),    # 2x2 -> 1x1: tap window (K 1152 -> 512), wgrad tiles whose tap never meets the image
    (3, 32, 8, 8, 64, 3, 1, 1),       # stride 1, rows not a multiple of 32 (192 = 6 tiles) ...
    (5, 16, 5, 7, 40, 3, 2, 1),       # ... ragged rows (60) and a partial column tile (40), 16 channels: wgrad stays tiled
    (2, 48, 6, 6, 36, 1, 1, 0),       # 1x1 taps, 36 outputs
    (4, 32, 1, 1, 64, 3, 1, 1),       # 1x1 image, 3x3 taps: the window is the centre tap alone
    (2, 24, 9, 9, 32, 4, 2, 1),       # k4 s2 (VQ / BetaTC encoders), 24 channels
]
# n, ci, h, w, co, k, s, p, op
CONVT = [
    (8, 128, 2, 2, 64, 3, 2, 1, 1),   # CIFAR decoder: 2x2 -> 4x4, classes with 1 / 2 / 2 / 4 taps
    (16, 256, 1, 1, 128, 3, 2, 1, 1), # 1x1 -> 2x2: every class keeps one tap, the others only meet padding
    (3, 32, 5, 3, 32, 4, 2, 1, 0),    # k4 s2 p1 (VQ decoders), ragged class rows (45)
    (2, 64, 4, 4, 32, 3, 2, 1, 1),
]


def _kernel(lib):
    return lib.movae_bench_last_kernel().decode()


@pytest.mark.parametrize("case", CONV)
@pytest.mark.parametrize("act", [None, "lrelu"])
def test_conv2d_through_kgemm(case, act, force, gpu_device):
    from movae_amd import ops

    n, ci, h, w, co, k, s, p = case
    x, wt, b = _rnd(n, ci, h, w, seed=1), _rnd(co, ci, k, k, seed=2, scale=0.2), _rnd(co, seed=3)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, stride=s, padding=p)
    if act:
        yr = F.leaky_relu(yr, 0.01)
    gy = _rnd(*yr.shape, seed=4)
    yr.backward(gy)
    xg = x.permute(0, 2, 3, 1).contiguous().to(gpu_device).requires_grad_(True)
    wg = wt.to(gpu_device).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bg = b.to(gpu_device).requires_grad_(True)
    y = ops.conv2d(xg, wg, bg, s, p, act, 0.01)
    assert _kernel(force).startswith("kgemm_k<0,"), _kernel(force)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(gpu_device))
    _close(y.permute(0, 3, 1, 2), yr, "y")
    _close(xg.grad.permute(0, 3, 1, 2), xr.grad, "dx (BWD gather form)")
    _close(wg.grad, wr.grad, "dw")
    _close(bg.grad, br.grad, "db")


@pytest.mark.parametrize("case", CONVT)
def test_conv_transpose2d_through_kgemm(case, force, gpu_device):
    from movae_amd import ops

    n, ci, h, w, co, k, s, p, op = case
    x, wt, b = _rnd(n, ci, h, w, seed=5), _rnd(ci, co, k, k, seed=6, scale=0.2), _rnd(co, seed=7)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, br, stride=s, padding=p, output_padding=op)
    gy = _rnd(*yr.shape, seed=8)
    yr.backward(gy)
    xg = x.permute(0, 2, 3, 1).contiguous().to(gpu_device).requires_grad_(True)
    wg = wt.to(gpu_device).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bg = b.to(gpu_device).requires_grad_(True)
    y = ops.conv_transpose2d(xg, wg, bg, s, p, op, None, 0.01)
    if yr.shape[2] % s == 0 and yr.shape[3] % s == 0:
        assert _kernel(force).startswith("kgemm_k<1,"), _kernel(force)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(gpu_device))
    _close(y.permute(0, 3, 1, 2), yr, "y (BWD gather form, output-parity classes)")
    _close(xg.grad.permute(0, 3, 1, 2), xr.grad, "dx (FWD gather form)")
    _close(wg.grad, wr.grad, "dw")
    _close(bg.grad, br.grad, "db")


@pytest.mark.parametrize("ks", [1, 2, 4, 8])
def test_every_split_degree_gives_the_same_sums_up_to_rounding(ks, force, gpu_device, monkeypatch):
    """KS = 1 / 2 / 4 / 8 (four row tiles per block ... one tile, eight reduction slices in a 512-thread block) on one problem, through the C ABI, against
    float64 on the CPU; the reduce-free fold is deterministic: two runs of one degree agree bit for bit."""
    import os
    import subprocess
    import sys

    code = f"""
import os, sys, torch
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
import movae_amd
from movae_amd import _lib as L, ops
import torch.nn.functional as F
L.load().movae_bench_force_kgemm(1)
g = torch.Generator().manual_seed(11)
x = torch.randn(12, 64, 4, 4, generator=g); w = torch.randn(96, 64, 3, 3, generator=g) * 0.1
ref = F.conv2d(x.double(), w.double(), None, stride=2, padding=1)
xg = x.permute(0, 2, 3, 1).contiguous().cuda(); wg = w.cuda().contiguous(memory_format=torch.channels_last)
y1 = ops.conv2d(xg, wg, None, 2, 1, None, 0.01); y2 = ops.conv2d(xg, wg, None, 2, 1, None, 0.01)
assert L.load().movae_bench_last_kernel().decode() == "kgemm_k<0,{8 if ks == 8 else 4},{ks},false>", L.load().movae_bench_last_kernel()
assert torch.equal(y1, y2)
err = (y1.permute(0, 3, 1, 2).cpu().double() - ref).abs().max().item() / ref.abs().max().item()
assert err < 2e-6, err
print("ok", err)
"""
    env = dict(os.environ, MOVAE_KGEMM_KS=str(ks), MOVAE_NO_REBUILD="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


def test_grouped_backward_through_kgemm(force, gpu_device):
    """Two cotangent groups at once (autojac's batched pull-back): dgrad over G * n images, grouped wgrad, against two single passes."""
    from movae_amd import ops

    n, ci, h, w, co = 8, 64, 4, 4, 96
    x = _rnd(n, h, w, ci, seed=1).to(gpu_device).requires_grad_(True)
    wt = _rnd(co, ci, 3, 3, seed=2, scale=0.1).to(gpu_device).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = ops.conv2d(x, wt, None, 2, 1, None, 0.01)
    ctx = y.grad_fn
    dy = _rnd(2, *y.shape, seed=3).to(gpu_device)
    dx, dw, _ = ops.Conv.backward_batched(ctx, 2, dy)[:3]
    for g in range(2):
        gx, gw = torch.autograd.grad(y, [x, wt], dy[g], retain_graph=True)
        _close(dx[g], gx, f"group {g} dx", rtol=1e-5, atol=1e-6)
        _close(dw[g], gw, f"group {g} dw", rtol=1e-5, atol=1e-6)


def test_weight_gradient_split_over_blocks(force, gpu_device):
    """kwgrad_k with the images split over two and three blocks per tile (partial slabs + the deterministic reduce), ragged last
    split, against the unsplit result; with accumulate the existing gradient is added exactly once."""
    from movae_amd import ops

    n, ci, h, w, co = 20, 32, 4, 4, 40
    x = _rnd(n, h, w, ci, seed=1).to(gpu_device).requires_grad_(True)
    wt = _rnd(co, ci, 3, 3, seed=2, scale=0.1).to(gpu_device).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = ops.conv2d(x, wt, None, 2, 1, None, 0.01)
    dy = _rnd(*y.shape, seed=3).to(gpu_device)
    (ref,) = torch.autograd.grad(y, [wt], dy, retain_graph=True)
    assert "kwgrad_k<" in _kernel(force), _kernel(force)
    for sp in (2, 3):
        prev = force.movae_bench_force_split(sp)
        try:
            (got,) = torch.autograd.grad(y, [wt], dy, retain_graph=True)
        finally:
            force.movae_bench_force_split(prev)
        _close(got, ref, f"dw, images split {sp} ways", rtol=1e-5, atol=1e-6)


def test_fused_batchnorm_chains_through_kgemm(force, gpu_device, monkeypatch):
    """conv -> BatchNorm(train) -> LeakyReLU -> conv with the statistics from the producer's epilogue, the normalisation applied
    while the consumer loads and the BatchNorm backward sums from the consumer's input gradient -- tests/test_hip_fused_bn.py's
    chains whose channel counts the family takes, forced through it."""
    import test_hip_fused_bn as T

    ran = 0
    for case in T.CHAINS:
        name, _, _, B, size, cin, cmid, cout = case
        if cin % 8 or cmid % 8 or cout % 4:
            continue
        T.test_conv_bn_act_conv_chain(case, True, gpu_device, monkeypatch)
        ran += 1
    assert ran >= 4


def test_batchnorm_finished_inside_the_producer_launch(force, gpu_device, monkeypatch):
    """movae_fuse_t::fin_* (opt-in: MOVAE_KGEMM_BN_FIN / movae_bench_kgemm_bn_fin): the kgemm forward's last-arriving block per
    column tile finishes the BatchNorm that follows -- no movae_bn_finalize call.  The fused-BatchNorm chains again with it ON
    (against PyTorch), and the BatchNorm's saved statistics / affine map / running statistics against the stand-alone finalize
    on one conv -> BatchNorm pair."""
    import test_hip_fused_bn as T
    from movae_amd import _lib as L
    from movae_amd import nn as mnn

    prev = force.movae_bench_kgemm_bn_fin(1)
    try:
        calls = []
        monkeypatch.setattr(L, "TRACE", lambda name, args: calls.append(name))
        ran = 0
        for case in T.CHAINS:
            name, _, _, B, size, cin, cmid, cout = case
            if cin % 8 or cmid % 8 or cout % 4:
                continue
            T.test_conv_bn_act_conv_chain(case, True, gpu_device, monkeypatch)
            ran += 1
        assert ran >= 4
        assert "movae_bn_finalize" not in calls, "the producer was asked to finish the BatchNorm itself"
        monkeypatch.setattr(L, "TRACE", None)

        res = {}
        for on in (1, 0):
            force.movae_bench_kgemm_bn_fin(on)
            torch.manual_seed(5)
            st = mnn.Stack(mnn.Conv2d(64, 96, 3, stride=2, padding=1), mnn.BatchNorm2d(96), mnn.LeakyReLU()).to(gpu_device).train()
            x = _rnd(24, 8, 8, 64, seed=9).to(gpu_device)
            out = st(x)
            res[on] = (out.scale.clone(), out.shift.clone(), st[1].running_mean.clone(), st[1].running_var.clone(),
                       int(st[1].num_batches_tracked), mnn.ops.materialize(out).clone())
        for a, b, what in zip(res[1], res[0], ("scale", "shift", "running_mean", "running_var", "num_batches_tracked", "output")):
            if isinstance(a, int):
                assert a == b == 1, what
            else:  # (the two fold the fp64 partial sums in different orders: equal to the last bit or one ulp apart)
                np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=3e-7, atol=1e-7, err_msg=what)
    finally:
        force.movae_bench_kgemm_bn_fin(prev)
