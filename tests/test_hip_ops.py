"""Parity of every HIP kernel family (through the C ABI via mo-vae_amd/ops.py) against plain
PyTorch fp32 on the CPU and against the oracle / golden vectors.  GPU only."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

pytestmark = pytest.mark.gpu

FP32_TOL = dict(rtol=2e-4, atol=2e-5)  # fp32 tolerance for O(1) activations with K <= ~5k reductions


@pytest.fixture(scope="module")
def M(gpu_device):
    import movae_amd  # noqa: F401
    from movae_amd import aggregation, ops

    movae_amd.load_library()
    return ops, aggregation


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def nhwc(x):  # NCHW cpu -> NHWC cuda leaf
    return x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)


def cl(w):  # weight cpu -> channels_last cuda leaf
    return w.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)


def back(t):  # NHWC cuda -> NCHW cpu
    return t.detach().cpu().permute(0, 3, 1, 2)


def close(a, b, what="", rtol=None, atol=None):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = max(1.0, float(b.abs().max()))
    np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=rtol or FP32_TOL["rtol"], atol=(atol or FP32_TOL["atol"]) * scale,
                               err_msg=what)


CONV_CASES = [
    # n, ci, h, w, co, k, s, p
    (2, 3, 16, 16, 8, 3, 2, 1),      # generic gather path (ci = 3)
    (3, 16, 9, 7, 24, 3, 2, 1),      # vector path, odd sizes, N not multiple of tile
    (2, 32, 8, 8, 64, 3, 1, 1),      # stride 1
    (2, 3, 16, 16, 16, 4, 2, 1),     # k4 s2 (VQ / BetaTC encoders)
    (4, 16, 4, 4, 3, 3, 1, 1),       # narrow output (final conv, co = 3)
    (2, 48, 6, 6, 32, 1, 1, 0),      # 1x1
    (5, 64, 2, 2, 128, 3, 2, 1),     # 2x2 -> 1x1 (CIFAR VAE tail), split-K path
    (2, 20, 5, 5, 12, 3, 2, 1),      # ci not multiple of 16 nor 4
    (3, 3, 13, 11, 32, 3, 2, 1),     # 3-channel input, 32 outputs: thin wgrad sweep kernel (thin = big side), ragged tiles
    (2, 3, 16, 16, 64, 4, 2, 1),     # ... with 4x4 taps and two 32-channel blocks
    (2, 32, 9, 7, 3, 3, 1, 1),       # 3-channel output, stride 1: thin wgrad sweep kernel (thin = small side)
    (2, 64, 40, 36, 3, 3, 1, 1),     # ... several tiles per image, two channel blocks
    (4, 32, 1, 1, 48, 3, 1, 1),      # 1x1 input, 3x3 taps: the tap window is the centre tap alone (and it is NOT a linear layer)
    (3, 16, 2, 2, 24, 4, 2, 1),      # 2x2 -> 1x1 with 4x4 taps: window rows/cols 1..2 of the stored kernel
    (2, 256, 2, 2, 512, 3, 2, 1),    # the C2 layer itself: K 2304 -> 1024
    # thin-channel MFMA kernels (blocks of 256 output pixels of one image):
    (2, 3, 32, 32, 32, 3, 2, 1),     # thin_in_mfma_k fwd, 3x3 stride 2, a block = the whole 16x16 output image (CIFAR first conv)
    (1, 3, 64, 64, 64, 4, 2, 1),     # ... 4x4 taps, 64 outputs = two column tiles, 8 rows of 32 per block (VQ / BetaTC first conv)
    (1, 3, 4, 512, 32, 3, 1, 1),     # ... blocks are row SEGMENTS (Wo a multiple of 256), stride 1
    (2, 32, 16, 16, 3, 3, 1, 1),     # thin_out_mfma_k fwd; its input gradient is thin_in_mfma_k BWD (32 outputs)
    (1, 64, 8, 32, 3, 3, 1, 1),      # ... two channel chunks fwd, two column tiles in the input gradient
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("act", [None, "lrelu"])
def test_conv2d_fwd_bwd(M, case, act):
    ops, _ = M
    n, ci, h, w, co, k, s, p = case
    x, wt, b = rnd(n, ci, h, w, seed=1), rnd(co, ci, k, k, seed=2, scale=0.2), rnd(co, seed=3)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, stride=s, padding=p)
    if act:
        yr = F.leaky_relu(yr, 0.01)
    gy = rnd(*yr.shape, seed=4)
    yr.backward(gy)
    xg, wg, bg = nhwc(x), cl(wt), b.cuda().requires_grad_(True)
    y = ops.conv2d(xg, wg, bg, s, p, act, 0.01)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().cuda())
    close(back(y), yr, "y")
    close(back(xg.grad), xr.grad, "dx")
    close(wg.grad, wr.grad, "dw")
    close(bg.grad, br.grad, "db")


CONVT_CASES = [
    # n, ci, h, w, co, k, s, p, op
    (2, 16, 4, 4, 8, 3, 2, 1, 1),
    (3, 32, 1, 1, 16, 3, 2, 1, 1),    # 1x1 -> 2x2 (CIFAR VAE decoder head)
    (2, 8, 5, 6, 12, 3, 2, 1, 1),     # generic path
    (2, 16, 8, 8, 3, 4, 2, 1, 0),     # k4 s2 p1 (VQ decoders), co = 3
    (2, 32, 4, 4, 32, 4, 2, 1, 0),
    (2, 32, 9, 5, 3, 4, 2, 1, 0),     # co = 3 with 32 input channels: thin wgrad sweep kernel through the convT mapping
    (2, 128, 20, 20, 3, 4, 2, 1, 0),  # ... four channel blocks, several tiles; LDS-tiled thin-output kernel with ragged tiles
    (2, 64, 16, 16, 3, 4, 2, 1, 0),   # thin-output kernel, exact tiles
    (3, 32, 8, 8, 3, 3, 2, 1, 1),     # thin-output kernel, 3x3 taps with output padding
    (2, 24, 16, 16, 2, 4, 2, 1, 0),   # thin-output kernel, 2 outputs, channels not a multiple of the chunk
    (2, 128, 16, 16, 3, 4, 2, 1, 0),  # VQ last layer: the input gradient is thin_in_mfma_k in FWD form with 128 outputs (four column tiles)
]


@pytest.mark.parametrize("case", CONVT_CASES)
@pytest.mark.parametrize("act", [None, "tanh"])
def test_conv_transpose2d_fwd_bwd(M, case, act):
    ops, _ = M
    n, ci, h, w, co, k, s, p, op = case
    x, wt, b = rnd(n, ci, h, w, seed=5), rnd(ci, co, k, k, seed=6, scale=0.2), rnd(co, seed=7)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, br, stride=s, padding=p, output_padding=op)
    if act:
        yr = torch.tanh(yr)
    gy = rnd(*yr.shape, seed=8)
    yr.backward(gy)
    xg, wg, bg = nhwc(x), cl(wt), b.cuda().requires_grad_(True)
    y = ops.conv_transpose2d(xg, wg, bg, s, p, op, act, 0.01)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().cuda())
    close(back(y), yr, "y")
    close(back(xg.grad), xr.grad, "dx")
    close(wg.grad, wr.grad, "dw")
    close(bg.grad, br.grad, "db")


@pytest.mark.parametrize("shape", [(7, 33, 19), (256, 512, 128), (4, 2048, 8), (32, 4096, 256)])
def test_linear_fwd_bwd(M, shape):
    ops, _ = M
    n, fin, fout = shape
    x, wt, b = rnd(n, fin, seed=1), rnd(fout, fin, seed=2, scale=0.05), rnd(fout, seed=3)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.linear(xr, wr, br)
    gy = rnd(n, fout, seed=4)
    yr.backward(gy)
    xg, wg, bg = x.cuda().requires_grad_(True), wt.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = ops.linear(xg, wg, bg)
    y.backward(gy.cuda())
    close(y, yr, "y")
    close(xg.grad, xr.grad, "dx")
    close(wg.grad, wr.grad, "dw")
    close(bg.grad, br.grad, "db")


@pytest.mark.parametrize("shape", [(4, 8, 5, 5), (6, 32, 8, 8), (3, 20, 3, 3), (64, 512, 1, 1), (2, 3, 4, 4),
                                   # rows > 1024: the three-launch path (partials / final / apply); 1024: the boundary of
                                   # the one-launch kernels (whole column in registers)
                                   (32, 16, 16, 16), (5, 24, 31, 33), (16, 32, 16, 16), (9, 6, 24, 24), (4, 8, 16, 16),
                                   (5, 8, 15, 15), (2, 4, 1, 1), (3, 12, 17, 19)])
def test_batchnorm_act_train(M, shape):
    ops, _ = M
    n, c, h, w = shape
    x = rnd(n, c, h, w, seed=11) * 2 + 0.5
    g, b = rnd(c, seed=12).abs() + 0.5, rnd(c, seed=13)
    bn = torch.nn.BatchNorm2d(c)
    with torch.no_grad():
        bn.weight.copy_(g)
        bn.bias.copy_(b)
    xr = x.clone().requires_grad_(True)
    yr = F.leaky_relu(bn(xr), 0.01)
    gy = rnd(*yr.shape, seed=14)
    yr.backward(gy)
    xg = nhwc(x)
    gg, bg = g.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    rm, rv = torch.zeros(c).cuda(), torch.ones(c).cuda()
    nbt = torch.tensor(41, dtype=torch.long).cuda()
    y = ops.batch_norm_act(xg, gg, bg, rm, rv, True, 1e-5, 0.1, "lrelu", 0.01, nbt)
    assert int(nbt.item()) == 42  # num_batches_tracked is incremented inside the statistics kernel
    y.backward(gy.permute(0, 2, 3, 1).contiguous().cuda())
    close(back(y), yr, "y")
    close(back(xg.grad), xr.grad, "dx", rtol=1e-3)
    close(gg.grad, bn.weight.grad, "dgamma", rtol=1e-3)
    close(bg.grad, bn.bias.grad, "dbeta", rtol=1e-3)
    close(rm, bn.running_mean, "running_mean")
    close(rv, bn.running_var, "running_var")
    # eval mode uses the running statistics
    bn.eval()
    ye = ops.batch_norm_act(xg.detach(), gg.detach(), bg.detach(), rm, rv, False, 1e-5, 0.1, None, 0.01, nbt)
    close(back(ye), bn(x), "eval")
    assert int(nbt.item()) == 42  # untouched in eval mode


def test_layout_roundtrip_and_flatten(M):
    ops, _ = M
    x = rnd(3, 5, 4, 6, seed=21)
    xg = x.cuda()
    y = ops.NchwToNhwc.apply(xg)
    assert torch.equal(y.cpu(), x.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(ops.NhwcToNchw.apply(y).cpu(), x)
    assert torch.equal(ops.flatten_nchw(y).cpu(), x.flatten(1))
    assert torch.equal(ops.unflatten_nchw(x.flatten(1).cuda(), 5, 4, 6).cpu(), x.permute(0, 2, 3, 1).contiguous())


def test_elementwise(M):
    ops, _ = M
    a, b = rnd(2, 4, 4, 7, seed=1), rnd(2, 4, 4, 7, seed=2)
    assert torch.equal(ops.add(a.cuda(), b.cuda()).cpu(), a + b)
    c = ops.concat_channels(a.cuda().requires_grad_(True), rnd(2, 4, 4, 3, seed=3).cuda())
    assert torch.equal(c.cpu()[..., :7], a) and c.shape[-1] == 10
    for kind, ref in [("relu", torch.relu), ("tanh", torch.tanh), ("sigmoid", torch.sigmoid),
                      ("lrelu", lambda t: F.leaky_relu(t, 0.01))]:
        xr = a.clone().requires_grad_(True)
        ref(xr).backward(b)
        xg = a.cuda().requires_grad_(True)
        y = ops.activation(xg, kind, 0.01)
        y.backward(b.cuda())
        close(y, ref(a), kind)
        close(xg.grad, xr.grad, kind + " grad")
    mu, lv, eps = rnd(5, 9, seed=4), rnd(5, 9, seed=5) * 0.5, rnd(5, 9, seed=6)
    mr, lr = mu.clone().requires_grad_(True), lv.clone().requires_grad_(True)
    zr = mr + eps * torch.exp(0.5 * lr)
    zr.backward(a.reshape(-1)[:45].reshape(5, 9))
    mg, lg = mu.cuda().requires_grad_(True), lv.cuda().requires_grad_(True)
    z = ops.reparameterize(mg, lg, eps.cuda())
    z.backward(a.reshape(-1)[:45].reshape(5, 9).cuda())
    close(z, zr, "z")
    close(mg.grad, mr.grad, "dmu")
    close(lg.grad, lr.grad, "dlv")


@pytest.mark.parametrize("name,rkey", [("mse", "r_tanh"), ("l1", "r_tanh"), ("smooth_l1", "r_sl1"), ("bce", "r_sig")])
def test_recon_losses_match_reference_fixture(M, name, rkey):
    """golden values come from the reference's utils/objectives.py (incl. the saturated BCE points)."""
    ops, _ = M
    fx = load_golden("objectives")
    x = torch.from_numpy(fx["x"]).cuda()
    r = torch.from_numpy(fx[rkey]).cuda().requires_grad_(True)
    v = ops.recon_loss(r, x, name, 1.0)
    v.backward()
    np.testing.assert_allclose(v.item(), fx[f"{name}.value"], rtol=2e-6)
    np.testing.assert_allclose(r.grad.cpu().numpy(), fx[f"{name}.grad"], rtol=1e-5, atol=1e-9)
    # lambda scaling and upstream gradient scaling
    r2 = torch.from_numpy(fx[rkey]).cuda().requires_grad_(True)
    (3.0 * ops.recon_loss(r2, x, name, 0.5)).backward()
    np.testing.assert_allclose(r2.grad.cpu().numpy(), 1.5 * fx[f"{name}.grad"], rtol=1e-5, atol=1e-9)


def test_kl_matches_reference_fixture(M):
    ops, _ = M
    fx = load_golden("objectives")
    mu = torch.from_numpy(fx["kl.mu"]).cuda().requires_grad_(True)
    lv = torch.from_numpy(fx["kl.log_var"]).cuda().requires_grad_(True)
    v = ops.kl_divergence(mu, lv, 1.0)
    v.backward()
    np.testing.assert_allclose(v.item(), fx["kl.value"], rtol=1e-6)
    np.testing.assert_allclose(mu.grad.cpu().numpy(), fx["kl.gmu"], rtol=1e-6, atol=1e-9)
    # exp(lv) - 1 cancels for small lv: allow one fp32 ulp of exp() in absolute terms
    np.testing.assert_allclose(lv.grad.cpu().numpy(), fx["kl.glv"], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("B,D,ds", [(5, 6, 1000), (32, 128, 1281167), (17, 70, 50000), (64, 16, 30000)])
def test_tc_decomposition_vs_oracle(M, B, D, ds):
    ops, _ = M
    from oracle import nets

    z, mu, lv = rnd(B, D, seed=1), rnd(B, D, seed=2) * 0.5, rnd(B, D, seed=3) * 0.3
    liw = nets.log_importance_weights(B, ds)
    zr, mr, lr = [t.clone().requires_grad_(True) for t in (z, mu, lv)]
    mat = nets._log_density_gaussian(zr.view(B, 1, D), mr.view(1, B, D), lr.view(1, B, D)) + liw.view(B, B, 1)
    lqz = torch.logsumexp(mat.sum(2), dim=1)
    lpq = torch.logsumexp(mat, dim=1).sum(1)
    lqzx = nets._log_density_gaussian(zr, mr, lr).sum(1)
    lpz = nets._log_density_gaussian(zr, torch.zeros_like(zr), torch.zeros_like(zr)).sum(1)
    ref = torch.stack([(lqzx - lqz).mean(), (lqz - lpq).mean(), (lpq - lpz).mean()])
    gw = torch.tensor([0.7, -1.3, 0.4])
    (ref * gw).sum().backward()
    zg, mg, lg = [t.cuda().requires_grad_(True) for t in (z, mu, lv)]
    out = ops.tc_decomposition(zg, mg, lg, liw.cuda())
    (out * gw.cuda()).sum().backward()
    close(out, ref, "terms", rtol=2e-5, atol=2e-5)
    close(zg.grad, zr.grad, "dz", rtol=2e-3, atol=2e-5)
    close(mg.grad, mr.grad, "dmu", rtol=2e-3, atol=2e-5)
    close(lg.grad, lr.grad, "dlv", rtol=2e-3, atol=2e-5)


@pytest.mark.parametrize("rows,K,D", [(48, 16, 8), (300, 512, 64), (1000, 100, 32), (77, 40, 10), (4096, 512, 64)])
def test_vector_quantize_vs_oracle(M, rows, K, D):
    ops, _ = M
    from oracle import nets

    x = rnd(rows, D, seed=1).reshape(1, rows, 1, D)  # NHWC with H = rows
    E = (torch.rand(K, D, generator=torch.Generator().manual_seed(2)) * 2 - 1) * 1.5
    xr, Er = x.permute(0, 3, 1, 2).contiguous().requires_grad_(True), E.clone().requires_grad_(True)
    qr, cr, er, ir = nets.vector_quantize(xr, Er)
    gq = rnd(*qr.shape, seed=3)
    (qr * gq).sum().backward(retain_graph=True)
    (0.25 * cr + 2.0 * er).backward()
    xg, Eg = x.cuda().requires_grad_(True), E.cuda().requires_grad_(True)
    q, c, e, idx, used = ops.vector_quantize(xg, Eg)
    ((q * gq.permute(0, 2, 3, 1).cuda()).sum() + 0.25 * c + 2.0 * e).backward()
    agree = (idx.cpu() == ir).float().mean().item()
    assert agree >= 0.999, f"index agreement {agree}"
    assert int(used.item()) == torch.unique(ir).numel() or agree < 1.0
    close(c, cr, "commitment", rtol=1e-4)
    close(e, er, "embedding", rtol=1e-4)
    if agree == 1.0:
        close(q.permute(0, 3, 1, 2), qr, "q")
        close(xg.grad.permute(0, 3, 1, 2), xr.grad, "dx", rtol=1e-3)
        close(Eg.grad, Er.grad, "dE", rtol=1e-3, atol=1e-5)


def test_vq_codebook_gradient_skewed_usage_and_bit_reproducible(M):
    """Few codes in use (the early-training regime): the sorted segmented sum must match a float64 scatter-add and give
    the same bits on every run (there are no float atomics)."""
    ops, _ = M
    rows, K, D = 20000, 64, 16
    E = torch.randn(K, D, generator=torch.Generator().manual_seed(5)) * 4.0
    pick = torch.tensor([3, 3, 3, 41, 3, 17])[torch.randint(0, 6, (rows,), generator=torch.Generator().manual_seed(6))]
    x = (E[pick] + 0.05 * torch.randn(rows, D, generator=torch.Generator().manual_seed(7))).reshape(1, rows, 1, D)
    outs = []
    for _ in range(3):
        xg, Eg = x.cuda().requires_grad_(True), E.cuda().requires_grad_(True)
        q, c, e, idx, used = ops.vector_quantize(xg, Eg)
        (0.25 * c + 2.0 * e).backward()
        outs.append(Eg.grad.cpu().clone())
    assert torch.equal(idx.cpu().reshape(-1), pick) and int(used.item()) == 3
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    want = torch.zeros(K, D, dtype=torch.float64)
    want.index_add_(0, pick, (E[pick].double() - x.reshape(rows, D).double()))
    want *= 2.0 * 2.0 / (rows * D)
    close(outs[0], want.float(), "dE", rtol=2e-4, atol=1e-6)
    assert not outs[0][0].any() and not outs[0][63].any()  # unused codes get exact zeros without a memset


def test_vq_first_index_on_ties(M):
    ops, _ = M
    E = torch.zeros(40, 8)
    E[3] = 1.0
    E[17] = 1.0  # duplicate code: argmin must return the first
    x = torch.ones(1, 5, 1, 8)
    _, _, _, idx, used = ops.vector_quantize(x.cuda(), E.cuda())
    assert idx.cpu().tolist() == [3] * 5 and int(used.item()) == 1


# ---------------------------------------------------------------- aggregation
CASES = ["kat", "k2", "k3", "k3_zero_row", "k4", "k4_conflict", "k5_rankdef", "k2_parallel"]


@pytest.mark.parametrize("case", CASES)
def test_gramian_and_combine(M, case):
    _, agg = M
    fx = load_golden("weightings")
    J = torch.from_numpy(fx[f"{case}.J"])
    G = agg.compute_gramian(J.cuda())
    close(G, torch.from_numpy(fx[f"{case}.G"]), "G", rtol=1e-5)
    w = torch.linspace(0.3, 1.7, J.shape[0])
    close(agg.combine(J.cuda(), w.cuda()), w @ J, "combine", rtol=1e-5)


def test_gramian_large_unaligned(M):
    _, agg = M
    J = rnd(3, 1_000_003, seed=9)
    buf = torch.zeros(3, 1_000_004).cuda()
    buf[:, :1_000_003] = J.cuda()
    G = agg.compute_gramian(buf[:, :1_000_003])
    close(G, (J.double() @ J.double().T).float(), "G", rtol=2e-6)
    w = torch.tensor([0.5, -1.0, 2.0])
    close(agg.combine(buf[:, :1_000_003], w.cuda()), w @ J, "g", rtol=1e-5)
    sim = agg.gd_similarity(buf[:, :1_000_003], w.cuda())
    ref = F.cosine_similarity(J.T @ w, J.mean(0), dim=0)
    np.testing.assert_allclose(sim.item(), ref.item(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("nt", ["none", "l2", "loss", "loss+"])
def test_mgda_weights_match_reference_code(M, case, nt):
    _, agg = M
    fx = load_golden("weightings")
    W = agg.MGDAWeighting(norm_type=nt)
    W.set_losses(torch.from_numpy(fx[f"{case}.losses"]).cuda())
    w = W(torch.from_numpy(fx[f"{case}.G"]).cuda())
    np.testing.assert_allclose(w.cpu().numpy(), fx[f"{case}.mgda.{nt}"], rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("case", CASES)
def test_stable_mgda_weights_match_reference_code(M, case):
    """StableMGDA (eigen regularisation, utils/torchmoo/mgda.py:286-317) vs vectors from the reference's MGDAWeighting."""
    _, agg = M
    fx = load_golden("weightings")
    w = agg.MGDAWeighting(norm_type="none", stable=True)(torch.from_numpy(fx[f"{case}.G"]).cuda())
    np.testing.assert_allclose(w.cpu().numpy(), fx[f"{case}.mgda.stable"], rtol=5e-4, atol=5e-6)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("sm", ["min", "median", "rmse"])
def test_aligned_mtl_weights_match_reference_code(M, case, sm):
    _, agg = M
    fx = load_golden("weightings")
    w = agg.AlignedMTLWeighting(None, scale_mode=sm)(torch.from_numpy(fx[f"{case}.G"]).cuda())
    want = fx[f"{case}.amtl.{sm}"]
    np.testing.assert_allclose(w.cpu().numpy(), want, rtol=5e-3, atol=1e-5 * np.abs(want).max())


@pytest.mark.parametrize("case", CASES)
def test_upgrad_weights_vs_oracle(M, case):
    _, agg = M
    from oracle import aggregation as OA

    fx = load_golden("weightings")
    G = fx[f"{case}.G"]
    w = agg.UPGradWeighting()(torch.from_numpy(G).cuda())
    np.testing.assert_allclose(w.cpu().numpy(), OA.upgrad_weights(G), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("norm", ["min_l2", "cosine"])
def test_nupgrad_pnupgrad_weights_vs_oracle(M, case, norm):
    """NUPGrad / both PNUPGrad branches (movae_weights_upgrad_norm) vs the oracle, whose normalisations are
    pinned by tests/golden/agg_variants.npz."""
    _, agg = M
    from oracle import aggregation as OA

    G = load_golden("weightings")[f"{case}.G"]
    w = agg.UPGradWeighting(norm=norm)(torch.from_numpy(G).cuda())
    want = OA.upgrad_weights(G, norm=norm)
    np.testing.assert_allclose(w.cpu().numpy(), want, rtol=2e-4, atol=1e-6 * max(1.0, np.abs(want).max()))


def test_nupgrad_pnupgrad_comfort_aggregators(M):
    _, agg = M
    from oracle import aggregation as OA

    fx = load_golden("weightings")
    J = torch.from_numpy(fx["k4_conflict.J"])
    G = fx["k4_conflict.G"]
    g = agg.NUPGrad()(J.cuda())
    close(g, torch.as_tensor(OA.upgrad_weights(G, norm="min_l2"), dtype=torch.float32) @ J, "nupgrad", rtol=2e-4)
    # PNUPGrad draws torch's CPU generator exactly like the reference: seed -> same branch sequence
    P = agg.PNUPGrad(prob=0.5)
    torch.manual_seed(123)
    coins = [torch.rand(1).item() < 0.5 for _ in range(6)]
    torch.manual_seed(123)
    for c in coins:
        want = torch.as_tensor(OA.upgrad_weights(G, norm="cosine" if c else "min_l2"), dtype=torch.float32) @ J
        close(P(J.cuda()), want, "pnupgrad", rtol=2e-4)
    assert any(coins) and not all(coins)
    # COMFORT = (1 - beta) MGDA + beta UPGrad, beta from the epoch schedule; hooks see the MGDA weighting
    C = agg.COMFORT(mgda_norm_type="l2")
    seen = {}
    C.weighting.register_forward_hook(lambda m, i, o: seen.update(w=o.clone()))
    for epoch, total in [(1, 10), (4, 10), (10, 10)]:
        C.set_epoch(epoch, total)
        beta = OA.beta_schedule(epoch, total)
        w = (1 - beta) * np.asarray(OA.mgda_weights(G, "l2"), dtype=np.float64) + beta * OA.upgrad_weights(G)
        close(C(J.cuda()), torch.as_tensor(w, dtype=torch.float32) @ J, f"comfort epoch {epoch}", rtol=5e-4)
        np.testing.assert_allclose(seen["w"].cpu().numpy(), OA.mgda_weights(G, "l2"), rtol=2e-4, atol=2e-6)
    Cs = agg.COMFORT(mgda_stable=True, mgda_min_eigenvalue_eps=1e-10)  # StableMGDA branch (comfort.py:100-106)
    Cs.set_epoch(1, 10)
    beta = OA.beta_schedule(1, 10)
    w = (1 - beta) * np.asarray(OA.mgda_weights(G, "none", stable=True), dtype=np.float64) + beta * OA.upgrad_weights(G)
    close(Cs(J.cuda()), torch.as_tensor(w, dtype=torch.float32) @ J, "comfort stable", rtol=5e-4)


@pytest.mark.parametrize("case", ["k2", "k2_parallel", "k3", "k3_zero_row", "k4", "k4_conflict", "k5_rankdef"])
def test_dualproj_pcgrad_imtlg_vs_oracle(M, case):
    """torchjd's DualProj / PCGrad / IMTLG (main.py:1196-1222) vs the oracle's restatement (pinned only by torchjd's
    published usage example, see test_oracle_golden): weights on the golden Gramians, PCGrad with the same seeded
    torch.randperm draws, and the K = 2 usage example itself."""
    _, agg = M
    from oracle import aggregation as OA

    fx = load_golden("weightings")
    G = fx[f"{case}.G"]
    Gd = torch.from_numpy(G).cuda()
    scale = max(1.0, float(np.abs(G).max()))
    want = OA.dualproj_weights(G)
    np.testing.assert_allclose(agg.DualProjWeighting()(Gd).cpu().numpy(), want, rtol=2e-4, atol=1e-6 * max(1.0, np.abs(want).max()))
    pref = np.linspace(0.5, 1.5, len(G))
    want = OA.dualproj_weights(G, pref=pref)
    np.testing.assert_allclose(agg.DualProjWeighting(pref_vector=torch.tensor(pref, dtype=torch.float32))(Gd).cpu().numpy(), want,
                               rtol=2e-4, atol=1e-6 * max(1.0, np.abs(want).max()))
    for seed in (0, 1, 2):
        torch.manual_seed(seed)
        want = OA.pcgrad_weights(G)
        torch.manual_seed(seed)
        np.testing.assert_allclose(agg.PCGradWeighting()(Gd).cpu().numpy(), want, rtol=1e-5, atol=1e-6)
    for c in (0.5, 1.0):
        want = OA.cagrad_weights(G, c)
        got = agg.CAGradWeighting(c)(Gd).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=1e-5 * max(1.0, np.abs(want).max()), err_msg=f"cagrad c={c}")
    want = OA.imtlg_weights(G)
    got = agg.IMTLGWeighting()(Gd).cpu().numpy()
    if np.linalg.matrix_rank(G.astype(np.float64), tol=len(G) * 1.2e-7 * np.linalg.norm(G, 2) * 10) == len(G):
        np.testing.assert_allclose(got, want, rtol=5e-3, atol=1e-5 * scale)  # pinv amplifies fp32 noise by cond(G)
    else:
        assert np.isfinite(got).all() and abs(got.sum() - 1.0) < 1e-4 or not got.any()


def test_aggregator_docstring_kats(M):
    """utils/torchmoo/mgda.py:54-86 and nupgrad.py:58-62."""
    _, agg = M
    J = torch.tensor([[-4.0, 1.0, 1.0], [6.0, 1.0, 1.0]]).cuda()
    np.testing.assert_allclose(agg.UPGrad()(J).cpu().numpy(), [0.2929, 1.9004, 1.9004], atol=5e-5)
    np.testing.assert_allclose(agg.MGDA()(J).cpu().numpy(), [0.0, 1.0, 1.0], atol=1e-5)
    # torchjd's usage examples for the same matrix
    np.testing.assert_allclose(agg.DualProj()(J).cpu().numpy(), [0.5563, 1.1109, 1.1109], atol=5e-5)
    np.testing.assert_allclose(agg.PCGrad()(J).cpu().numpy(), [0.5848, 3.8012, 3.8012], atol=5e-5)
    np.testing.assert_allclose(agg.IMTLG()(J).cpu().numpy(), [0.0767, 1.0, 1.0], atol=5e-5)
    np.testing.assert_allclose(agg.CAGrad(c=0.5)(J).cpu().numpy(), [0.1835, 1.2041, 1.2041], atol=5e-5)
    np.testing.assert_allclose(agg.MGDA(norm_type="l2")(J).cpu().numpy(), [1.0, 1.0, 1.0], atol=1e-5)
    A = agg.MGDA(norm_type="loss")
    A.set_losses(torch.tensor([0.5, 2.0]).cuda())
    np.testing.assert_allclose(A(J).cpu().numpy(), [3.49, 1.0, 1.0], atol=5e-4)
    A = agg.MGDA(norm_type="loss+")
    A.set_losses(torch.tensor([0.5, 2.0]).cuda())
    np.testing.assert_allclose(A(J).cpu().numpy(), [4.1606, 1.0, 1.0], atol=5e-4)
    np.testing.assert_allclose(agg.Sum()(J).cpu().numpy(), [2.0, 2.0, 2.0], atol=1e-6)
    np.testing.assert_allclose(agg.Mean()(J).cpu().numpy(), [1.0, 1.0, 1.0], atol=1e-6)
    with pytest.raises(RuntimeError):
        agg.MGDA(norm_type="loss")(J)  # losses not set


def test_weighting_forward_hook_sees_jacobian_and_weights(M):
    """main.py:1248-1250 registers hooks on aggregator.weighting: inputs[0] is J, output is w."""
    _, agg = M
    J = torch.tensor([[-4.0, 1.0, 1.0], [6.0, 1.0, 1.0]]).cuda()
    seen = {}
    A = agg.UPGrad()
    A.weighting.register_forward_hook(lambda m, inp, out: seen.update(J=inp[0], w=out))
    A(J)
    assert seen["J"].shape == (2, 3) and seen["w"].shape == (2,)


def test_adam_and_clip(M):
    import movae_amd._lib as L

    n = 10007
    p, g = rnd(n, seed=1), rnd(n, seed=2)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-3)
    pg, m, v = p.cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    lib = L.load()
    for step in range(1, 4):
        gi = g * step
        pr.grad = gi.clone()
        opt.step()
        L.check(lib.movae_adam_step(pg.data_ptr(), gi.cuda().data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-3, 0.9, 0.999,
                                    1e-8, 0.0, 0, step, L.stream_ptr(pg.device)))
    close(pg, pr, "adam", rtol=1e-5, atol=1e-6)
    gg = g.cuda().clone()
    ss = torch.empty((), device="cuda")
    ws = L.workspace(gg.device)
    L.check(lib.movae_sumsq(gg.data_ptr(), n, ss.data_ptr(), ws.data_ptr(), ws.numel(), L.stream_ptr(gg.device)))
    np.testing.assert_allclose(ss.item(), (g.double() ** 2).sum().item(), rtol=1e-6)
    L.check(lib.movae_scale_by_clip(gg.data_ptr(), n, ss.data_ptr(), 1.0, L.stream_ptr(gg.device)))
    gr = g.clone().requires_grad_(True)
    gr.grad = g.clone()
    torch.nn.utils.clip_grad_norm_([gr], 1.0)
    close(gg, gr.grad, "clip", rtol=1e-5)


@pytest.mark.parametrize("mode", ["adam", "adam_wd", "adamw", "adam_device_step"])
def test_fused_adam_matches_torch_adam(M, mode):
    """optim.FusedAdam (movae_adam_multi, one launch for the whole list) vs torch.optim.Adam/AdamW on CPU: ragged
    sizes, a channels_last conv weight, a parameter whose grad is None on one step (own bias correction), an lr
    change between steps, and a state_dict round trip into torch's own optimizer."""
    from movae_amd.optim import FusedAdam, FusedAdamW

    shapes = [(7,), (64, 32, 3, 3), (1,), (1023,), (130, 5), (4096 * 3 + 1,)]
    ref = [rnd(*s, seed=10 + i) for i, s in enumerate(shapes)]
    ref[1] = ref[1].contiguous(memory_format=torch.channels_last)
    pr = [t.clone().requires_grad_(True) for t in ref]
    pg = [t.cuda().requires_grad_(True) for t in ref]
    assert pg[1].is_contiguous(memory_format=torch.channels_last) and not pg[1].is_contiguous()
    kw = dict(lr=1e-3, weight_decay=0.05 if mode in ("adam_wd", "adamw") else 0.0)
    if mode == "adamw":
        o_ref, o_gpu = torch.optim.AdamW(pr, **kw), FusedAdamW(pg, **kw)
    else:
        o_ref = torch.optim.Adam(pr, **kw)
        o_gpu = FusedAdam(pg, device_step=(mode == "adam_device_step"), **kw)
    for step in range(1, 6):
        if step == 4:
            for o in (o_ref, o_gpu):
                o.param_groups[0]["lr"] = 3e-4
        for i, (a, b) in enumerate(zip(pr, pg)):
            gi = rnd(*shapes[i], seed=100 * step + i) * (0.5 + step)
            skip = (i == 3 and step == 2 and mode != "adam_device_step")  # one shared device counter in device_step mode
            a.grad = None if skip else gi.clone()
            b.grad = None if skip else gi.cuda()
        o_ref.step()
        o_gpu.step()
    for i, (a, b) in enumerate(zip(pr, pg)):
        close(b, a, f"{mode} param {i}", rtol=2e-5, atol=2e-6)
    sd = o_gpu.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}
    assert float(sd["state"][0]["step"]) == 5.0
    close(sd["state"][1]["exp_avg"], o_ref.state_dict()["state"][1]["exp_avg"], "exp_avg", rtol=1e-5, atol=1e-7)
    twin = torch.optim.Adam([t.detach().clone().requires_grad_(True) for t in pg], lr=1e-3)
    twin.load_state_dict(sd)  # checkpoints are interchangeable with torch's optimizer
    # and back: torch's state loads into FusedAdam and the next step continues from it
    o2 = FusedAdam(pg, lr=3e-4, weight_decay=kw["weight_decay"], decoupled_weight_decay=(mode == "adamw"),
                   device_step=(mode == "adam_device_step"))
    o2.load_state_dict(sd)
    for i, (a, b) in enumerate(zip(pr, pg)):
        gi = rnd(*shapes[i], seed=999 + i)
        a.grad, b.grad = gi.clone(), gi.cuda()
    o_ref.step()
    o2.step()
    for i, (a, b) in enumerate(zip(pr, pg)):
        close(b, a, f"{mode} after reload, param {i}", rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("max_norm", [0.5, 1e6])
def test_fused_clip_grad_norm_matches_torch(M, max_norm):
    """optim.clip_grad_norm_ (movae_clip_grad_norm_multi: 3 launches for the whole list, no host sync) vs
    torch.nn.utils.clip_grad_norm_ (main.py:211-212): clipping and the no-op case, 70 tensors (two kernel-argument
    chunks), a channels_last gradient, a parameter without gradient."""
    from movae_amd.optim import clip_grad_norm_

    shapes = [(5,), (16, 8, 3, 3), (1,), (1000,)] + [(i + 2, 3) for i in range(66)]
    ref = [torch.nn.Parameter(rnd(*s, seed=20 + i)) for i, s in enumerate(shapes)]
    gpu = [torch.nn.Parameter(t.detach().clone().cuda()) for t in ref]
    for i, (a, b) in enumerate(zip(ref, gpu)):
        if i == 2:
            continue  # no gradient
        g = rnd(*shapes[i], seed=300 + i) * 0.3
        if i == 1:
            g = g.contiguous(memory_format=torch.channels_last)
        a.grad, b.grad = g.clone(), g.cuda()
    want = torch.nn.utils.clip_grad_norm_(ref, max_norm)
    got = clip_grad_norm_(gpu, max_norm)
    np.testing.assert_allclose(got.item(), want.item(), rtol=1e-6)
    for i, (a, b) in enumerate(zip(ref, gpu)):
        if i != 2:
            close(b.grad, a.grad, f"clipped grad {i}", rtol=1e-6, atol=1e-7)
    assert gpu[2].grad is None


@pytest.mark.parametrize("shape", [(2, 3, 7, 5), (3, 3, 16, 16), (1, 3, 1, 9), (2, 1, 4, 4)])
def test_sobel_edge_losses(M, shape):
    """csrc/edge.hip vs the oracle's restatement of models/gg_vae.py:125-156 (depthwise F.conv2d Sobel, max over channels,
    batch-max normalisation, smooth-L1 on gradient magnitudes), forward and backward, ragged shapes."""
    ops, _ = M
    from oracle import nets as ON

    n, c, h, w = shape
    x, r = rnd(n, c, h, w, seed=31) * 0.5, rnd(n, c, h, w, seed=32) * 0.5
    if c != 3:
        pytest.skip("the reference's Sobel buffers are built for 3 channels (gg_vae.py:52-53)")
    for name, fn, ofn in (("edge_weighted", ops.edge_weighted_pixel_loss, ON.edge_weighted_pixel_loss),
                          ("edge_matching", ops.edge_matching_loss, ON.edge_matching_loss)):
        rr = r.clone().requires_grad_(True)
        want = 1.7 * ofn(x, rr)
        want.backward()
        rg = nhwc(r)
        got = fn(rg, x.permute(0, 2, 3, 1).contiguous().cuda(), 1.7)
        got.backward()
        np.testing.assert_allclose(got.item(), want.item(), rtol=2e-5, atol=1e-7, err_msg=name)
        close(back(rg.grad), rr.grad, name + " grad", rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("mode", ["mag", "signed_mse", "maxnorm", "angle", "masked", "cosine"])
@pytest.mark.parametrize("case", ["rand", "far", "flat"])
def test_edge_matching_variants(M, mode, case):
    """Every `enum movae_edge_match` variant vs the reference's own numbers (tests/golden/edge_variants.npz: GGVQVAE /
    GGVAE methods on one (inputs, recons) pair) and vs the oracle on a ragged shape; "far" leaves the quadratic zone of the
    smooth-L1, "flat" has exactly-zero Sobel responses (the max-normalised variant's tie handling, the masked variant)."""
    ops, _ = M
    from oracle import nets as ON

    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "edge_variants.npz"))
    key = f"{case}.{mode}"
    if key + ".loss" not in fx.files:
        pytest.skip("atan2 / clamp branches at exactly-zero responses give NaN / 1e20-scale gradients in the reference")
    x, r = torch.from_numpy(fx["x"]), torch.from_numpy(fx[f"{case}.recons"])
    rg = nhwc(r)
    got = ops.edge_matching_loss(rg, x.permute(0, 2, 3, 1).contiguous().cuda(), 1.0, mode)
    got.backward()
    # the angle variant's smooth-L1 argument sits near +-pi: a last-bit atan2 difference moves single elements only
    np.testing.assert_allclose(got.item(), fx[key + ".loss"], rtol=3e-5, atol=1e-7, err_msg=key)
    close(back(rg.grad), torch.from_numpy(fx[key + ".grad"]), key + " grad", rtol=5e-4, atol=3e-6)
    if "gg_vae." + mode in {k.split(".", 1)[1].rsplit(".", 1)[0] for k in fx.files if k.startswith(case + ".gg_vae.")}:
        np.testing.assert_allclose(got.item(), fx[f"{case}.gg_vae.{mode}.loss"], rtol=3e-5, atol=1e-7)
    # ragged shape, non-unit scale, against the oracle
    x2, r2 = rnd(2, 3, 7, 5, seed=41) * 0.5, rnd(2, 3, 7, 5, seed=42) * 0.5
    rr = r2.clone().requires_grad_(True)
    want = 0.6 * ON.edge_matching_variant(x2, rr, mode)
    want.backward()
    rg2 = nhwc(r2)
    got2 = ops.edge_matching_loss(rg2, x2.permute(0, 2, 3, 1).contiguous().cuda(), 0.6, mode)
    got2.backward()
    np.testing.assert_allclose(got2.item(), want.item(), rtol=3e-5, atol=1e-7, err_msg=mode)
    close(back(rg2.grad), rr.grad, mode + " grad (ragged)", rtol=5e-4, atol=3e-6)


PAIR_SHAPES = [  # (transposed, n, hi, wi, ci, ho, wo, co, k, stride, pad)
    (False, 64, 16, 16, 32, 8, 8, 64, 3, 2, 1),   # conv dgrad = BWD gather <128,32> ... pairs with wgrad<64,64>
    (False, 64, 8, 8, 64, 4, 4, 128, 3, 2, 1),    # BWD <64,64> + wgrad <64,64>, per-class split-K on both sides
    (False, 32, 2, 2, 256, 1, 1, 512, 3, 2, 1),
    (True, 64, 4, 4, 128, 8, 8, 64, 3, 2, 1),     # transposed: dgrad = FWD gather <64,64>
    (True, 32, 16, 16, 32, 32, 32, 32, 3, 2, 1),  # FWD <128,32> + wgrad <32,128>
    (False, 8, 32, 32, 3, 16, 16, 32, 3, 2, 1),   # 3-channel end: thin kernels, nothing pairs (the pending dgrad is flushed)
    (False, 16, 1, 1, 128, 1, 1, 64, 1, 1, 0),    # linear layer
    (False, 5, 7, 9, 8, 4, 5, 12, 3, 2, 1),       # ragged, partial tiles
    (False, 16, 8, 8, 64, 8, 8, 64, 3, 1, 1),     # stride 1: a single parity class on the BWD-gather side
    (False, 8, 16, 16, 32, 8, 8, 64, 4, 2, 1),    # 4x4 taps, stride 2 (the VQ-VAE encoders)
    (True, 8, 8, 8, 64, 16, 16, 32, 4, 2, 1),     # ... and their transposed twins
    (False, 4, 8, 8, 128, 8, 8, 32, 1, 1, 0),     # 1x1 conv on a feature map (residual stacks)
    (True, 3, 5, 6, 16, 10, 12, 8, 3, 2, 1),      # transposed, ragged, output_padding 1
]


@pytest.mark.parametrize("groups", [1, 2])
@pytest.mark.parametrize("shape", PAIR_SHAPES)
def test_paired_dgrad_wgrad_is_bit_identical(M, shape, groups):
    """movae_conv[T]2d_dgrad_wgrad_grouped (one igemm2_pair launch when both sides land on the small-tile MFMA kernels) vs
    the two separate entry points on the same operands: the same tile bodies run with the same split-K plan, so dx, dW and
    the bias gradients must match bit for bit -- also for shapes where nothing pairs and for G = 2 cotangent groups."""
    import ctypes as C

    import movae_amd._lib as L

    tr, n, hi, wi, ci, ho, wo, co, k, stride, pad = shape
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(7)
    dy = torch.randn(groups, n, ho, wo, co, generator=g).to(dev)
    x = torch.randn(n, hi, wi, ci, generator=g).to(dev)
    w = (torch.randn(ci, k, k, co, generator=g) if tr else torch.randn(co, k, k, ci, generator=g)).to(dev) * 0.1
    ws = L.workspace(dev)
    st = torch.cuda.current_stream().cuda_stream
    pre = "movae_convT2d_" if tr else "movae_conv2d_"
    geom = (n, hi, wi, ci, ho, wo, co, k, k, stride, pad)
    arr = C.c_void_p * groups

    def outputs():
        return (torch.full((groups, n, hi, wi, ci), float("nan"), device=dev), [torch.full_like(w, float("nan")) for _ in range(groups)],
                [torch.full((co,), float("nan"), device=dev) for _ in range(groups)])

    dx1, dw1, db1 = outputs()
    L.call(pre + "dgrad", dy.data_ptr(), w.data_ptr(), dx1.data_ptr(), groups * n, *geom[1:], ws.data_ptr(), ws.numel(), st)
    L.call(pre + "wgrad_grouped", groups, dy.data_ptr(), x.data_ptr(), arr(*[t.data_ptr() for t in dw1]), arr(*[t.data_ptr() for t in db1]),
           *geom, 0, ws.data_ptr(), ws.numel(), st)
    dx2, dw2, db2 = outputs()
    L.call(pre + "dgrad_wgrad_grouped", groups, dy.data_ptr(), w.data_ptr(), x.data_ptr(), dx2.data_ptr(),
           arr(*[t.data_ptr() for t in dw2]), arr(*[t.data_ptr() for t in db2]), *geom, 0, ws.data_ptr(), ws.numel(), st)
    torch.cuda.synchronize()
    assert torch.isfinite(dx1).all() and torch.equal(dx1, dx2)
    for a, b in zip(dw1 + db1, dw2 + db2):
        assert torch.isfinite(a).all() and torch.equal(a, b)


def test_to_nhwc_remembers_only_unmodified_constant_inputs(M):
    """ops.to_nhwc hands out the previous conversion of the same NCHW tensor only while that tensor is unmodified, needs no
    gradient and no capture boundary lies in between (the step converts its input batch twice)."""
    ops, _ = M
    x = torch.rand(4, 3, 8, 8).cuda()
    a = ops.to_nhwc(x)
    assert ops.to_nhwc(x) is a                      # same object, same version: remembered
    x.mul_(2.0)                                     # in-place change bumps the version counter
    b = ops.to_nhwc(x)
    assert b is not a and torch.equal(b, x.permute(0, 2, 3, 1))
    y = x.clone()
    assert ops.to_nhwc(y) is not b                  # another tensor
    xr = torch.rand(4, 3, 8, 8).cuda().requires_grad_(True)
    c1, c2 = ops.to_nhwc(xr), ops.to_nhwc(xr)       # differentiable inputs always get their own autograd node
    assert c1 is not c2
    ops.forget_nhwc()
    assert ops.to_nhwc(y) is not None
    v = torch.rand(4, 8, 8, 3).cuda().permute(0, 3, 1, 2)  # already NHWC memory: zero-copy view
    assert ops.to_nhwc(v).data_ptr() == v.data_ptr()


def test_restack_rebuilds_stacked_views_only(M):
    """autojac._restack: G equal, adjacent, dense slices of one buffer (contiguous, or a permutation of it such as the NCHW
    view of an NHWC feature) come back as the [G, ...] view of that buffer (no copy); anything else stays a list."""
    from movae_amd import autojac

    buf = torch.arange(2 * 3 * 4, dtype=torch.float32).cuda().reshape(2, 3, 4)
    r = autojac._restack([buf[0].view(12), buf[1].view(12)])
    assert isinstance(r, torch.Tensor) and r.shape == (2, 12) and r.data_ptr() == buf.data_ptr() and torch.equal(r, buf.view(2, 12))
    assert isinstance(autojac._restack([buf[1], buf[0]]), list)                  # wrong order
    assert isinstance(autojac._restack([buf[0], buf[0].clone()]), list)         # different storage
    assert isinstance(autojac._restack([buf[0, :2], buf[1, :2]]), list)          # not adjacent
    r = autojac._restack([buf[0].t(), buf[1].t()])                               # dense, permuted: the stacked view keeps the strides
    assert isinstance(r, torch.Tensor) and r.data_ptr() == buf.data_ptr() and torch.equal(r, torch.stack([buf[0].t(), buf[1].t()]))
    assert isinstance(autojac._restack([buf[0, :, ::2], buf[1, :, ::2]]), list)  # not dense


def test_invalid_arguments_raise(M):
    ops, agg = M
    with pytest.raises(RuntimeError):
        ops.conv2d(torch.zeros(1, 4, 4, 3), torch.zeros(8, 3, 3, 3))  # CPU tensors: no CPU path
    with pytest.raises(ValueError):
        agg.compute_gramian(torch.zeros(9, 10).cuda())  # K > MOVAE_MAX_K


@pytest.mark.parametrize("m,k,n", [(256, 512, 128), (32, 256, 128), (7, 64, 20)])
def test_linear_pair_matches_two_linears(M, m, k, n):
    """ops.linear_pair (fc_mu || fc_var in one launch, movae_linear_pair_*): forward, plain backward and the batched pull-back with
    2 and 5 cotangent groups (the entry point takes four at a time) against two torch linears on the CPU."""
    ops, _ = M
    x, w1, b1, w2, b2 = rnd(m, k, seed=1), rnd(n, k, seed=2) * 0.1, rnd(n, seed=3), rnd(n, k, seed=4) * 0.1, rnd(n, seed=5)
    ref = [t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    r1, r2 = torch.nn.functional.linear(ref[0], ref[1], ref[2]), torch.nn.functional.linear(ref[0], ref[3], ref[4])
    g1, g2 = rnd(m, n, seed=6), rnd(m, n, seed=7)
    (r1 * g1).sum().backward(retain_graph=True)
    (r2 * g2).sum().backward()
    dev = [t.cuda().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    assert ops.linear_pair_ok(*dev) == (n % 4 == 0 and k % 4 == 0)
    y1, y2 = ops.linear_pair(*dev)
    close(y1, r1, "y1")
    close(y2, r2, "y2")
    ((y1 * g1.cuda()).sum() + (y2 * g2.cuda()).sum()).backward()
    for got, want, nm in zip(dev, ref, ("dx", "dw1", "db1", "dw2", "db2")):
        close(got.grad, want.grad, nm, rtol=2e-3, atol=2e-4)
    if not ops.linear_pair_ok(*dev):
        return
    # batched pull-back: G cotangent groups at once, one of the two outputs without a cotangent in the second case
    keep = ops.linear_pair(*dev)  # (a fresh tape: the backward above released the first one's saved tensors)
    fn = keep[0].grad_fn
    for G, with2 in ((2, True), (5, True), (3, False)):
        c1 = torch.stack([rnd(m, n, seed=10 + g) for g in range(G)])
        c2 = torch.stack([rnd(m, n, seed=30 + g) for g in range(G)]) if with2 else None
        with torch.no_grad():
            dx, dw1, db1, dw2, db2 = fn._forward_cls.backward_batched(fn, G, c1.cuda(), c2.cuda() if with2 else None)
        for g in range(G):
            want_dx = c1[g] @ w1 + (c2[g] @ w2 if with2 else 0)
            close(dx[g], want_dx, f"G={G} dx[{g}]", rtol=2e-3, atol=2e-4)
            close(dw1[g], c1[g].t() @ x, f"G={G} dw1[{g}]", rtol=2e-3, atol=2e-4)
            close(db1[g], c1[g].sum(0), f"G={G} db1[{g}]", rtol=2e-3, atol=2e-4)
            close(dw2[g], (c2[g].t() @ x) if with2 else torch.zeros(n, k), f"G={G} dw2[{g}]", rtol=2e-3, atol=2e-4)
            close(db2[g], c2[g].sum(0) if with2 else torch.zeros(n), f"G={G} db2[{g}]", rtol=2e-3, atol=2e-4)


def _philox4x32_10(c, k):
    """Philox4x32-10 (Salmon et al., SC'11) on numpy uint32: c [n, 4] counters, k (k0, k1) -> [n, 4] words."""
    c = [c[:, i].astype(np.uint64) for i in range(4)]
    k0, k1 = np.uint64(k[0]), np.uint64(k[1])
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c[0], np.uint64(0xCD9E8D57) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & m32, p1 & m32, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & m32, p0 & m32]
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & m32, (k1 + np.uint64(0xBB67AE85)) & m32
    return np.stack(c, 1)


def test_reparameterize_rng_philox_stream_moments_and_backward(M):
    """ops.reparameterize_rng: the noise is the documented stream (Philox4x32-10 keyed by the seed, counter block = (quad, draw),
    Box-Muller on the four words -- restated here in numpy), standard normal in its moments, fresh on every call (the launch
    advances the draw counter itself), and the op's values / gradients are those of ops.reparameterize fed the same eps."""
    ops, _ = M
    dev = torch.device("cuda")
    n_rows, d = 4096, 250  # (not a multiple of 4 per row: the quads run over the flat array; the tail quad is partial)
    mu, lv = rnd(n_rows, d, seed=1).to(dev).requires_grad_(True), (rnd(n_rows, d, seed=2) * 0.3).to(dev).requires_grad_(True)
    seed = 0x1234_5678_9ABC_DEF
    state = torch.tensor([seed, 7], dtype=torch.int64, device=dev)
    z = ops.reparameterize_rng(mu, lv, state)
    assert state.cpu().tolist() == [seed, 8], "the launch advances the draw counter"
    eps = ((z - mu) / torch.exp(0.5 * lv)).detach().cpu().double().numpy().reshape(-1)
    # the documented stream
    n = eps.size
    q = np.arange((n + 3) // 4, dtype=np.uint64)
    ctr = np.stack([q & np.uint64(0xFFFFFFFF), q >> np.uint64(32), np.full_like(q, 7), np.zeros_like(q)], 1)
    w = _philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32)).astype(np.float64)
    u = (np.floor(w / 256.0) + 0.5) / 16777216.0
    r0, t0, r1, t1 = np.sqrt(-2 * np.log(u[:, 0])), 2 * np.pi * u[:, 1], np.sqrt(-2 * np.log(u[:, 2])), 2 * np.pi * u[:, 3]
    want = np.stack([r0 * np.cos(t0), r0 * np.sin(t0), r1 * np.cos(t1), r1 * np.sin(t1)], 1).reshape(-1)[:n]
    np.testing.assert_allclose(eps, want, rtol=0, atol=2e-4)  # (fp32 log / sincos and the division above)
    # moments of a standard normal (1 M samples: the standard errors are 1e-3, 1.4e-3, 5e-3)
    assert abs(eps.mean()) < 5e-3 and abs(eps.var() - 1.0) < 7e-3 and abs(((eps - eps.mean()) ** 4).mean() / eps.var() ** 2 - 3.0) < 0.03
    # a second call draws different noise, the same state the same noise
    z2 = ops.reparameterize_rng(mu, lv, state)
    assert not torch.equal(z2, z)
    state.copy_(torch.tensor([seed, 7], dtype=torch.int64))
    assert torch.equal(ops.reparameterize_rng(mu, lv, state), z)
    # gradients: those of the plain op with this eps
    g = rnd(n_rows, d, seed=3).to(dev)
    dmu, dlv = torch.autograd.grad(z, [mu, lv], g)
    eps_t = ((z - mu) / torch.exp(0.5 * lv)).detach()
    zr = ops.reparameterize(mu, lv, eps_t)
    rmu, rlv = torch.autograd.grad(zr, [mu, lv], g)
    assert torch.equal(dmu, rmu)
    np.testing.assert_allclose(dlv.cpu().numpy(), rlv.cpu().numpy(), rtol=1e-4, atol=1e-6)  # (eps_t is eps up to the rounding of the division above)


@pytest.mark.parametrize("k", [2, 3, 5])
@pytest.mark.parametrize("norm", ["trace", "min_l2", "cosine"])
def test_gram_upgrad_in_two_launches_equals_the_three(M, k, norm):
    """movae_gram_upgrad (the solver kernel folds the Gramian's block partials itself) against movae_gram + movae_weights_upgrad_norm:
    the same G and the same weights, bit for bit; the aggregator's `weighting` takes the fused form only without hooks on the inner
    weighting and reports the same values to hooks on itself."""
    ops, agg = M
    J = rnd(k, 300_001, seed=20 + k).cuda()
    J[1] = 0.3 * J[0] + 0.7 * J[1]  # correlated rows: an active constraint in the dual-cone projection
    w8 = agg.UPGradWeighting(None, 1e-4, 1e-4, norm=norm)
    G_ref = agg.compute_gramian(J)
    w_ref = w8(G_ref)
    G, w = w8.from_jacobian(J)
    assert torch.equal(G, G_ref) and torch.equal(w, w_ref)
    a = agg.UPGrad()
    seen = []
    a.weighting.register_forward_hook(lambda mod, inp, out: seen.append(out.clone()))
    g1 = a(J)
    h = a.gramian_weighting.register_forward_hook(lambda mod, inp, out: None)  # a hook on the inner weighting: the separate calls
    g2 = a(J)
    h.remove()
    assert torch.equal(g1, g2) and len(seen) == 2 and torch.equal(seen[0], seen[1])


_VQ_ORDER_CASES = [(20000, 64, 16, True), (8192 + 37, 512, 64, False), (3000, 1500, 8, False)]  # rows, K, D, skewed usage


def _vq_codebook_grads(ops):
    out = []
    for i, (rows, K, D, skew) in enumerate(_VQ_ORDER_CASES):
        g = torch.Generator().manual_seed(40 + i)
        E = torch.randn(K, D, generator=g) * 2.0
        pick = (torch.tensor([3, 3, 3, 41, 3, 17])[torch.randint(0, 6, (rows,), generator=g)] if skew
                else torch.randint(0, K, (rows,), generator=g))
        x = (E[pick] + 0.01 * torch.randn(rows, D, generator=g)).reshape(1, rows, 1, D)
        xg, Eg = x.cuda().requires_grad_(True), E.cuda().requires_grad_(True)
        q, c, e, idx, used = ops.vector_quantize(xg, Eg)
        (0.25 * c + 2.0 * e).backward()
        out.append(Eg.grad.cpu().numpy().copy())
    return out


def test_vq_counting_placement_equals_the_radix_sort(M, tmp_path):
    """The codebook gradient's (code, row) order comes from a three-launch counting placement (vq.hip: vq_hist_k / vq_scan_k /
    vq_place_k); MOVAE_VQ_RADIX=1 (read once per process: a child process here) keeps the radix sort it replaced.  Same order, so the
    segmented sums are the same bits: skewed and uniform code usage, rows not a multiple of the 256-row blocks, more codes than one
    scan round (K > 1024)."""
    import subprocess
    import sys

    ops, _ = M
    here = _vq_codebook_grads(ops)
    code = f"""
import os, sys, numpy as np
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r})
import movae_amd
from movae_amd import ops
import test_hip_ops as T
np.savez({str(tmp_path / "radix.npz")!r}, *T._vq_codebook_grads(ops))
"""
    env = dict(os.environ, MOVAE_VQ_RADIX="1", MOVAE_NO_REBUILD="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    ref = np.load(tmp_path / "radix.npz")
    for i, a in enumerate(here):
        assert np.array_equal(a, ref[f"arr_{i}"]), f"case {_VQ_ORDER_CASES[i]}: placement and radix sort give different codebook gradients"
