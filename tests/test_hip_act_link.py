"""The activation derivative of a conv + activation pair applied by the NEXT conv's input-gradient epilogue (ops.ActLink,
movae_fuse_t::ep_act_*): chains of nn.Stack modules against the plain PyTorch fp32 chain on the CPU, through every kernel form
that takes part (FWD / BWD gather, unsplit epilogue and split-K reduce, paired launch, nested Stacks), plus the fallback where the
consumer's kernel has no such epilogue.  GPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(got, want, what, rtol=1e-3, atol=1e-4):
    got, want = got.detach().cpu().float().numpy(), want.detach().cpu().float().numpy()
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol * max(1.0, float(np.abs(want).max())), err_msg=what)


# (name, batch, size, [(kind, cin, cout, k, stride, pad, out_pad, act)], nested)
CHAINS = [
    ("enc3", 8, 16, [("conv", 8, 16, 3, 2, 1, 0, "lrelu"), ("conv", 16, 32, 3, 2, 1, 0, "lrelu"), ("conv", 32, 64, 3, 1, 1, 0, None)], False),
    ("enc3_nested", 8, 16, [("conv", 8, 16, 3, 2, 1, 0, "lrelu"), ("conv", 16, 32, 3, 2, 1, 0, "lrelu"), ("conv", 32, 64, 3, 1, 1, 0, "relu")], True),
    ("dec3", 8, 4, [("convT", 32, 32, 3, 2, 1, 1, "lrelu"), ("convT", 32, 16, 3, 2, 1, 1, "relu"), ("conv", 16, 8, 3, 1, 1, 0, None)], True),
    # deep + tiny spatial extent: split-K on every pass (the reduce applies the derivative)
    ("deep_split", 4, 2, [("conv", 256, 256, 3, 1, 1, 0, "lrelu"), ("conv", 256, 128, 3, 1, 1, 0, "lrelu"), ("conv", 128, 64, 1, 1, 0, 0, None)], False),
    # big tiles, unsplit epilogue
    ("wide", 16, 16, [("conv", 64, 128, 3, 1, 1, 0, "relu"), ("conv", 128, 128, 3, 1, 1, 0, "lrelu"), ("convT", 128, 64, 4, 2, 1, 0, None)], False),
    # consumer without the epilogue (3-channel thin kernel): the producer runs its own activation backward
    ("thin_consumer", 4, 16, [("conv", 16, 16, 3, 1, 1, 0, "lrelu"), ("conv", 16, 3, 3, 1, 1, 0, None)], False),
    # ... except the 32-channel form of that kernel (last layers of the VAE / BetaTC-VAE decoders), whose output tile passes through LDS
    ("thin_consumer32", 4, 16, [("convT", 16, 32, 3, 2, 1, 1, "lrelu"), ("conv", 32, 3, 3, 1, 1, 0, None)], False),
    # Stacks held as separate attributes, run through nn.chain (BetaTC-VAE: decoder -> final_layer)
    ("chained", 4, 8, [("convT", 32, 32, 3, 2, 1, 1, "lrelu"), ("convT", 32, 32, 3, 2, 1, 1, "lrelu"), ("conv", 32, 3, 3, 1, 1, 0, None)], "chain"),
]


def _build(spec, nested):
    from movae_amd import nn as mnn

    mods, groups = [], []
    for kind, cin, cout, k, s, p, op, act in spec:
        conv = mnn.Conv2d(cin, cout, k, s, p) if kind == "conv" else mnn.ConvTranspose2d(cin, cout, k, s, p, op)
        layer = [conv] + ([mnn.LeakyReLU() if act == "lrelu" else mnn.ReLU()] if act else [])
        groups.append(layer)
        mods.extend(layer)
    if nested == "chain":  # first layer | the rest, joined by nn.chain
        a, b = mnn.Stack(mnn.Stack(*groups[0])), mnn.Stack(*[m for g in groups[1:] for m in g])

        class Chained(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.a, self.b = a, b

            def forward(self, x):
                return mnn.chain(x, self.a, self.b)

        return Chained(), [g[0] for g in groups]
    return (mnn.Stack(*[mnn.Stack(*g) for g in groups]) if nested else mnn.Stack(*mods)), [g[0] for g in groups]


@pytest.mark.parametrize("case", CHAINS, ids=[c[0] for c in CHAINS])
@pytest.mark.parametrize("fused", [True, False])
def test_conv_act_conv_chain(case, fused, gpu_device, monkeypatch):
    import movae_amd  # noqa: F401
    from movae_amd import _lib as L, ops

    name, B, size, spec, nested = case
    monkeypatch.setattr(ops, "FUSE_ACT", fused)
    calls = []
    monkeypatch.setattr(L, "TRACE", lambda nm, a: calls.append(nm))
    torch.manual_seed(7)
    stack, convs = _build(spec, nested)
    stack = stack.to(gpu_device)
    x = torch.randn(B, spec[0][1], size, size)
    # ---- PyTorch fp32 reference on the CPU ----
    xr = x.clone().requires_grad_(True)
    ws = [(c.weight.detach().cpu().contiguous().clone().requires_grad_(True), c.bias.detach().cpu().clone().requires_grad_(True)) for c in convs]
    h = xr
    for (kind, cin, cout, k, s, p, op, act), (w, b) in zip(spec, ws):
        h = F.conv2d(h, w, b, stride=s, padding=p) if kind == "conv" else F.conv_transpose2d(h, w, b, stride=s, padding=p, output_padding=op)
        if act:
            h = F.leaky_relu(h, 0.01) if act == "lrelu" else F.relu(h)
    cot = torch.randn(h.shape, generator=torch.Generator().manual_seed(3))
    (h * cot).sum().backward()
    # ---- HIP ----
    xh = x.to(gpu_device).requires_grad_(True)
    out = stack(ops.to_nhwc(xh)).permute(0, 3, 1, 2)
    _close(out, h, f"{name}: output", rtol=5e-4, atol=5e-5)
    calls.clear()
    (out * cot.to(gpu_device)).sum().backward()
    _close(xh.grad, xr.grad, f"{name}: dx")
    for i, (c, (w, b)) in enumerate(zip(convs, ws)):
        _close(c.weight.grad, w.grad, f"{name}: dW{i}")
        _close(c.bias.grad, b.grad, f"{name}: db{i}")
    n_act = sum(1 for sp in spec if sp[7])
    n_bwd = sum(1 for nm in calls if nm.startswith("movae_act_bwd"))
    if not fused:
        assert n_bwd == n_act, calls
    elif name == "thin_consumer":
        assert n_bwd == 1, calls  # the 3-channel consumer's kernel has no such epilogue: the producer's own pass ran
    else:
        last_has_act = spec[-1][7] is not None  # the chain's last activation has no consumer inside the Stack
        assert n_bwd == (1 if last_has_act else 0), calls


def test_two_readers_fail_loudly(gpu_device):
    """The link is only valid when the consumer is the only reader of the producer's output; nn.Stack guarantees that.  Used by hand
    with a second reader, the producer's backward must refuse rather than apply the derivative to a sum that already holds it."""
    import movae_amd  # noqa: F401
    from movae_amd import nn as mnn, ops

    torch.manual_seed(0)
    c1, c2 = mnn.Conv2d(8, 16, 3, 1, 1).to(gpu_device), mnn.Conv2d(16, 16, 3, 1, 1).to(gpu_device)
    x = torch.randn(2, 8, 8, 8, device=gpu_device)
    link = ops.ActLink()
    y1 = c1(x, "lrelu", False, ops.ConvFusion(act_out=link))
    y2 = c2(y1, None, False, ops.ConvFusion(act_in=link))
    with pytest.raises(RuntimeError, match="another reader"):
        (y2.sum() + y1.sum()).backward()


@pytest.mark.parametrize("fused", [True, False])
def test_vqvae2_resblock_standalone_relu(fused, gpu_device, monkeypatch):
    """models/vq_vae2.py ResBlock: ReLU -> Conv3x3 -> ReLU -> Conv1x1, plus the input (which therefore has two readers -- the link
    concerns the stand-alone ReLU's OUTPUT, read by the 3x3 conv only).  Two blocks in a row, against the PyTorch chain."""
    import movae_amd  # noqa: F401
    from movae_amd import _lib as L, nn as mnn, ops
    from movae_amd.models.vq_vae2 import ResBlock

    monkeypatch.setattr(ops, "FUSE_ACT", fused)
    calls = []
    monkeypatch.setattr(L, "TRACE", lambda nm, a: calls.append(nm))
    torch.manual_seed(5)
    blocks = mnn.Stack(ResBlock(32, 16), ResBlock(32, 16), mnn.ReLU()).to(gpu_device)
    x = torch.randn(4, 32, 8, 8)
    xr = x.clone().requires_grad_(True)
    h, ws = xr, []
    for rb in list(blocks)[:2]:
        c3, c1 = rb.conv[1], rb.conv[3]
        w = [t.detach().cpu().contiguous().clone().requires_grad_(True) for t in (c3.weight, c3.bias, c1.weight, c1.bias)]
        ws.append(w)
        h = h + F.conv2d(F.relu(F.conv2d(F.relu(h), w[0], w[1], padding=1)), w[2], w[3])
    h = F.relu(h)
    cot = torch.randn(h.shape, generator=torch.Generator().manual_seed(2))
    (h * cot).sum().backward()
    xh = x.to(gpu_device).requires_grad_(True)
    out = blocks(ops.to_nhwc(xh)).permute(0, 3, 1, 2)
    _close(out, h, "output", rtol=5e-4, atol=5e-5)
    calls.clear()
    (out * cot.to(gpu_device)).sum().backward()
    _close(xh.grad, xr.grad, "dx")
    for i, (rb, w) in enumerate(zip(list(blocks)[:2], ws)):
        for got, want, nm in ((rb.conv[1].weight.grad, w[0].grad, "dW3"), (rb.conv[1].bias.grad, w[1].grad, "db3"),
                              (rb.conv[3].weight.grad, w[2].grad, "dW1"), (rb.conv[3].bias.grad, w[3].grad, "db1")):
            _close(got, want, f"block {i} {nm}")
    n_bwd = sum(1 for nm in calls if nm.startswith("movae_act_bwd"))
    # unfused: 2 blocks x (stand-alone ReLU + epilogue ReLU) + the trailing ReLU; fused: only the trailing one (no conv reads it)
    assert n_bwd == (1 if fused else 5), calls
    # the identity branch's cotangent: added by the 3x3 conv's input-gradient epilogue (no launch), else one explicit add per block
    assert calls.count("movae_add") == (0 if fused else 2), calls


@pytest.mark.parametrize("fused", [True, False])
def test_vqvae_residual_layer(fused, gpu_device, monkeypatch):
    """models/vq_vae.py ResidualLayer: x + Conv1x1(ReLU(Conv3x3(x))), bias-free -- the branch starts with a conv, whose input-gradient
    epilogue adds the identity cotangent (no activation in front).  (The grouped forms of the batched pull-back run through the
    same blocks in tests/test_hip_parity_full.py: C3 Aligned-MTL and C4 MGDA at the config shapes.)"""
    groups = 1
    import movae_amd  # noqa: F401
    from movae_amd import _lib as L, nn as mnn, ops
    from movae_amd.models.vq_vae import ResidualLayer

    monkeypatch.setattr(ops, "FUSE_ACT", fused)
    calls = []
    monkeypatch.setattr(L, "TRACE", lambda nm, a: calls.append(nm))
    torch.manual_seed(9)
    head = mnn.Conv2d(8, 64, 3, 1, 1).to(gpu_device)
    layers = mnn.Stack(ResidualLayer(64, 64), ResidualLayer(64, 64), mnn.LeakyReLU()).to(gpu_device)
    x = torch.randn(4, 8, 8, 8)
    xr = x.clone().requires_grad_(True)
    hw = [t.detach().cpu().contiguous().clone().requires_grad_(True) for t in (head.weight, head.bias)]
    h = F.conv2d(xr, hw[0], hw[1], padding=1)
    ws = []
    for rl in list(layers)[:2]:
        w3 = rl.resblock[0].weight.detach().cpu().contiguous().clone().requires_grad_(True)
        w1 = rl.resblock[2].weight.detach().cpu().contiguous().clone().requires_grad_(True)
        ws.append((w3, w1))
        h = h + F.conv2d(F.relu(F.conv2d(h, w3, padding=1)), w1)
    h = F.leaky_relu(h, 0.01)
    gen = torch.Generator().manual_seed(4)
    cots = [torch.randn(h.shape, generator=gen) for _ in range(groups)]
    ref = [torch.autograd.grad((h * c).sum(), [xr, hw[0]] + [t for pair in ws for t in pair], retain_graph=True) for c in cots]
    xh = x.to(gpu_device).requires_grad_(True)
    out = layers(head(ops.to_nhwc(xh))).permute(0, 3, 1, 2)
    _close(out, h, "output", rtol=5e-4, atol=5e-5)
    params = [xh, head.weight] + [t for rl in list(layers)[:2] for t in (rl.resblock[0].weight, rl.resblock[2].weight)]
    calls.clear()
    got = [torch.autograd.grad((out * cots[0].to(gpu_device)).sum(), params)]
    for g in range(groups):
        for a, b, nm in zip(got[g], ref[g], ["dx", "dW_head", "dW3_0", "dW1_0", "dW3_1", "dW1_1"]):
            _close(a, b, f"group {g} {nm}")
    assert calls.count("movae_add") == (0 if fused else 2), calls


@pytest.mark.parametrize("fused", [True, False])
def test_standalone_activation_before_a_nested_stack(fused, gpu_device, monkeypatch):
    """models/vq_vae.py:218-236: residual layers -> LeakyReLU -> Sequential(Conv / ConvTranspose, LeakyReLU).  The stand-alone
    activation's only reader is the first conv of the NEXT Stack: that conv's input gradient applies its derivative."""
    import movae_amd  # noqa: F401
    from movae_amd import _lib as L, nn as mnn, ops

    monkeypatch.setattr(ops, "FUSE_ACT", fused)
    calls = []
    monkeypatch.setattr(L, "TRACE", lambda nm, a: calls.append(nm))
    torch.manual_seed(11)
    c0, c1, c2 = mnn.Conv2d(8, 32, 3, 1, 1), mnn.ConvTranspose2d(32, 16, 4, 2, 1), mnn.Conv2d(16, 8, 1, 1, 0)
    stack = mnn.Stack(mnn.Stack(c0), mnn.LeakyReLU(), mnn.Stack(c1, mnn.LeakyReLU()), mnn.Stack(c2, mnn.LeakyReLU())).to(gpu_device)
    x = torch.randn(4, 8, 8, 8)
    xr = x.clone().requires_grad_(True)
    ws = [(c.weight.detach().cpu().contiguous().clone().requires_grad_(True), c.bias.detach().cpu().clone().requires_grad_(True)) for c in (c0, c1, c2)]
    h = F.leaky_relu(F.conv2d(xr, ws[0][0], ws[0][1], padding=1), 0.01)
    h = F.leaky_relu(F.conv_transpose2d(h, ws[1][0], ws[1][1], stride=2, padding=1), 0.01)
    h = F.leaky_relu(F.conv2d(h, ws[2][0], ws[2][1]), 0.01)
    cot = torch.randn(h.shape, generator=torch.Generator().manual_seed(5))
    (h * cot).sum().backward()
    xh = x.to(gpu_device).requires_grad_(True)
    out = stack(ops.to_nhwc(xh)).permute(0, 3, 1, 2)
    _close(out, h, "output", rtol=5e-4, atol=5e-5)
    calls.clear()
    (out * cot.to(gpu_device)).sum().backward()
    _close(xh.grad, xr.grad, "dx")
    for i, (c, (w, b)) in enumerate(zip((c0, c1, c2), ws)):
        _close(c.weight.grad, w.grad, f"dW{i}")
        _close(c.bias.grad, b.grad, f"db{i}")
    n_bwd = sum(1 for nm in calls if nm.startswith("movae_act_bwd"))
    assert n_bwd == (1 if fused else 3), calls  # fused: only the last activation (no consumer) runs a backward pass of its own
