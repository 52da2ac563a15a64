"""Opt-in bf16 operands (include/movae.h: movae_set_compute_dtype; `--dtype bf16`): the 128x128 implicit-GEMM kernels round their
two operands to bf16 on the way into LDS and multiply on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  NOT the parity path
(the reference is fp32 end to end, BASELINE.json only names bf16 for configs[1]); this file states the mode's OWN tolerance:

  * op level (forward, input gradient, weight gradient of one C3 layer) against the plain PyTorch fp32 op on the CPU:
    relative L2 error <= 1e-2 -- an operand rounded to 8 significant bits carries 2^-9 relative error, a product of two 2^-8,
    and the errors of a K-term sum add like its terms (measured 2-4e-3);
  * a C3-shaped VQ-VAE step (`--arch vq_vae`, 64x64, K=512, D=64, Aligned-MTL) against the fp32 ORACLE on the same inputs:
    every loss within 2e-2 relative, every parameter's aggregated gradient within 3e-2 relative L2 of the oracle's
    (measured: worst parameter 7.6e-3);
  * the fp32 mode is bit-for-bit unaffected by having visited the bf16 mode.
GPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture()
def bf16(gpu_device):
    import movae_amd  # noqa: F401
    from movae_amd import _lib as L

    prev = L.set_compute_dtype("bf16")
    yield L
    L.set_compute_dtype(prev)


def _rel(got, want):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    return float((got - want).norm() / want.norm().clamp_min(1e-30))


def _conv_pair(gpu_device, transposed=False):
    from movae_amd import ops

    g = torch.Generator().manual_seed(3)
    n, c, hw = 64, 256, 16
    x = torch.randn(n, c, hw, hw, generator=g)
    w = torch.randn(c, c, 3, 3, generator=g) * (1.0 / (c * 9)) ** 0.5
    b = torch.randn(c, generator=g) * 0.1
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, b, stride=1, padding=1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xg = x.permute(0, 2, 3, 1).contiguous().to(gpu_device).requires_grad_(True)
    wg = w.to(gpu_device).contiguous(memory_format=torch.channels_last).requires_grad_(True)

    def run():
        xg.grad = wg.grad = None
        y = ops.conv2d(xg, wg, b.to(gpu_device), 1, 1, None, 0.01)
        y.backward(gy.permute(0, 2, 3, 1).contiguous().to(gpu_device))
        return y.detach().clone(), xg.grad.detach().clone(), wg.grad.detach().clone()

    return run, (yr.permute(0, 2, 3, 1), xr.grad.permute(0, 2, 3, 1), wr.grad)


def test_bf16_conv_passes_within_stated_tolerance(gpu_device):
    import movae_amd  # noqa: F401
    from movae_amd import _lib as L

    run, (yr, dxr, dwr) = _conv_pair(gpu_device)
    y32, dx32, dw32 = run()
    k32 = L.load().movae_bench_last_kernel().decode()
    prev = L.set_compute_dtype("bf16")
    try:
        yb, dxb, dwb = run()
        kb = L.load().movae_bench_last_kernel().decode()
    finally:
        L.set_compute_dtype(prev)
    y32b, dx32b, dw32b = run()
    assert ",true>" in kb and ",true>" not in k32, (k32, kb)  # the bf16 instantiations really ran (and only when asked)
    for name, got, want, f32 in (("y", yb, yr, y32), ("dx", dxb, dxr, dx32), ("dw", dwb, dwr, dw32)):
        e32, eb = _rel(f32, want), _rel(got, want)
        print(f"[bf16 conv 256->256 3x3 @16x16] {name}: rel-L2 fp32 kernels {e32:.2e}, bf16 operands {eb:.2e}")
        assert e32 < 1e-5, (name, e32)
        assert 1e-4 < eb < 1e-2, (name, eb)  # (the lower bound: the bf16 path must actually have rounded something)
    assert torch.equal(y32, y32b) and torch.equal(dx32, dx32b) and torch.equal(dw32, dw32b), "fp32 mode changed by a visit to bf16"


def test_bf16_c3_shaped_step_against_the_fp32_oracle(bf16, gpu_device):
    from movae_amd import aggregation, autojac
    from movae_amd.models import get_network
    from oracle import nets
    from oracle.step import OracleTrainer

    class Args:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    B, size = 64, 64
    kw = dict(embedding_dim=64, num_embeddings=512, hidden_dims=[128, 256], num_residual_layers=2)
    a = Args(arch="vq_vae", batch_size=B, dataset_size=162770, recons_objective="mse", recons_activation=None, loss_weights=None,
             aggregator="aligned_mtl", agg_norm_eps=1e-4, agg_reg_eps=1e-4, mgda_epsilon=1e-5, mgda_max_iters=250, pref_weights=None, **kw)
    torch.manual_seed(42)
    net = get_network(size, 3, a, gpu_device).to(gpu_device).train()
    tr = OracleTrainer(nets.make_cfg("vq_vae", size, B, 162770, **kw), seed=42, agg="aligned_mtl")
    for n, p in net.named_parameters():
        assert torch.equal(p.detach().cpu(), tr.params[n].detach()), n
    x = torch.rand(B, 3, size, size, generator=torch.Generator().manual_seed(43))
    _, old, ograds, _ = tr.grads(x)
    xg = x.to(gpu_device)
    out = net(xg)
    ld = net.loss_function(xg, args=out)
    assert list(ld.keys()) == list(old.keys())
    for k, v in ld.items():
        np.testing.assert_allclose(v.detach().item(), float(old[k].detach()), rtol=2e-2, atol=1e-6, err_msg=f"loss {k} (bf16 operands vs fp32 oracle)")
    A = aggregation.make_aggregator(a)
    net.zero_grad(set_to_none=True)
    autojac.mtl_backward(losses=[v for k, v in ld.items() if k != "total_loss"], features=[out[f] for f in net.features], aggregator=A,
                         retain_graph=True)
    torch.cuda.synchronize()
    worst = ("", 0.0)
    for n, p in net.named_parameters():
        want = ograds[n]
        if float(want.abs().max()) < 1e-9:
            continue
        e = _rel(p.grad, want)
        worst = max(worst, (n, e), key=lambda t: t[1])
        assert e < 3e-2, f"{n}: rel-L2 {e:.2e} (bf16 operands vs fp32 oracle)"
    print(f"[bf16 C3-shaped vq_vae step, B={B}] losses within 2e-2; worst per-parameter gradient rel-L2 {worst[1]:.2e} ({worst[0]})")
