"""Deferred weight-gradient reduces (include/movae.h: movae_reduce_defer; ops.deferred_reduces): a split-K
reduce whose destination is a gradient sink is parked and the next implicit-GEMM launch of the backward carries it as extra
blocks.  The arithmetic and its order are those of the stand-alone reduce kernels, so everything here is compared BIT FOR BIT
against the same computation with MOVAE_DEFER_REDUCE off: the op-level hand-off through every carrier kernel family (pair, tiled
input / weight gradient, kgemm), the flush paths (second armed call, a foreign call, the block's exit), and whole aggregated
steps of the BASELINE model families.  GPU only."""
import ctypes as C

import numpy as np
import pytest
import torch


pytestmark = pytest.mark.gpu


def _stats(lib, reset=False):
    out = (C.c_longlong * 3)()
    pending = lib.movae_reduce_defer_stats(out, 1 if reset else 0)
    return list(out), pending


@pytest.fixture()
def park_any_size(gpu_device):
    """the op-level tests park whatever the layer produces (the product only parks launch-bound reduces: movae_reduce_defer_max_bytes)"""
    import movae_amd  # noqa: F401
    from movae_amd import _lib as L

    lib = L.load()
    prev = lib.movae_reduce_defer_max_bytes(1 << 40)
    yield lib
    lib.movae_reduce_defer_max_bytes(prev)


def _rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _layer_chain(dev, sizes, n, hw, G):
    """conv layers (k3 s1 p1) with random operands: [(x, w, dy)] per layer, NHWC"""
    out = []
    for i, (ci, co) in enumerate(sizes):
        out.append((_rnd(n, hw, hw, ci, seed=10 + i).to(dev), (_rnd(co, ci, 3, 3, seed=20 + i) * 0.1).to(dev).contiguous(memory_format=torch.channels_last),
                    _rnd(G, n, hw, hw, co, seed=30 + i).to(dev)))
    return out


def _run_chain(lib, L, layers, G, defer, force_kgemm=0):
    """dgrad+wgrad (pair call) of every layer in turn into sink buffers; returns (dx list, dw list, stats)"""
    prev = lib.movae_bench_force_kgemm(force_kgemm)
    try:
        _stats(lib, reset=True)
        dev = layers[0][0].device
        ws0 = L.workspace(dev)
        st = L.stream_ptr(dev)
        res = []
        for x, w, dy in layers:
            n, h, wd, ci = x.shape
            co = w.shape[0]
            wm = w.permute(0, 2, 3, 1)
            assert wm.is_contiguous()
            dx = torch.empty((G,) + tuple(x.shape), device=dev)
            dws = [torch.full((co, 3, 3, ci), float("nan"), device=dev) for _ in range(G)]
            arr = (C.c_void_p * G)(*[t.data_ptr() for t in dws])
            if defer:
                wsp, wsb = L.defer_arm(dev)
            else:
                wsp, wsb = ws0.data_ptr(), ws0.numel()
            L.check(lib.movae_conv2d_dgrad_wgrad_grouped(G, dy.data_ptr(), wm.data_ptr(), x.data_ptr(), dx.data_ptr(), arr, None, n, h, wd, ci, h, wd,
                                                         co, 3, 3, 1, 1, 0, wsp, wsb, st), "pair")
            res.append((dx, dws, lib.movae_bench_last_kernel().decode()))
        stats_before_flush, pending = _stats(lib)
        L.check(lib.movae_reduce_flush(), "flush")
        torch.cuda.synchronize()
        return res, stats_before_flush, pending
    finally:
        lib.movae_bench_force_kgemm(prev)


@pytest.mark.parametrize("sizes,n,hw,G,kg", [
    ([(64, 128), (128, 64), (64, 128)], 8, 8, 2, 0),     # tiled kernels: igemm2_pair carries the previous layer's reduce
    ([(128, 256), (256, 128), (128, 256)], 16, 4, 2, 1),  # kgemm input gradient + tiled weight gradient: kpair_k carries it
    ([(32, 128), (128, 32), (32, 128)], 4, 16, 1, 0),     # 32-wide layers (VQ-VAE-2 residual blocks): unpaired bwd / wgrad kernels carry it
    ([(256, 256), (256, 256)], 2, 32, 1, 0),              # 128x128 tiles, many slabs
])
def test_parked_reduce_rides_on_the_next_launch_bit_exact(sizes, n, hw, G, kg, gpu_device, park_any_size):
    import movae_amd  # noqa: F401
    from movae_amd import _lib as L

    lib = L.load()
    layers = _layer_chain(gpu_device, sizes, n, hw, G)
    ref, s0, _ = _run_chain(lib, L, layers, G, defer=False, force_kgemm=kg)
    got, s1, pending = _run_chain(lib, L, layers, G, defer=True, force_kgemm=kg)
    assert s0 == [0, 0, 0]
    parked, carried, alone = s1
    if parked == 0:
        pytest.skip(f"no split-K reduce at these shapes ({[k for _, _, k in got]})")
    # every parked reduce was carried by a later layer's launch; only the last one may still wait for movae_reduce_flush
    assert carried + pending == parked and alone == 0, (s1, pending, [k for _, _, k in got])
    for (dx0, dw0, k0), (dx1, dw1, k1) in zip(ref, got):
        assert k0 == k1
        assert torch.equal(dx0, dx1), k0
        for a, b in zip(dw0, dw1):
            assert not torch.isnan(b).any() and torch.equal(a, b), k0


def test_flush_paths(gpu_device, park_any_size):
    """A second armed call while one is parked, a call outside _lib.DEFER_PASS, and the block's exit each launch the parked
    reduce stand-alone; an unarmed call never parks."""
    import movae_amd  # noqa: F401
    from movae_amd import _lib as L
    from movae_amd import ops

    lib = L.load()
    _stats(lib, reset=True)
    x = _rnd(8, 8, 8, 64, seed=1).to(gpu_device)
    w = (_rnd(128, 64, 3, 3, seed=2) * 0.1).to(gpu_device).contiguous(memory_format=torch.channels_last)
    dy = _rnd(8, 8, 8, 128, seed=3).to(gpu_device)
    wm = w.permute(0, 2, 3, 1)
    st = L.stream_ptr(gpu_device)

    def wgrad(armed):
        dw = torch.full((128, 3, 3, 64), float("nan"), device=gpu_device)
        if armed:
            wsp, wsb = L.defer_arm(gpu_device)
        else:
            ws0 = L.workspace(gpu_device)
            wsp, wsb = ws0.data_ptr(), ws0.numel()
        L.call("movae_conv2d_wgrad_grouped", 1, dy.data_ptr(), x.data_ptr(), (C.c_void_p * 1)(dw.data_ptr()), None, 8, 8, 8, 64, 8, 8, 128, 3, 3, 1,
               1, 0, wsp, wsb, st)
        return dw

    want = wgrad(False)
    torch.cuda.synchronize()
    assert _stats(lib)[0] == [0, 0, 0]
    with ops.deferred_reduces():
        a = wgrad(True)
        assert _stats(lib) == ([1, 0, 0], 1)
        b = wgrad(True)  # the same kernel family carries the first one (a weight-gradient launch is a carrier too)
        s, pending = _stats(lib)
        assert s[0] == 2 and s[1] + s[2] == 1 and pending == 1
        L.call("movae_add", want.data_ptr(), want.data_ptr(), torch.empty_like(want).data_ptr(), want.numel(), st)  # DEFER_PASS: stays parked
        assert _stats(lib)[1] == 1
        L.call("movae_sumsq", want.data_ptr(), want.numel(), torch.empty(1, device=gpu_device).data_ptr(), L.workspace(gpu_device).data_ptr(),
               L.workspace(gpu_device).numel(), st)  # not one of the backward's ops: flushes first
        s, pending = _stats(lib)
        assert pending == 0 and s[1] + s[2] == 2
        c = wgrad(True)
        assert _stats(lib)[1] == 1
    assert _stats(lib)[1] == 0  # the block's exit
    torch.cuda.synchronize()
    for t in (a, b, c):
        assert torch.equal(t, want)
    # outside the block nothing is armed by ops (and an armed call is disarmed by the entry point that consumed it)
    assert lib.movae_reduce_defer(0) == 0


@pytest.mark.parametrize("tag,batch", [("C2", 64), ("C3", 8), ("C4", 2), ("C5", 4)])
def test_aggregated_step_identical_with_and_without_deferral(tag, batch, gpu_device, monkeypatch, park_any_size):
    """One aggregated step (forward, per-loss backward, batched pull-back, UPGrad) of every BASELINE architecture -- the
    configuration's layer shapes at a reduced batch -- with and without deferral: the same gradients bit for bit, and the
    deferral did happen (reduces were parked, and carried by later launches)."""
    import movae_amd  # noqa: F401
    from movae_amd import _lib as L
    from movae_amd import aggregation, autojac, ops
    from movae_amd.models import get_network
    from movae_amd.models.betatc_vae import BetaTCVAE
    from conftest import cfg_from_meta
    from test_hip_models import Args, _full_case

    lib = L.load()
    fx, m = _full_case(tag)
    c = cfg_from_meta(m)
    size = int(m["input_size"])
    res = {}
    for on in (True, False):
        monkeypatch.setattr(ops, "DEFER_REDUCE", on)
        args = Args(arch=c["arch"], batch_size=batch, dataset_size=c["dataset_size"], recons_objective="mse", recons_activation=None,
                    loss_weights=None, **{k: v for k, v in c.items() if k in ("latent_dim", "hidden_dims", "embedding_dim",
                                                                             "num_embeddings", "num_residual_layers", "anneal_steps")})
        torch.manual_seed(3)
        BetaTCVAE.num_iter = 0
        net = get_network(size, num_channels=3, args=args, device=gpu_device).to(gpu_device).train()
        x = torch.rand(batch, 3, size, size, generator=torch.Generator().manual_seed(4)).to(gpu_device)
        if "latent_dim" in c:
            net.eps_override = torch.randn(batch, c["latent_dim"], generator=torch.Generator().manual_seed(5)).to(gpu_device)
        _stats(lib, reset=True)
        out = net(x)
        ld = net.loss_function(x, args=out)
        comp = [v for k, v in ld.items() if k != "total_loss"]
        autojac.mtl_backward(losses=comp, features=[out[f] for f in net.features], aggregator=aggregation.UPGrad())
        torch.cuda.synchronize()
        s, pending = _stats(lib)
        assert pending == 0
        if not on:
            assert s == [0, 0, 0]
        res[on] = ({n: (None if p.grad is None else p.grad.detach().cpu().numpy().copy()) for n, p in net.named_parameters()}, s)
        del net, out, ld, comp
    BetaTCVAE.num_iter = 0
    (g1, s1), (g0, _) = res[True], res[False]
    parked, carried, alone = s1
    nonfinite = {on: [n for n, g in res[on][0].items() if g is not None and not np.isfinite(g).all()] for on in (True, False)}
    assert not nonfinite[True] and not nonfinite[False], f"{tag}: non-finite gradients {nonfinite} (stats {s1})"
    assert parked >= 2 and carried >= 1 and carried + alone == parked, f"{tag}: deferral statistics {s1}"
    for n in g0:
        assert (g0[n] is None) == (g1[n] is None), n
        if g0[n] is not None:
            assert np.isfinite(g1[n]).all(), n
            assert np.array_equal(g0[n], g1[n]), f"{tag} {n}: deferral changed the gradient (stats {s1})"
