#!/usr/bin/env python3
"""Benchmark of the MO-VAE training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config C2]

One "step" = one pass of the hot path over one synthetic batch resident in HBM: forward, loss
terms, K per-loss backward passes + Jacobian, device-side Gram / weight solve / combine (or plain
backward for `sum`), [all-reduce of the aggregated gradient when N > 1], optimizer step.
Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for how `roofline` is obtained.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

CONFIGS = {
    # BASELINE.json:configs -- shapes per SURVEY.md section 8 (C1..C5); batch is per GPU (weak scaling)
    "C1": dict(arch="vae", agg="sum", batch=128, size=32, dataset_size=50000, latent_dim=128, hidden_dims=[32, 64, 128, 256, 512],
               flops_per_img=78.6e6, label="CIFAR-10 32x32 --arch vae --agg sum bs=128"),
    "C2": dict(arch="vae", agg="upgrad", batch=256, size=32, dataset_size=50000, latent_dim=128, hidden_dims=[32, 64, 128, 256, 512],
               flops_per_img=98.9e6, label="CIFAR-10 32x32 --arch vae --agg upgrad bs=256"),
    "C3": dict(arch="vq_vae", agg="aligned_mtl", batch=128, size=64, dataset_size=162770, embedding_dim=64, num_embeddings=512,
               hidden_dims=[128, 256], num_residual_layers=2, flops_per_img=11.9e9,
               label="CelebA 64x64 --arch vq_vae K=512 D=64 --agg aligned_mtl bs=128"),
    "C4": dict(arch="vq_vae2", agg="mgda_ln", batch=8, size=256, dataset_size=30000, embedding_dim=64, num_embeddings=512,
               hidden_dims=[128, 256], num_residual_layers=2, flops_per_img=37.2e9,
               label="CelebA-HQ 256x256 --arch vq_vae2 --agg mgda_ln bs=8/GPU"),
    "C5": dict(arch="betatc_vae", agg="upgrad", batch=32, size=256, dataset_size=1281167, latent_dim=128,
               hidden_dims=[32, 64, 128, 256, 512], anneal_steps=200, flops_per_img=13.35e9,
               label="ImageNet 256x256 --arch betatc_vae --agg upgrad bs=32/GPU"),
}

CONV_CALLS = {"movae_conv2d_fwd", "movae_conv2d_dgrad", "movae_conv2d_wgrad", "movae_convT2d_fwd", "movae_convT2d_dgrad",
              "movae_convT2d_wgrad", "movae_conv2d_wgrad_grouped", "movae_convT2d_wgrad_grouped",
              "movae_conv2d_dgrad_wgrad_grouped", "movae_convT2d_dgrad_wgrad_grouped"}
PAIR_CALLS = {"movae_linear_pair_fwd", "movae_linear_pair_bwd"}  # fc_mu || fc_var (conv family: 1x1 convs on a 1x1 image)
POOL_BATCHES = 24               # distinct synthetic batches cycled by the timed loop (>= 20)
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA, dense (--dtype bf16 only)
HBM_PEAK_GBS = 8000.0


class Args:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _base(name):
    """movae_conv2d_fwd_f -> movae_conv2d_fwd: the *_f entry points (BatchNorm fused into the conv, include/movae.h) take the
    plain argument list plus one trailing movae_fuse_t pointer."""
    return name[:-2] if name.endswith("_f") else name


def _geom_offset(name, a):
    """(index of `n` in the C-ABI argument tuple, cotangent groups) -- the grouped wgrads carry `groups` first."""
    name = _base(name)
    if "dgrad_wgrad" in name:  # (groups, dy, w, x, dx, dw[], dbias[], n, ...)
        return 7, int(a[0])
    if name.endswith("_grouped"):
        return 5, int(a[0])
    return (4 if name.endswith("_fwd") or name.endswith("_wgrad") else 3), 1


def conv_call_flops(name, a):
    """Nominal algorithmic FLOPs of one conv-family C-ABI call (2 * MACs, padding taps counted the way
    SURVEY 8d counts them): conv = out_pixels*k*k*ci*co, transposed conv = in_pixels*k*k*ci*co."""
    off, groups = _geom_offset(name, a)
    n, hi, wi, ci, ho, wo, co, kh, kw = a[off: off + 9]
    pix = hi * wi if "convT" in name else ho * wo
    both = 2.0 if "dgrad_wgrad" in name else 1.0  # the paired call computes the input AND the weight gradient
    return both * 2.0 * groups * n * pix * kh * kw * ci * co


def _valid_taps_1d(n_from, n_to, k, stride, pad):
    """#(position, tap) pairs of one axis whose partner position p * stride - pad + tap lies inside [0, n_to)."""
    return sum(1 for p in range(n_from) for t in range(k) if 0 <= p * stride - pad + t < n_to)


def conv_call_flops_executed(name, a):
    """FLOPs of the multiply-adds that meet real data: taps that only ever multiply zero padding (or fall off the output of a
    transposed conv) are not counted.  The nominal count above includes them; on the deep CIFAR layers they are a
    large share (2x2 -> 1x1, k3 s2 p1: 4 of 9 taps execute)."""
    off, groups = _geom_offset(name, a)
    n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad = a[off: off + 11]
    if "convT" in name:  # every input pixel scatters to hi * s - pad + tap
        pairs = _valid_taps_1d(hi, ho, kh, stride, pad) * _valid_taps_1d(wi, wo, kw, stride, pad)
    else:                # every output pixel gathers from ho * s - pad + tap
        pairs = _valid_taps_1d(ho, hi, kh, stride, pad) * _valid_taps_1d(wo, wi, kw, stride, pad)
    both = 2.0 if "dgrad_wgrad" in name else 1.0
    return both * 2.0 * groups * n * pairs * ci * co


def conv_call_bytes(name, a):
    """Algorithmic bytes of one conv-family call: each operand and the result touched once (fp32)."""
    off, groups = _geom_offset(name, a)
    n, hi, wi, ci, ho, wo, co, kh, kw = a[off: off + 9]
    if "dgrad_wgrad" in name:  # dy[G], x, w read; dx[G], dW[G] written
        return 4.0 * (n * hi * wi * ci * (1 + groups) + groups * n * ho * wo * co + (1 + groups) * kh * kw * ci * co)
    return 4.0 * (n * hi * wi * ci + groups * (n * ho * wo * co + kh * kw * ci * co))


def pair_call_info(name, a):
    """(flops, algorithmic bytes, shape) of a movae_linear_pair_* call (include/movae.h)."""
    if name.endswith("_fwd"):   # (x, w1, b1, w2, b2, y1, y2, m, n, k, stream)
        m, n, k = a[7:10]
        return 2 * 2.0 * m * n * k, 4.0 * (m * k + 2 * n * k + 2 * m * n), [m, 1, 1, k, 1, 1, 2 * n, 1, 1, 1, 0]
    g, (m, n, k) = int(a[0]), a[11:14]  # (groups, dy1, dy2, w1, w2, x, dx, dw1, dw2, db1, db2, m, n, k, stream)
    passes = (1 if a[6] else 0) + (1 if a[7] else 0)
    return g * 2 * passes * 2.0 * m * n * k, 4.0 * (m * k * (1 + g) + 2 * g * m * n + 2 * (1 + g) * n * k), [m, 1, 1, k, 1, 1, 2 * n, 1, 1, 1, 0, g]


def conv_call_key(name, a):
    off, groups = _geom_offset(name, a)
    return (name,) + tuple(a[off: off + 11]) + ((groups,) if groups > 1 else ())


def build_workload(cfg, device, seed=0, capturable=False, pool=None):
    import movae_amd  # noqa: F401
    from movae_amd import aggregation
    from movae_amd.models import get_network
    from movae_amd.train import make_optimizer

    a = Args(arch=cfg["arch"], batch_size=cfg["batch"], dataset_size=cfg["dataset_size"], recons_objective="mse",
             recons_activation=None, loss_weights=None, aggregator=cfg["agg"], agg_norm_eps=1e-4, agg_reg_eps=1e-4,
             mgda_epsilon=1e-5, mgda_max_iters=250, pref_weights=None, optimizer="adam", lr=1e-3, wd=0, momentum=0.9,
             max_grad_norm=None, **{k: cfg[k] for k in ("latent_dim", "hidden_dims", "embedding_dim", "num_embeddings",
                                                        "num_residual_layers", "anneal_steps") if k in cfg})
    torch.manual_seed(seed)
    net = get_network(cfg["size"], 3, a, device).to(device).train()
    opt = make_optimizer(net, a, capturable=capturable)
    agg = aggregation.make_aggregator(a)
    g = torch.Generator().manual_seed(seed + 1)
    pool = [torch.rand(cfg["batch"], 3, cfg["size"], cfg["size"], generator=g).to(device)
            for _ in range(POOL_BATCHES if pool is None else pool)]
    return net, opt, agg, a, pool


def _fp64_oracle(tr):
    """The oracle of `tr` in float64 with the same initial values and its own Adam (the yardstick of elbo_check)."""
    from collections import OrderedDict

    from oracle.step import OracleTrainer

    t64 = OracleTrainer(tr.cfg, seed=0, agg=tr.agg)
    with torch.no_grad():
        for k, v in tr.sd.items():
            t64.sd[k] = v.detach().double().requires_grad_(v.requires_grad) if v.is_floating_point() else v.detach().clone()
    t64.params = OrderedDict((n, t64.sd[n]) for n in tr.params)
    t64.opt = torch.optim.Adam(list(t64.params.values()), lr=1e-3)
    return t64


def cpu_baseline(cfg, seconds, device=None, check_steps=20):
    """The CPU oracle ("port" of the reference step, oracle/step.py) timed on this box's host cores; with `device`, also the
    ELBO check: a fresh HIP model and the oracle take the same `check_steps` optimisation steps (same init, same batches, same
    CPU-drawn eps) and the loss dicts of the last step are reported side by side."""
    from oracle import nets
    from oracle.step import OracleTrainer

    # a 1-GPU box's CPU share is 16 cores; PyTorch's default of one thread per visible core (128) oversubscribes
    # these small convolutions and is ~10x slower than 16 threads
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    kw = {k: cfg[k] for k in ("latent_dim", "hidden_dims", "embedding_dim", "num_embeddings", "num_residual_layers", "anneal_steps")
          if k in cfg}
    ocfg = nets.make_cfg(cfg["arch"], cfg["size"], cfg["batch"], cfg["dataset_size"], **kw)
    tr = OracleTrainer(ocfg, seed=0, agg=cfg["agg"])
    g = torch.Generator().manual_seed(1)
    need_eps = nets.ARCHS[cfg["arch"]]["needs_eps"]
    nb = 4
    xs = [torch.rand(cfg["batch"], 3, cfg["size"], cfg["size"], generator=g) for _ in range(nb)]
    es = [torch.randn(cfg["batch"], cfg.get("latent_dim", 1), generator=g) if need_eps else None for _ in range(nb)]
    check = None
    if device is not None and check_steps > 0:
        from movae_amd.models.betatc_vae import BetaTCVAE
        from movae_amd.train import train_step

        BetaTCVAE.num_iter = 0
        net, opt, agg, a, _ = build_workload(cfg, device, seed=0, pool=0)
        hip = ora = ora64 = None
        # the yardstick: the same oracle in float64, stepped alongside.  A fp32 trajectory is chaotic in its small components
        # (the weighted KL term): what two correct fp32 runs may differ by is what the fp32 oracle differs from float64 by.
        t64 = _fp64_oracle(tr) if cfg["arch"] in ("vae", "betatc_vae") else None  # (VQ: float64 quantises near-tied rows differently)
        for i in range(check_steps):
            if need_eps:
                net.eps_override = es[i % nb].to(device)
            ld, _ = train_step(net, xs[i % nb].to(device), opt, agg, a)
            hip = {k: v.detach().item() for k, v in ld.items()}
            ora = tr.step(xs[i % nb], es[i % nb])
            if t64 is not None:
                ora64 = t64.step(xs[i % nb].double(), es[i % nb].double() if es[i % nb] is not None else None)
        rel = {k: abs(hip[k] - ora[k]) / max(abs(ora[k]), 1e-12) for k in ora}
        check = dict(steps=check_steps, hip_losses=hip, oracle_losses=ora, max_rel_diff=max(rel.values()), rel_diff=rel,
                     note="same init / batches / eps on both sides; eager HIP step vs oracle/step.py")
        if ora64 is not None:
            own = {k: abs(ora[k] - ora64[k]) / max(abs(ora64[k]), 1e-12) for k in ora}
            lim = {k: max(1e-3, 4.0 * own[k]) for k in ora}
            check.update(oracle_fp64_losses=ora64, oracle_fp32_vs_fp64_rel=own, limit_rel=lim,
                         within_limit=all(rel[k] <= lim[k] for k in ora),
                         yardstick="limit = max(1e-3, 4 x |fp32 oracle - fp64 oracle|) per component, relative "
                                   "(tests/test_hip_parity_full.py::trajectory_limits)")
        del net, opt
        BetaTCVAE.num_iter = 0
        tr = OracleTrainer(ocfg, seed=0, agg=cfg["agg"])
    tr.step(xs[0], es[0])  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        tr.step(xs[n % nb], es[n % nb])
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= 2000:
            break
    out = dict(value=cfg["batch"] * n / dt, unit="images/sec", cores=torch.get_num_threads(), kind="port",
               sample=f"{n} steps of {cfg['label']} (oracle/step.py, PyTorch-CPU fp32, {dt:.1f} s)")
    if check is not None:
        out["elbo_check"] = check
    return out


FAMILY = [("loss", "movae_combine_losses"), ("loss", "movae_vae_losses"), ("conv", "movae_conv"), ("conv", "movae_linear_pair"), ("batchnorm", "movae_bn_"), ("batchnorm", "movae_scale_shift"), ("loss", "movae_recon"), ("loss", "movae_kl"), ("loss", "movae_tc"),
          ("vq", "movae_vq"), ("aggregation", "movae_gram"), ("aggregation", "movae_weights"), ("aggregation", "movae_combine"),
          ("aggregation", "movae_gd_"), ("optimizer", "movae_adam"), ("optimizer", "movae_sumsq"), ("optimizer", "movae_scale_by"),
          ("elementwise", "movae_")]


def family_of(name):
    return next(f for f, pre in FAMILY if name.startswith(pre))


def measure_dominant_kernel(recorded, device, reps=20, live=False):
    """HIP-event timing of the launches captured from one real step: every captured call is re-issued `reps`
    times back to back on the launch stream between two events.
    live=False (eager step): conv-family calls only, on freshly synthesized operands of the recorded geometry.
    live=True  (calls recorded while the step was captured into a hipGraph, whose pool keeps every operand
    alive): every C-ABI call of the step, on the step's own operands.
    Returns (conv rows, {family: us per step} for the non-conv calls)."""
    import movae_amd._lib as L

    lib = L.load()
    rows = []
    ws = L.workspace(device)

    def synth(name, a):
        """Fresh, owned operand buffers of the recorded call's geometry (the step's own activations are
        recycled by the caching allocator, so their recorded addresses must not be reused)."""
        off, groups = _geom_offset(name, a)
        n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad = a[off: off + 11]
        x = torch.randn(n * hi * wi * ci, device=device)
        y = torch.randn(groups * n * ho * wo * co, device=device)
        w = torch.randn(co * kh * kw * ci, device=device) * 0.05
        b = torch.randn(co, device=device)
        geom = (n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad)
        tail = (ws.data_ptr(), ws.numel())
        if name.endswith("_grouped"):
            import ctypes
            ws_ = [torch.randn(co * kh * kw * ci, device=device) for _ in range(groups)]
            arr = (ctypes.c_void_p * groups)(*[t.data_ptr() for t in ws_])
            if "dgrad_wgrad" in name:
                dxs = torch.empty(groups * n * hi * wi * ci, device=device)
                args = (groups, y.data_ptr(), w.data_ptr(), x.data_ptr(), dxs.data_ptr(), arr, None) + geom + (0,) + tail
                return args, (x, y, w, dxs, ws_, arr)
            args = (groups, y.data_ptr(), x.data_ptr(), arr, None) + geom + (0,) + tail
            return args, (x, y, ws_, arr)
        if name.endswith("_fwd"):
            args = (x.data_ptr(), w.data_ptr(), b.data_ptr() if a[2] else 0, y.data_ptr()) + geom + (a[15], a[16]) + tail
        elif name.endswith("_dgrad"):
            args = (y.data_ptr(), w.data_ptr(), x.data_ptr()) + geom + tail
        else:
            args = (y.data_ptr(), x.data_ptr(), w.data_ptr(), b.data_ptr() if a[3] else 0) + geom + (0,) + tail
        return args, (x, y, w, b)

    def timed(fn, a):
        # the launches are captured into a hipGraph first so that the timed interval contains device work
        # only (a Python/ctypes launch costs more host time than these kernels run for); the events are recorded
        # on the stream the graph is launched on
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        si = len(a) - 2 if getattr(fn, "__name__", "").endswith("_f") else len(a) - 1  # *_f: (..., stream, fuse)

        def with_stream(sp):
            return a[:si] + (sp,) + a[si + 1:]

        with torch.cuda.stream(side):
            a_side = with_stream(side.cuda_stream)
            for _ in range(2):
                fn(*a_side)
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            a_cap = with_stream(torch.cuda.current_stream(device).cuda_stream)
            for _ in range(reps):
                fn(*a_cap)
        graph.replay()
        stream = torch.cuda.current_stream(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        graph.replay()
        e1.record(stream)
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps

    other = {}
    for name, rec in recorded:
        fn = getattr(lib, name)
        if name in PAIR_CALLS and live:
            us = timed(fn, tuple(rec))
            fl, by, shape = pair_call_info(name, rec)
            rows.append(dict(call=name, kernel=lib.movae_bench_last_kernel().decode(), shape=shape, us=us, us_main=us, gflop=fl / 1e9,
                             gflop_executed=fl / 1e9, alg_bytes=by))
            continue
        if _base(name) not in CONV_CALLS:
            if live:
                fam = family_of(name)
                us = timed(fn, tuple(rec))
                other[fam] = other.get(fam, 0.0) + us
                rows.append(dict(call=name, kernel=fam, shape=[x for x in rec if isinstance(x, int) and 0 < x < (1 << 31)][:4],
                                 us=us, us_main=us, gflop=0.0))
            continue
        if live:
            a = tuple(rec)
        else:
            a0, keep = synth(name, rec)
            a = a0 + (0,)
        us = timed(fn, a)                      # the whole call: main kernel + split-K reduce / bias column-sum
        kernel = lib.movae_bench_last_kernel().decode()
        lib.movae_bench_main_kernel_only(1)
        try:
            us_main = timed(fn, a)             # the main kernel alone (what rocprofv3 lists under `kernel`)
        finally:
            lib.movae_bench_main_kernel_only(0)
        rows.append(dict(call=name, kernel=kernel, shape=list(conv_call_key(name, a)[1:]), us=us, us_main=us_main,
                         gflop=conv_call_flops(name, a) / 1e9, gflop_executed=conv_call_flops_executed(name, a) / 1e9,
                         alg_bytes=conv_call_bytes(name, a)))
    return rows, other


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=str, default="C2", choices=sorted(CONFIGS))
    ap.add_argument("--agg", type=str, default=None)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps steps each (median reported)")
    ap.add_argument("--min-gpu-seconds", type=float, default=6.0, help="keep repeating the timed block for at least this long")
    ap.add_argument("--elbo-check-steps", type=int, default=20, help="HIP vs oracle loss check inside the cpu_baseline leg (0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the step as one captured hipGraph (auto = on)")
    ap.add_argument("--kernel-table", type=str, default=None, help="write the per-launch conv table (JSON) here")
    ap.add_argument("--dtype", choices=["fp32", "bf16"], default="fp32",
                    help="fp32 = the reference's arithmetic (default, the headline).  bf16 (opt-in, never the default): bf16 operands / "
                         "fp32 accumulate in the 128x128 implicit-GEMM kernels; the line's `dtype` then says bf16")
    args = ap.parse_args()

    cfg = dict(CONFIGS[args.config])
    if args.agg:
        cfg["agg"] = args.agg
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        print(f"bench.py --gpus {args.gpus} must be launched with torch.distributed.run --nproc-per-node {args.gpus}", file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    device = torch.device(f"cuda:{local_rank % torch.cuda.device_count()}")
    torch.cuda.set_device(device)

    import movae_amd  # noqa: F401
    import movae_amd._lib as L
    from movae_amd.parallel import DataParallelGrads
    from movae_amd.train import train_step

    from movae_amd.train import GraphedTrainStep

    L.set_compute_dtype(args.dtype)  # fp32 unless --dtype bf16 (opt-in; include/movae.h: movae_set_compute_dtype)
    dp = DataParallelGrads.from_env() if (world > 1 or os.environ.get("MOVAE_FORCE_DP")) else None
    use_graph = args.graph != "off"  # every hot-path model is capturable (no host reads inside the step)
    # the CPU leg runs FIRST (rank 0, N = 1): the GPU then stays busy from here to the end of the run, which is what a coarse
    # utilisation sampler around the process can see
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg, args.cpu_seconds, device=device, check_steps=args.elbo_check_steps)
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    net, opt, agg, a, pool = build_workload(cfg, device, capturable=use_graph)
    if dp is not None:
        dp.attach(net)
    # the roofline block (like the CPU baseline) belongs to the N = 1 line; the scaling runs report throughput only
    want_roofline = rank == 0 and world == 1 and dp is None and not args.no_roofline
    graphed = GraphedTrainStep(net, opt, agg, a, pool[0], dp=dp, record_calls=want_roofline) if use_graph else None
    if graphed is not None and dp is not None:
        # watchdog: if graph replay + collective misbehaves on this node (observed when several ranks share one
        # GPU under gloo), every rank falls back to the eager step together
        torch.cuda.synchronize()
        dp.barrier()
        t_probe = time.perf_counter()
        for i in range(4):
            graphed.step(pool[i % len(pool)])
        torch.cuda.synchronize()
        probe = torch.tensor([(time.perf_counter() - t_probe) / 4], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(probe, op=torch.distributed.ReduceOp.MAX)
        if float(probe.item()) > 0.05:  # a healthy replay step is ~3 ms
            graphed, use_graph = None, False

    def step(i, eager=False):
        if graphed is not None and not eager:
            return graphed.step(pool[i % len(pool)])
        return train_step(net, pool[i % len(pool)], opt, agg, a, dp)

    for i in range(args.warmup):
        step(i)

    def timed_block(first):
        """EXACTLY --steps steps between barrier + device synchronisation on both sides; max over ranks."""
        if dp is not None:
            dp.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ld = None
        for i in range(args.steps):
            ld, _ = step(first + i)
        torch.cuda.synchronize()
        if dp is not None:
            dp.barrier()
        dt = time.perf_counter() - t0
        if dp is not None:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt, ld

    # the timed block is repeated (>= --repeats times, and until the GPU has been timed for --min-gpu-seconds): the reported
    # value is the MEDIAN block, min / max are given beside it
    times = []
    first, ld = args.warmup, None
    t_begin = time.perf_counter()
    while True:
        dt, ld = timed_block(first)
        first += args.steps
        times.append(dt)
        enough = len(times) >= args.repeats and (time.perf_counter() - t_begin) >= args.min_gpu_seconds
        if dp is not None:  # all ranks must leave the loop together
            flag = torch.tensor([1 if enough else 0], device=device, dtype=torch.int32)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            enough = bool(flag.item())
        if enough or len(times) >= 5000:
            break
    ts = sorted(times)
    elapsed = ts[len(ts) // 2]
    final_loss = float(ld["total_loss"].item())

    roofline = None
    issued_gflop = issued_gflop_x = None
    if want_roofline:
        live = graphed is not None and len(graphed.calls) > 0
        if live:
            recorded = graphed.calls  # operands owned by the captured graph's pool: re-issued in place
        else:
            recorded = []
            L.TRACE = lambda name, cargs: recorded.append((name, cargs))
            keep = step(0, eager=True)  # noqa: F841
            L.TRACE = None
        torch.cuda.synchronize()
        all_rows, other = measure_dominant_kernel(recorded, device, live=live)
        rows = [r for r in all_rows if _base(r["call"]) in CONV_CALLS or r["call"] in PAIR_CALLS]
        tot_us = sum(r["us"] for r in rows)
        tot_gf = sum(r["gflop"] for r in rows)
        issued_gflop, issued_gflop_x = tot_gf, sum(r.get("gflop_executed", 0.0) for r in rows)
        # group the main-kernel timings by kernel symbol (the way rocprofv3 --stats does) and report the one with
        # the largest time per step; the whole conv family (all kernels + their epilogue launches) is given beside it
        groups = {}
        for r in rows:
            gk = groups.setdefault(r["kernel"], dict(us=0.0, gflop=0.0, gflop_x=0.0, n=0))
            gk["us"] += r["us_main"]
            gk["gflop"] += r["gflop"]
            gk["gflop_x"] += r["gflop_executed"]
            gk["n"] += 1
        mfma_groups = {k: v for k, v in groups.items() if k.startswith(("igemm", "kgemm", "kpair"))} or groups
        dom = max(mfma_groups, key=lambda k: mfma_groups[k]["us"])
        d = groups[dom]
        # memory-side bytes per launch of that kernel from the committed PMC passes (profiles/pmc_traffic_<cfg>.json)
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", f"pmc_traffic_{args.config}.json")
        if os.path.exists(tpath):
            kern = json.load(open(tpath)).get("kernels", {})
            parts = [kern.get(k.strip(), {}).get("hbm_bytes_per_launch") for k in dom.split(" + ")]  # (a call of two main launches: both)
            traffic = sum(parts) if all(v is not None for v in parts) else None
            if traffic is not None:
                traffic_source = f"profiles/pmc_traffic_{args.config}.json (rocprofv3 --pmc passes of tools/pmc_traffic.py, not this run)"
        alg_bytes = sum(r["alg_bytes"] for r in rows if r["kernel"] == dom) / d["n"]
        achieved = d["gflop"] * 1e9 / (d["us"] * 1e-6) / 1e12 if d["us"] > 0 else 0.0
        achieved_x = d["gflop_x"] * 1e9 / (d["us"] * 1e-6) / 1e12 if d["us"] > 0 else 0.0
        fam = tot_gf * 1e9 / (tot_us * 1e-6) / 1e12 if tot_us > 0 else 0.0
        bf_dom = ",true>" in dom and dom.startswith("igemm2_")  # the dominant launch multiplies on the bf16 pipe (--dtype bf16)
        peak = BF16_MFMA_PEAK_TFLOPS if bf_dom else FP32_MFMA_PEAK_TFLOPS
        roofline = dict(bound="mfma", kernel=dom + (" (implicit-GEMM conv, v_mfma_f32_32x32x16_bf16)" if bf_dom else " (implicit-GEMM conv, v_mfma_f32_32x32x2_f32)"),
                        achieved=round(achieved, 3), peak=peak, unit="TFLOP/s",
                        frac=round(achieved / peak, 4), traffic=traffic, traffic_source=traffic_source,
                        achieved_executed_taps=round(achieved_x, 3), frac_executed_taps=round(achieved_x / peak, 4),
                        flop_counting="achieved: nominal 2*MACs incl. padding taps (SURVEY 8d); *_executed_taps: only taps that meet data",
                        launches_per_step=d["n"], avg_launch_us=round(d["us"] / d["n"], 2),
                        flop_per_launch=round(d["gflop"] * 1e9 / d["n"]), algorithmic_bytes_per_launch=round(alg_bytes),
                        kernel_us_per_step=round(d["us"], 1),
                        per_kernel={k: dict(launches=v["n"], avg_us=round(v["us"] / v["n"], 2),
                                            tflops=round(v["gflop"] * 1e3 / v["us"], 2) if v["us"] > 0 else 0.0,
                                            tflops_executed_taps=round(v["gflop_x"] * 1e3 / v["us"], 2) if v["us"] > 0 else 0.0)
                                    for k, v in sorted(groups.items(), key=lambda kv: -kv[1]["us"])},
                        conv_family=dict(launches_per_step=len(rows), us_per_step=round(tot_us, 1), tflops=round(fam, 3),
                                         frac=round(fam / FP32_MFMA_PEAK_TFLOPS, 4)),
                        other_families_us_per_step={k: round(v, 1) for k, v in sorted(other.items())} if other else None)
        if args.kernel_table:
            with open(args.kernel_table, "w") as f:
                json.dump(all_rows, f, indent=1)
    if rank == 0:
        n_img = cfg["batch"] * world * args.steps
        out = {
            "metric": "training images/sec", "value": n_img / elapsed, "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": cfg["label"], "arch": cfg["arch"], "aggregator": cfg["agg"], "per_gpu_batch": cfg["batch"],
                       "global_batch": cfg["batch"] * world, "image": f"3x{cfg['size']}x{cfg['size']}",
                       "parallelism": f"dp{world}", "optimizer": "adam", "final_total_loss": final_loss,
                       "launch": ("hipGraph replay" + (f" ({graphed.dp_form})" if dp is not None else "")) if use_graph else "eager",
                       "timed_blocks": {"n": len(ts), "steps_each": args.steps, "ms_per_step_min": ts[0] / args.steps * 1e3,
                                        "ms_per_step_median": elapsed / args.steps * 1e3, "ms_per_step_max": ts[-1] / args.steps * 1e3},
                       "distinct_batches": len(pool),
                       "step_gflop": cfg["flops_per_img"] * cfg["batch"] / 1e9,
                       "step_gflop_note": ("step_gflop is SURVEY 8d's nominal formula ((1+2K) F_s + 3 F_t with K = every loss); "
                                           "step_gflop_issued sums the nominal FLOPs of the conv-family calls the step really makes "
                                           "(a loss with no path to the features costs no pull-back: K_eff < K for the VQ archs), "
                                           "step_gflop_issued_executed_taps counts only taps that meet data -- divide THOSE by ms_per_step"),
                       "step_gflop_issued": round(issued_gflop, 3) if issued_gflop is not None else None,
                       "step_gflop_issued_executed_taps": round(issued_gflop_x, 3) if issued_gflop_x is not None else None,
                       "parity": ("OPT-IN bf16 operands / fp32 accumulation in the 128x128 implicit-GEMM kernels only (everything else fp32); NOT "
                                  "the parity path: held to its own tolerance, tests/test_hip_bf16.py" if args.dtype == "bf16" else
                                  "fp32 end to end (the reference's arithmetic; bf16 operands are an opt-in, --dtype bf16, never this line); models / losses / "
                                  "MGDA / Aligned-MTL pinned by golden vectors of the reference, UPGrad + mtl_backward restate torchjd "
                                  "(absent: docstring KAT, unit-weights invariant, scipy cross-check); element-wise vs the oracle at "
                                  "this shape and the 1-epoch ELBO trajectory: tests/test_hip_parity_full.py")},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dp is not None:
        dp.shutdown()


if __name__ == "__main__":
    main()
