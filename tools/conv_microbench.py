#!/usr/bin/env python3
"""Per-shape timing of the conv-family C-ABI entry points (development tool).

    python tools/conv_microbench.py [--reps 20] [--only wgrad] [--shapes c2|c3]

Each call is captured `reps` times into a hipGraph and timed with events, so the number is device
time for the call (main kernel + split-K reduce + bias column-sum), comparable with
`rocprofv3 --kernel-trace --stats -- python tools/conv_microbench.py ...`.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import movae_amd  # noqa: E402,F401
import movae_amd._lib as L  # noqa: E402

# (kind, n, hi, wi, ci, ho, wo, co, k, s, p) kind: conv|convT ; geometry as passed to the C ABI
C2 = [
    ("conv", 256, 32, 32, 3, 16, 16, 32, 3, 2, 1), ("conv", 256, 16, 16, 32, 8, 8, 64, 3, 2, 1),
    ("conv", 256, 8, 8, 64, 4, 4, 128, 3, 2, 1), ("conv", 256, 4, 4, 128, 2, 2, 256, 3, 2, 1),
    ("conv", 256, 2, 2, 256, 1, 1, 512, 3, 2, 1), ("conv", 256, 1, 1, 512, 1, 1, 128, 1, 1, 0),
    ("conv", 256, 1, 1, 128, 1, 1, 512, 1, 1, 0),
    ("convT", 256, 1, 1, 512, 2, 2, 256, 3, 2, 1), ("convT", 256, 2, 2, 256, 4, 4, 128, 3, 2, 1),
    ("convT", 256, 4, 4, 128, 8, 8, 64, 3, 2, 1), ("convT", 256, 8, 8, 64, 16, 16, 32, 3, 2, 1),
    ("convT", 256, 16, 16, 32, 32, 32, 32, 3, 2, 1), ("conv", 256, 32, 32, 32, 32, 32, 3, 3, 1, 1),
]
C3 = [
    ("conv", 32, 64, 64, 3, 32, 32, 128, 4, 2, 1), ("conv", 32, 32, 32, 128, 16, 16, 256, 4, 2, 1),
    ("conv", 32, 16, 16, 256, 16, 16, 256, 3, 1, 1), ("conv", 32, 16, 16, 256, 16, 16, 256, 1, 1, 0),
    ("conv", 32, 16, 16, 256, 16, 16, 64, 1, 1, 0), ("conv", 32, 16, 16, 64, 16, 16, 256, 3, 1, 1),
    ("convT", 32, 16, 16, 256, 32, 32, 128, 4, 2, 1), ("convT", 32, 32, 32, 128, 64, 64, 3, 4, 2, 1),
]


# full C3 batch: the layers whose 128x128 tiles dominate C3-C5 (tools/gpu_pmc_igemm128.sh)
C3BIG = [("conv", 128, 16, 16, 256, 16, 16, 256, 3, 1, 1), ("conv", 128, 32, 32, 128, 16, 16, 256, 4, 2, 1),
         ("convT", 128, 16, 16, 256, 32, 32, 128, 4, 2, 1)]


# the 3-channel image ends of C5 (BetaTC-VAE 256x256, bs 32) and C3 (thin_* kernels)
THIN = [("conv", 32, 256, 256, 3, 128, 128, 32, 4, 2, 1), ("conv", 32, 256, 256, 32, 256, 256, 3, 3, 1, 1),
        ("conv", 128, 64, 64, 3, 32, 32, 128, 4, 2, 1), ("convT", 128, 32, 32, 128, 64, 64, 3, 4, 2, 1)]


def time_call(fn, args, reps):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn(*(args + (side.cuda_stream,)))
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(reps):
            fn(*(args + (s,)))
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--shapes", type=str, default="c2")
    ap.add_argument("--index", type=int, nargs="*", default=None)
    ap.add_argument("--sweep-split", type=int, nargs="*", default=None,
                    help="time every call with the split-K factor pinned to each value (0 = the library's heuristic)")
    ap.add_argument("--lib", type=str, default="", help="A/B: load this prebuilt libmovae_hip.so instead of the in-tree one")
    ap.add_argument("--kgemm", type=int, default=0, help="movae_bench_force_kgemm: 1 = the block-internal split-K kernels wherever they "
                    "can serve, -1 = never, 0 = the library's heuristic")
    ap.add_argument("--no-dbias", action="store_true", help="weight gradient without the bias gradient (a conv in front of a BatchNorm)")
    a = ap.parse_args()
    if a.lib:
        L.LIB_PATH = os.path.abspath(a.lib)
        os.environ["MOVAE_NO_REBUILD"] = "1"
    lib = L.load()
    lib.movae_bench_force_kgemm(a.kgemm)
    dev = torch.device("cuda:0")
    ws = L.workspace(dev)
    shapes = {"c2": C2, "c3": C3, "c3big": C3BIG, "thin": THIN}[a.shapes]
    tot = 0.0
    for i, (kind, n, hi, wi, ci, ho, wo, co, k, s, p) in enumerate(shapes):
        if a.index is not None and i not in a.index:
            continue
        x = torch.randn(n * hi * wi * ci, device=dev)
        y = torch.randn(n * ho * wo * co, device=dev)
        w = torch.randn(co * k * k * ci, device=dev) * 0.05
        b = torch.randn(co, device=dev)
        geom = (n, hi, wi, ci, ho, wo, co, k, k, s, p)
        tail = (ws.data_ptr(), ws.numel())
        pre = "movae_convT2d_" if kind == "convT" else "movae_conv2d_"
        calls = {
            "fwd": (getattr(lib, pre + "fwd"), (x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr()) + geom + (0, 0.01) + tail),
            "dgrad": (getattr(lib, pre + "dgrad"), (y.data_ptr(), w.data_ptr(), x.data_ptr()) + geom + tail),
            "wgrad": (getattr(lib, pre + "wgrad"), (y.data_ptr(), x.data_ptr(), w.data_ptr(), None if a.no_dbias else b.data_ptr()) + geom + (0,) + tail),
        }
        pix = hi * wi if kind == "convT" else ho * wo
        gf = 2.0 * n * pix * k * k * ci * co / 1e9
        line = f"{i:2d} {kind:5s} {n}x{hi}x{wi}x{ci}->{ho}x{wo}x{co} k{k}s{s} {gf:6.3f}GF |"
        for name, (fn, args) in calls.items():
            if a.only and a.only != name:
                continue
            if a.sweep_split:
                res = []
                for sp in a.sweep_split:
                    lib.movae_bench_force_split(sp)
                    res.append((sp, time_call(fn, args, a.reps)))
                lib.movae_bench_force_split(0)
                fn(*(args + (0,)))
                kern = lib.movae_bench_last_kernel().decode()
                best = min(res, key=lambda r: r[1])
                print(f"{line} {name:5s} {kern:22s} best S={best[0]:3d} {best[1]:6.1f}us | " +
                      " ".join(f"{sp}:{us:.1f}" for sp, us in res), flush=True)
                continue
            us = time_call(fn, args, a.reps)
            tot += us
            kern = lib.movae_bench_last_kernel().decode()
            line += f" {name} {us:7.1f}us {gf / us * 1e3:6.1f}TF/s {kern[:18]:18s}|"
        if not a.sweep_split:
            print(line, flush=True)
    print(f"sum {tot:.1f} us")


if __name__ == "__main__":
    main()
