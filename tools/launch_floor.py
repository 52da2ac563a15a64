#!/usr/bin/env python3
"""Development tool: per-launch floor of dependent kernels inside a replayed hipGraph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import movae_amd  # noqa: F401
import movae_amd._lib as L

lib = L.load()
dev = torch.device("cuda:0")


def timed(build, reps=5):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        build(side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        build(torch.cuda.current_stream().cuda_stream)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for n in (1024, 65536, 1 << 20, 1 << 22, 1 << 24):
    x = torch.randn(n, device=dev)
    y = torch.empty_like(x)
    N = 200

    def chain(st):
        for i in range(N):
            a, b = (x, y) if i % 2 == 0 else (y, x)
            lib.movae_act_fwd(a.data_ptr(), b.data_ptr(), n, 1, 0.01, st)

    us = timed(chain)
    print(f"act_fwd n={n:9d}: {us / N:7.2f} us per dependent launch  ({2 * 4 * n / (us / N) / 1e6:7.1f} GB/s)")
