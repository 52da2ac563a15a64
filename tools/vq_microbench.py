#!/usr/bin/env python3
"""Development tool: timing of the VQ lookup and its backward for uniform and skewed code usage.

    python tools/vq_microbench.py [--lib other/libmovae_hip.so]     (A/B against a prebuilt library)"""
import os
import sys

if "--lib" in sys.argv:  # before the package loads the in-tree library
    os.environ["MOVAE_NO_REBUILD"] = "1"

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import movae_amd  # noqa: E402,F401
import movae_amd._lib as L  # noqa: E402

if "--lib" in sys.argv:
    L.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from movae_amd import ops  # noqa: E402


def run(rows, K, D, used_codes, reps=10):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    E = torch.randn(K, D, generator=g) * 3
    pick = torch.randint(0, used_codes, (rows,), generator=g)
    x = (E[pick] + 0.01 * torch.randn(rows, D, generator=g)).reshape(1, rows, 1, D).to(dev).requires_grad_(True)
    Eg = E.to(dev).requires_grad_(True)
    for _ in range(2):
        q, c, e, idx, used = ops.vector_quantize(x, Eg)
        (q.sum() + c + e).backward()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        q, c, e, idx, used = ops.vector_quantize(x, Eg)
        (q.sum() + c + e).backward()
    e1.record()
    e1.synchronize()
    eager = e0.elapsed_time(e1) * 1e3 / reps
    # the lookup alone, replayed from a graph (device time)
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.vector_quantize(x.detach(), Eg.detach())
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_):
            for _ in range(reps):
                ops.vector_quantize(x.detach(), Eg.detach())
        g_.replay()
        e0.record()
        g_.replay()
        e1.record()
        e1.synchronize()
    print(f"rows {rows} K {K} D {D} used {used_codes}: {eager:8.1f} us per fwd+bwd (eager), lookup {e0.elapsed_time(e1) * 1e3 / reps:6.1f} us (graph)")


if __name__ == "__main__":
    run(32768, 512, 64, 512)
    run(32768, 512, 64, 20)
    run(8192, 512, 64, 512)
