#!/usr/bin/env python3
"""Per-shape timing of movae_bn_act_fwd / movae_bn_act_bwd(_grouped) (development tool)."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import movae_amd  # noqa: E402,F401
import movae_amd._lib as L  # noqa: E402
from conv_microbench import time_call  # noqa: E402

SHAPES = [(262144, 32), (65536, 32), (16384, 64), (4096, 128), (1024, 256), (256, 512),
          (131072, 128), (524288, 128), (32768, 256)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    lib = L.load()
    dev = torch.device("cuda:0")
    ws = L.workspace(dev)
    for rows, c in SHAPES:
        y = torch.randn(rows * c, device=dev)
        out = torch.empty_like(y)
        g, b = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
        mean, rstd = torch.empty(c, device=dev), torch.empty(c, device=dev)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        fwd = (y.data_ptr(), g.data_ptr(), b.data_ptr(), out.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rm.data_ptr(),
               rv.data_ptr(), 0, rows, c, 1e-5, 0.1, 1, 1, 0.01, ws.data_ptr(), ws.numel())
        t_f = time_call(lib.movae_bn_act_fwd, fwd, a.reps)
        line = f"{rows:7d} x {c:4d} ({rows * c * 4 / 1e6:6.1f} MB) | fwd {t_f:6.1f}us {3 * rows * c * 4 / t_f / 1e6:5.2f} TB/s |"
        for G in (1, 2):
            dout = torch.randn(G * rows * c, device=dev)
            dy = torch.empty_like(dout)
            dgs = [torch.empty(c, device=dev) for _ in range(G)]
            dbs = [torch.empty(c, device=dev) for _ in range(G)]
            arr = C.c_void_p * G
            bwd = (G, dout.data_ptr(), y.data_ptr(), g.data_ptr(), b.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dy.data_ptr(),
                   arr(*[t.data_ptr() for t in dgs]), arr(*[t.data_ptr() for t in dbs]), rows, c, 1, 0.01, 0, ws.data_ptr(), ws.numel())
            t_b = time_call(lib.movae_bn_act_bwd_grouped, bwd, a.reps)
            line += f" bwd G={G} {t_b:6.1f}us {(2 + 3 * G) * rows * c * 4 / t_b / 1e6:5.2f} TB/s |"
        print(line, flush=True)


if __name__ == "__main__":
    main()
