"""Development probe (GPU): is the fp32 MFMA accumulation of the conv kernels unbiased?  1x1 convolutions of all-positive and
of mixed-sign data against float64, next to torch's fp32 matmul: mean signed error and rms error relative to sum |products|.
(Measured: no bias, rms grows like sqrt(K) * 1.3e-8 -- round-to-nearest sequential accumulation, identical to hipBLASLt's.)

    python tools/mfma_rounding_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import movae_amd
from movae_amd import nn as mnn
dev = torch.device("cuda:0")
torch.manual_seed(0)
for K in (64, 256, 1024, 4096):
    for sign in ("pos", "mixed"):
        x = torch.rand(8192, 1, 1, K, device=dev) if sign == "pos" else torch.randn(8192, 1, 1, K, device=dev)
        conv = mnn.Conv2d(K, 256, 1, 1, 0).to(dev)
        with torch.no_grad():
            if sign == "pos":
                conv.weight.uniform_(0, 1)
            conv.bias.zero_()
            y = conv(x).reshape(8192, 256)
            w = conv.weight.reshape(256, K)
            y64 = x.reshape(8192, K).double() @ w.double().T
            yt = x.reshape(8192, K) @ w.T
            scale = (x.reshape(8192, K).double().abs() @ w.double().abs().T)  # sum of |products|
            for name, got in (("hip", y), ("torch", yt)):
                e = (got.double() - y64) / scale
                print(f"K={K:5d} {sign:5s} {name:5s} mean signed err/sum|prod| {float(e.mean()):+.2e}  rms {float(e.pow(2).mean().sqrt()):.2e}")
