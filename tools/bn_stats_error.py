"""Development probe (GPU): error of the fused BatchNorm statistics (conv epilogue / split-K reduce partials, fp64 fold) and of the
stand-alone statistics kernel against fp64 statistics of the same y, next to torch.var_mean in fp32, at the config shapes.

    python tools/bn_stats_error.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
import movae_amd
from movae_amd import nn as mnn, ops
from test_hip_fused_bn import STAT_SHAPES
dev = torch.device("cuda:0")
for fs in (True, False):
    ops.FUSE_STATS = fs
    print("FUSE_STATS", fs)
    for shape in STAT_SHAPES:
        kind, B, size, cin, cout, k, s, p, op = shape
        torch.manual_seed(11)
        conv = (mnn.Conv2d(cin, cout, k, s, p) if kind == "conv" else mnn.ConvTranspose2d(cin, cout, k, s, p, op)).to(dev)
        bn = mnn.BatchNorm2d(cout).to(dev).train()
        x = torch.randn(B, size, size, cin, device=dev)
        out = mnn.Stack(conv, bn, mnn.LeakyReLU()).to(dev)(x)
        y = out.y.detach().double().reshape(-1, cout)
        mean = y.mean(0); var = ((y - mean) ** 2).mean(0); rstd = 1.0 / torch.sqrt(var + bn.eps)
        got_rstd = out.scale.detach().double(); got_mean = -out.shift.detach().double() / got_rstd
        # fp32 torch two-pass for comparison
        y32 = out.y.detach().reshape(-1, cout)
        v32, m32 = torch.var_mean(y32, 0, unbiased=False)
        r32 = 1.0 / torch.sqrt(v32.double() + bn.eps)
        print(f"  {shape}: rstd rel err hip {float(((got_rstd-rstd)/rstd).abs().max()):.2e} torch32 {float(((r32-rstd)/rstd).abs().max()):.2e} | mean err/std hip {float(((got_mean-mean)*rstd).abs().max()):.2e} torch32 {float(((m32.double()-mean)*rstd).abs().max()):.2e}")
