#!/usr/bin/env python3
"""Would dgrad and wgrad of one layer gain from running side by side?  For every C2 layer: N iterations of
(dgrad; wgrad) queued on ONE stream vs dgrad on stream A and wgrad on stream B (separate scratch arenas), wall time per
pair with a device sync around the whole batch.  Development probe for a horizontally fused backward launch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import movae_amd  # noqa: E402,F401
import movae_amd._lib as L  # noqa: E402
from conv_microbench import C2  # noqa: E402


def main():
    lib = L.load()
    dev = torch.device("cuda:0")
    ws0, ws1 = L.workspace(dev), L.workspace(dev, slot=1)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    N = 200
    tot_seq = tot_par = 0.0
    for i, (kind, n, hi, wi, ci, ho, wo, co, k, s, p) in enumerate(C2):
        x = torch.randn(n * hi * wi * ci, device=dev)
        y = torch.randn(n * ho * wo * co, device=dev)
        w = torch.randn(co * k * k * ci, device=dev) * 0.05
        dx, dw = torch.empty_like(x), torch.empty_like(w)
        geom = (n, hi, wi, ci, ho, wo, co, k, k, s, p)
        pre = "movae_convT2d_" if kind == "convT" else "movae_conv2d_"
        fd, fw = getattr(lib, pre + "dgrad"), getattr(lib, pre + "wgrad")

        def dgrad(ws, st):
            fd(y.data_ptr(), w.data_ptr(), dx.data_ptr(), *geom, ws.data_ptr(), ws.numel(), st.cuda_stream)

        def wgrad(ws, st):
            fw(y.data_ptr(), x.data_ptr(), dw.data_ptr(), 0, *geom, 0, ws.data_ptr(), ws.numel(), st.cuda_stream)

        res = []
        for par in (False, True):
            for it in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(N):
                    dgrad(ws0, sa)
                    wgrad(ws1 if par else ws0, sb if par else sa)
                torch.cuda.synchronize()
                t = (time.perf_counter() - t0) / N * 1e6
            res.append(t)
        tot_seq += res[0]
        tot_par += res[1]
        print(f"{i:2d} {kind:5s} {n}x{hi}x{wi}x{ci}->{ho}x{wo}x{co}  one stream {res[0]:6.1f} us   two streams {res[1]:6.1f} us", flush=True)
    print(f"sum one stream {tot_seq:.1f} us, two streams {tot_par:.1f} us")


if __name__ == "__main__":
    main()
