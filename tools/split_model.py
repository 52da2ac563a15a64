#!/usr/bin/env python3
"""Development tool: evaluate a split-K selection rule against `conv_microbench.py --sweep-split` logs.

    python tools/split_model.py gpurun_out/sweep_c2.log gpurun_out/sweep_c3.log

For every (shape, pass) line the rule picks a split factor from the shape's tile count / k-tile count /
output size; the table shows the measured time at that factor next to the measured optimum.  The rule
here mirrors choose_split() in mo-vae_amd/csrc/conv_igemm.hip -- keep the two in sync.
"""
import math
import re
import sys


def ceil_div(a, b):
    return -(-a // b)


def problem(kind, n, hi, wi, ci, ho, wo, co, k, s, pas, bm, bn):
    """-> (form, tiles, nk, out_bytes) the way the launchers compute them (BK2 = 32)."""
    if (kind, pas) in (("conv", "fwd"), ("convT", "dgrad")):
        if kind == "conv":
            M, N, K = n * ho * wo, co, k * k * ci
        else:
            M, N, K = n * hi * wi, ci, k * k * co
        return "fwd", ceil_div(M, bm) * ceil_div(N, bn), ceil_div(K, 32), M * N * 4
    if (kind, pas) in (("conv", "dgrad"), ("convT", "fwd")):
        if kind == "conv":
            H, W, N, Cr = hi, wi, ci, co
        else:
            H, W, N, Cr = ho, wo, co, ci
        Mmax = n * ceil_div(H, s) * ceil_div(W, s)
        return ("bwd", ceil_div(Mmax, bm) * ceil_div(N, bn) * s * s, ceil_div(ceil_div(k, s) ** 2 * Cr, 32), n * H * W * N * 4)
    if kind == "conv":
        M, N, K = co, k * k * ci, n * ho * wo
    else:
        M, N, K = ci, k * k * co, n * hi * wi
    return "wgrad", ceil_div(M, bm) * ceil_div(N, bn), ceil_div(K, 32), M * N * 4


def choose_split(form, tiles, nk, out_bytes, bm, bn):
    """The candidate rule (mirror of the C++)."""
    big = bm * bn >= 128 * 128
    cap = 512                                    # two co-resident blocks on each of the 256 CUs
    tk = {"fwd": 1.0, "bwd": 1.25, "wgrad": 0.9}[form] * (bm * bn) / 4096.0 * (0.83 if big else 1.0)
    best, best_t = 1, None
    for S in range(1, min(nk, 256) + 1):
        kps = ceil_div(nk, S)
        if ceil_div(nk, kps) != S:
            continue
        blocks = tiles * S
        rounds = ceil_div(blocks, cap)
        share = 1.0 if blocks <= 256 else 1.6    # a CU that holds two blocks runs each ~1.6x slower
        t = rounds * (2.0 + kps * tk * share)
        if S > 1:
            t += 2.5 + S * out_bytes * 0.87e-6
        if best_t is None or t < best_t * 0.97:   # prefer fewer splits unless the gain is clear
            best, best_t = S, t
    return best


def main():
    rows = []
    for path in sys.argv[1:]:
        for line in open(path):
            m = re.match(r"\s*(\d+) (conv|convT)\s+(\d+)x(\d+)x(\d+)x(\d+)->(\d+)x(\d+)x(\d+) k(\d+)s(\d+)\s+[\d.]+GF \| "
                         r"(\w+)\s+(\S+)\s+best S=\s*(\d+)\s+([\d.]+)us \| (.*)", line)
            if not m:
                continue
            idx, kind = int(m.group(1)), m.group(2)
            n, hi, wi, ci, ho, wo, co, k, s = (int(m.group(i)) for i in range(3, 12))
            pas, kern = m.group(12), m.group(13)
            km = re.match(r"igemm2?_\w+<(\d+),(\d+)>", kern)
            if not km:
                continue
            bm, bn = int(km.group(1)), int(km.group(2))
            meas = {int(a): float(b) for a, b in (x.split(":") for x in m.group(16).split())}
            form, tiles, nk, ob = problem(kind, n, hi, wi, ci, ho, wo, co, k, s, pas, bm, bn)
            rows.append((path[-6:-4], idx, pas, kern, form, tiles, nk, ob, meas))
    tot_rule = tot_best = tot_cur = 0.0
    for tag, idx, pas, kern, form, tiles, nk, ob, meas in rows:
        S = choose_split(form, tiles, nk, ob, *map(int, re.match(r".*<(\d+),(\d+)>", kern).groups()))
        # nearest measured factor
        keys = sorted(kk for kk in meas if kk > 0)
        Sm = min(keys, key=lambda kk: abs(math.log(kk) - math.log(S)))
        best = min(meas[kk] for kk in keys)
        tot_rule += meas[Sm]
        tot_best += best
        tot_cur += meas[0]
        flag = "" if meas[Sm] <= best * 1.08 else "  <--"
        print(f"{tag} {idx:2d} {pas:5s} {kern:22s} tiles {tiles:4d} nk {nk:5d} out {ob / 1e6:6.2f}MB  rule S={S:3d} (~{Sm:3d}) "
              f"{meas[Sm]:6.1f}us  best {best:6.1f}  current {meas[0]:6.1f}{flag}")
    print(f"sum: rule {tot_rule:.1f}  best {tot_best:.1f}  current {tot_cur:.1f}")


if __name__ == "__main__":
    main()
