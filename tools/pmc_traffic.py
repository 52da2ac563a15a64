"""rocprofv3 counter CSVs -> profiles/pmc_traffic_<cfg>.json (memory-side bytes per launch of every kernel).

Collect on the GPU box in two separate passes, each with --kernel-trace only (MI355X_MICROARCH.md, HBM section):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc/f -o f -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc/w -o w -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5
    python tools/pmc_traffic.py gpurun_out/pmc/f gpurun_out/pmc/w profiles/pmc_traffic_C2.json

FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch; gfx950 tallies 128-byte read requests as 64 bytes, so the read
side is doubled (the guide's gfx950 correction).  Infinity-Cache hits are included: this is memory-side traffic, an upper
bound on HBM bytes.  Kernel names are reduced to `symbol<template args>` -- the form bench.py's roofline block uses."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^(v2|thin|lin|kg)::", "", name)
    depth, out = 0, []
    for ch in name:  # cut the argument list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    name = re.sub(r",\s+", ",", "".join(out)).strip()
    # igemm2_fwd / _bwd / _wgrad carry a trailing `BF` template flag (bf16 operands, opt-in): the fp32 instantiations keep the
    # names bench.py and the earlier rounds' profiles use
    return re.sub(r"^(igemm2_(?:fwd|bwd|wgrad)<[^>]*),false>$", r"\1>", name)


def per_kernel(directory, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for path in glob.glob(f"{directory}/**/*counter_collection.csv", recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == counter:
                    k = short(row["Kernel_Name"])
                    tot[k] += float(row["Counter_Value"]) * 1024.0
                    cnt[k] += 1
    return tot, cnt


def main():
    fdir, wdir, out = sys.argv[1:4]
    ft, fc = per_kernel(fdir, "FETCH_SIZE")
    wt, wc = per_kernel(wdir, "WRITE_SIZE")
    kernels = {}
    for k in ft:
        fetch = 2.0 * ft[k] / max(fc[k], 1)
        write = wt.get(k, 0.0) / max(wc.get(k, 0), 1)
        kernels[k] = {"launches": fc[k], "fetch_bytes": round(fetch), "write_bytes": round(write), "hbm_bytes_per_launch": round(fetch + write)}
    doc = {"method": __doc__.split("\n\n")[2].replace("\n", " ") + " Collected as in this file's docstring (tools/pmc_traffic.py).",
           "unit": "bytes per launch", "kernels": kernels}
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(out, len(kernels), "kernels")


if __name__ == "__main__":
    main()
