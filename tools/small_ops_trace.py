"""Which Python lines issue the torch-native helper kernels (fill / cat / add / copy) of one eager step -- candidates
for removal from the captured graph.  `python tools/small_ops_trace.py [C2]` on a GPU box."""
import os
import sys
import traceback
from collections import Counter

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

WATCH = ("fill", "zero", "cat", "stack", "add", "copy", "ones", "mul", "div", "sum", "clone", "index", "sub")


class Spy(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.hits = Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        out = func(*args, **(kwargs or {}))
        if any(w in name for w in WATCH):
            dev = [a.device.type for a in list(args) + ([out] if isinstance(out, torch.Tensor) else []) if isinstance(a, torch.Tensor)]
            if "cuda" in dev:
                frames = [f for f in traceback.extract_stack() if "mo-vae_amd" in f.filename or "movae_amd" in f.filename]
                where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in frames[-3:][::-1]) or "(autograd engine)"
                self.hits[(name, where)] += 1
        return out


def main():
    import movae_amd  # noqa: F401
    from movae_amd.train import train_step

    cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C2"])
    dev = torch.device("cuda:0")
    net, opt, agg, a, pool = bench.build_workload(cfg, dev, capturable=True)
    for _ in range(2):
        train_step(net, pool[0], opt, agg, a)
    with Spy() as spy:
        train_step(net, pool[0], opt, agg, a)
    for (name, where), n in sorted(spy.hits.items(), key=lambda kv: -kv[1]):
        print(f"{n:3d}  {name:28s} {where}")


if __name__ == "__main__":
    main()
