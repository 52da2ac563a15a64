"""Where the data-parallel step's time goes (one rank over the real RCCL backend; run on a GPU box):
graph 1 (forward, backward to the features), graph 1b (shared trunk + aggregation), the two bucket all-reduces, graph 2 (clip, Adam), each timed alone with
HIP events, against the whole step and the single-device graph.  `python tools/dp_breakdown.py [C2]`"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MOVAE_FORCE_DP", "1")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

import bench  # noqa: E402


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    import movae_amd  # noqa: F401
    from movae_amd.parallel import DataParallelGrads
    from movae_amd.train import GraphedTrainStep

    cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C2"])
    dev = torch.device("cuda:0")
    t_single = float("nan")
    if "nosingle" not in sys.argv:  # (the single-device graph first: reference point, and it warms the allocator / kernels)
        net, opt, agg, a, pool = bench.build_workload(cfg, dev, capturable=True)
        single = GraphedTrainStep(net, opt, agg, a, pool[0])
        t_single = timed(lambda: single.step(pool[0]))
    dp = DataParallelGrads.from_env()
    net2, opt2, agg2, a2, pool2 = bench.build_workload(cfg, dev, capturable=True)
    dp.attach(net2)
    gs = GraphedTrainStep(net2, opt2, agg2, a2, pool2[0], dp=dp)
    it = [0]

    def rotating():
        it[0] += 1
        gs.step(pool2[it[0] % len(pool2)])

    out = {"single_graph_us": t_single, "dp_step_us": timed(lambda: gs.step(pool2[0])), "dp_step_rotating_inputs_us": timed(rotating),
           "graph1_us": timed(gs.graph.replay), "all_reduce_a_us": timed(lambda: gs.reduce(gs.flat_a)),
           "bucket_a_MB": gs.flat_a.numel() * 4 / 1e6}
    if gs.graph2 is not None:  # (None: the collective and the optimizer are inside graph 1)
        out["graph2_us"] = timed(gs.graph2.replay)
    if gs.graph_b is not None:
        out.update(graph1b_us=timed(gs.graph_b.replay), all_reduce_b_us=timed(lambda: gs.reduce(gs.flat_b)),
                   bucket_b_MB=gs.flat_b.numel() * 4 / 1e6)
    print({k: round(v, 1) for k, v in out.items()})
    dp.shutdown()


if __name__ == "__main__":
    main()
