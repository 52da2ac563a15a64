import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# the CPU oracle runs small convolutions: PyTorch's default of one thread per visible core (128 on a GPU box, whose CPU
# share is 16) oversubscribes them and is ~10x slower than 16 threads
torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


#: child processes of tests/test_dp_two_ranks_gpu.py, started at the end of collection -- i.e. before any test (and so
#: before anything in this process) has touched the GPU: a process that has initialised HIP must not start GPU children
DP_CHILDREN = {}


def pytest_collection_finish(session):
    import socket
    import subprocess
    import tempfile

    if not any(it.name.startswith("test_two_rank_") for it in session.items) or not os.path.exists("/dev/kfd"):
        return
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = tempfile.mkdtemp(prefix="movae_dp2_")
    worker = os.path.join(ROOT, "tests", "dp_worker.py")
    procs = []
    for r in range(2):
        log = open(os.path.join(out, f"rank{r}.log"), "w")
        procs.append(subprocess.Popen([sys.executable, worker, "--rank", str(r), "--world", "2", "--port", str(port), "--out", out],
                                      stdout=log, stderr=subprocess.STDOUT))
    DP_CHILDREN.update(out=out, procs=procs)


def load_golden(name):
    """npz fixtures are plain arrays (allow_pickle stays False)."""
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def meta_of(fx):
    m = {}
    for s in fx["meta"]:
        k, v = str(s).split("=", 1)
        m[k] = v
    return m


def cfg_from_meta(m, lambda_strings=None):
    import ast

    kw = {}
    for k in ("latent_dim", "embedding_dim", "num_embeddings", "num_residual_layers", "anneal_steps"):
        if k in m:
            kw[k] = int(m[k])
    if "hidden_dims" in m:
        kw["hidden_dims"] = list(ast.literal_eval(m["hidden_dims"]))
    kw["recons_objective"] = m.get("objective", "mse")
    return dict(arch=m["arch"], input_size=int(m["input_size"]), batch_size=int(m["B"]),
                dataset_size=int(m["dataset_size"]), **kw)


@pytest.fixture(scope="session")
def gpu_device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
