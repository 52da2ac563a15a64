import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
