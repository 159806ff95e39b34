import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (REPO, os.path.join(REPO, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in z.files:
        v = z[k]
        if v.dtype.kind in "US":
            out[k] = v.tolist()                 # lists of names stay Python strings
        else:
            out[k] = torch.from_numpy(v) if v.ndim > 0 else v.item()
    return out


def golden_weights(g):
    return {k[2:]: v for k, v in g.items() if k.startswith("W:")}


def golden_noise(g):
    return {k[len("noise_"):]: v for k, v in g.items() if k.startswith("noise_")}


def rmsd(a, b):
    return float(((a - b) ** 2).sum(-1).mean(-1).sqrt().max())


@pytest.fixture(scope="session")
def small_model_inputs():
    from physdock_amd.configs import small_config
    from physdock_amd.params import param_shapes, seeded_state_dict
    from physdock_amd.synthetic import small_batch
    cfg = small_config()
    return cfg, seeded_state_dict(param_shapes(cfg), seed=0), small_batch(seed=0)
