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


def pool_rounding_bound(prefix_exp_end, chunk_sizes, s_pool):
    """Per-element rounding the REFERENCE's own token pooling (diffusion_conditioning.py:168-176: cumsum over all atoms -> gather the
    chunk ends -> diff -> / (n + 1e-3)) puts on its result: each fp32 prefix is the exact (double) prefix rounded to nearest, half an
    ulp each for the end and the start prefix of a token (the start prefix of token t is the end prefix of token t - 1, zero for the
    first), divided by n + 1e-3, plus two ulps of the result for the difference and the division themselves.
    prefix_exp_end [T, C] int8: frexp exponents of the end prefixes (ulp(C) = 2^(e - 24)); chunk_sizes [T]; s_pool [T, C]."""
    e = prefix_exp_end.double()
    half_ulp_end = torch.pow(2.0, e - 25)
    half_ulp_start = torch.cat([torch.zeros_like(half_ulp_end[:1]), half_ulp_end[:-1]])
    n = chunk_sizes.double()[:, None] + 1e-3
    sp = s_pool.abs().float()
    ulp_res = (torch.nextafter(sp, torch.full_like(sp, 3e38)) - sp).double()
    return ((half_ulp_end + half_ulp_start) / n + 2 * ulp_res).float()


def rmsd(a, b):
    return float(((a - b) ** 2).sum(-1).mean(-1).sqrt().max())


@pytest.fixture(scope="session")
def small_model_inputs():
    from physdock_amd.configs import small_config
    from physdock_amd.params import param_shapes, seeded_state_dict
    from physdock_amd.synthetic import small_batch
    cfg = small_config()
    return cfg, seeded_state_dict(param_shapes(cfg), seed=0), small_batch(seed=0)
