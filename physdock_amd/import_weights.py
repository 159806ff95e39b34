"""Checkpoint loading with the reference's key convention (reference utils/import_weights.py:140-150):
the released `params.pt` is a flat state dict whose keys carry a 6-character prefix ("model.")."""
from __future__ import annotations

import torch


def import_state_dict(model: torch.nn.Module, ckpt_path, prefix_len: int = 6):
    params = torch.load(ckpt_path, weights_only=False, map_location="cpu")
    params = {k[prefix_len:]: v for k, v in params.items()}
    model.load_state_dict(params, strict=True)
    return model


def import_unicore_ckpt(model: torch.nn.Module, ckpt_path, load_ema_state: bool = True, remove_compile_prefix: bool = False):
    """Uni-Core training checkpoint (reference utils/import_weights.py:6-28): EMA weights when present."""
    ck = torch.load(ckpt_path, weights_only=False, map_location="cpu")
    params = None
    if load_ema_state:
        try:
            params = ck["ema"]["params"]
        except (KeyError, TypeError):
            params = None
    if params is None:
        params = ck["model"]
    n = 16 if remove_compile_prefix else 6
    model.load_state_dict({k[n:]: v for k, v in params.items()}, strict=True)
    return model
