"""`ConfidenceModule` - the reference's confidence head (PhysDock/models/layers/confidence_module.py:13-88) on the HIP
kernels.

The reference keeps this module out of the released model (model.py:15,68 are commented out; the file's first line says
so), but its config block (`config.model.confidence_module`, configs.py:141-150) and the class are part of the package.
Same constructor signature, same parameter names (strict ``load_state_dict`` of a reference module's state dict works),
same ``forward(batch, s, z, x_pred) -> (p_pae, p_pde, p_plddt)``.  The Pairformer and AtomTransformer stacks inside it run
through the trunk's kernels (engine.Engine.pairformer / atom_transformer); three small element-wise kernels
(csrc/confidence.hip) cover the entry and exit.  No PyTorch compute fallback: on a machine without the built library or
without a GPU the call raises.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .configs import ConfigDict
from .engine import Engine
from .model import _register
from .packing import PackedWeights
from .params import confidence_param_shapes


class ConfidenceModule(nn.Module):
    def __init__(self, c_a: int, c_ap: int, c_s: int, c_z: int, inf: float, eps: float, no_blocks_heads: int,
                 no_blocks_atom: int = 3, c_pae: int = 64, c_pde: int = 64, c_plddt: int = 50):
        super().__init__()
        self.dims = dict(c_a=c_a, c_ap=c_ap, c_s=c_s, c_z=c_z, no_blocks_heads=no_blocks_heads, no_blocks_atom=no_blocks_atom,
                         c_pae=c_pae, c_pde=c_pde, c_plddt=c_plddt)
        for k in ("c_a", "c_s", "c_z"):
            assert self.dims[k] % 32 == 0, f"{k} must be a multiple of the head width 32"
        self.inf, self.eps = float(inf), float(eps)
        for name, shape in confidence_param_shapes(**self.dims).items():
            _register(self, name, torch.zeros(shape))
        self._engine: Optional[Engine] = None
        self.register_load_state_dict_post_hook(lambda m, k: m._invalidate())

    def _invalidate(self):
        self._engine = None

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._invalidate()
        return r

    def engine(self, device) -> Engine:
        if self._engine is None or self._engine.device != device:
            if device.type != "cuda":
                raise RuntimeError("physdock_amd.ConfidenceModule runs on an MI355X (HIP) device only; there is no CPU path "
                                   "(the CPU oracle lives in oracle/ and is test-only)")
            params = {"confidence_module." + k: v for k, v in self.state_dict().items()}
            if any(v.device != device for v in params.values()):
                raise RuntimeError("module parameters and batch must be on the same device")
            cfg = ConfigDict({"model": {"diffusion_conditioning": {"inf": self.inf, "eps": self.eps}}})
            self._engine = Engine(PackedWeights(params, cfg), cfg, device)
        return self._engine

    @torch.no_grad()
    def forward(self, batch, s: torch.Tensor, z: torch.Tensor, x_pred: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """batch keys read: token_id_to_centre_atom_id [T], atom_id_to_token_id [A], ap_mask [A,A], z_mask [T,T]
        (confidence_module.py:62-65); s [T,c_s], z [T,T,c_z]; x_pred [B,A,3] of which only pose 0 is used (:66,80)."""
        device = x_pred.device
        eng = self.engine(device)
        T, A = s.shape[0], x_pred.shape[-2]
        # layout-only boundary work: dtypes, contiguity and - for token / atom counts that are not multiples of 4 - padding
        # with masked tokens / atoms exactly like PhysDock._prepare_batch (padded entries are inert under the masks)
        pa, pt = (-A) % 4, (-T) % 4
        if pa and not pt:
            pt = 4
        b = {"_A_real": A, "_T_real": T}
        b["ap_mask"] = F.pad(batch["ap_mask"].float(), (0, pa, 0, pa)).contiguous()
        b["z_mask"] = F.pad(batch["z_mask"].float(), (0, pt, 0, pt)).contiguous()
        a2t = batch["atom_id_to_token_id"].to(torch.int64)
        b["atom_id_to_token_id"] = torch.cat([a2t, torch.full((pa,), T, device=device, dtype=torch.int64)]).contiguous()
        ctr = batch["token_id_to_centre_atom_id"].to(torch.int64)
        b["token_id_to_centre_atom_id"] = torch.cat([ctr, torch.zeros(pt, device=device, dtype=torch.int64)]).contiguous()
        s_p = F.pad(s.float(), (0, 0, 0, pt)).contiguous()
        z_p = F.pad(z.float(), (0, 0, 0, pt, 0, pt)).contiguous().reshape((T + pt) * (T + pt), -1)
        x0 = F.pad(x_pred[0].float(), (0, 0, 0, pa)).contiguous()
        p_pae, p_pde, p_plddt = eng.confidence(b, s_p, z_p, x0, self.dims)
        if pa or pt:
            p_pae, p_pde, p_plddt = p_pae[:T, :T].contiguous(), p_pde[:T, :T].contiguous(), p_plddt[:A].contiguous()
        return p_pae, p_pde, p_plddt

    @classmethod
    def from_config(cls, config):
        """`ConfidenceModule(**config.model.confidence_module)` as the reference would build it (model.py:68)"""
        return cls(**dict(config.model.confidence_module))

