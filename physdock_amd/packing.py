"""Weight re-packing done once at load time (pure layout changes, no arithmetic on values
except folding a norm gain / LayerNorm affine into the following projection)."""
from __future__ import annotations

import torch


def pack_glu(Wa, Wb, ba=None, bb=None):
    """Interleave two [H,K] projections in 32-row groups -> [2H,K] so that every 64 packed
    output columns hold [a(32) | b(32)] of the same hidden indices (pd_gemm glu epilogue)."""
    H, K = Wa.shape
    assert Wb.shape == (H, K) and H % 32 == 0
    W = torch.stack([Wa.reshape(H // 32, 32, K), Wb.reshape(H // 32, 32, K)], dim=1).reshape(2 * H, K)
    b = None
    if ba is not None or bb is not None:
        ba = ba if ba is not None else torch.zeros(H, dtype=Wa.dtype, device=Wa.device)
        bb = bb if bb is not None else torch.zeros(H, dtype=Wa.dtype, device=Wa.device)
        b = torch.stack([ba.reshape(H // 32, 32), bb.reshape(H // 32, 32)], dim=1).reshape(2 * H)
    return W.contiguous(), (b.contiguous() if b is not None else None)
