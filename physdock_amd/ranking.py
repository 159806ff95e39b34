"""Pose ranking / selection - the consumer of the sampler's output (reference redocking.py:326-423).

Device side: align every accepted pose into the ground-truth frame (pocket-weighted Kabsch, redocking.py:341-342),
ligand RMSD to the ground truth (:382), the pairwise ligand-RMSD matrix (:389-390) and the template re-selection
metric (:326-335).  Host side: the K-means(5) + medoid choice on that (n x n, n <= ~100) matrix exactly as the
reference does it with scikit-learn (:392-416); when scikit-learn is missing a deterministic Lloyd iteration with
farthest-point seeding is used instead (documented divergence: the cluster labels then differ from sklearn's).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .model import weighted_rigid_align


def pairwise_ligand_rmsd(x_aligned: torch.Tensor, ligand_idx: torch.Tensor, x_gt: torch.Tensor | None = None):
    """x_aligned [n,A,3] (device), ligand_idx int32 [L] -> (D [n,n], rmsd_to_gt [n] or None)"""
    L_ = ops._lib.init()
    n, A = x_aligned.shape[0], x_aligned.shape[1]
    x = x_aligned.float().contiguous()
    D = torch.empty(n, n, device=x.device)
    r = torch.empty(n, device=x.device) if x_gt is not None else None
    ops.check(L_.pd_pairwise_rmsd(ops.ptr(x), ops.ptr(ligand_idx), ops.ptr(x_gt.float().contiguous()) if x_gt is not None else None,
                                  ops.ptr(D), ops.ptr(r) if r is not None else None, n, A, int(ligand_idx.numel()), ops.stream()),
              "pd_pairwise_rmsd")
    return D, r


def get_representatives(distance_matrix: np.ndarray, num_clusters: int = 5):
    """redocking.py:392-408: K-means on the rows of the distance matrix, medoid (min mean in-cluster distance) per cluster"""
    n = len(distance_matrix)
    coords = np.asarray(distance_matrix, dtype=np.float64).reshape(n, n)
    try:
        from sklearn.cluster import KMeans
        labels = KMeans(n_clusters=num_clusters, random_state=0).fit(coords).labels_
    except ImportError:                       # deterministic fallback, see module docstring
        centers = [int(np.argmin(coords.mean(1)))]
        for _ in range(1, num_clusters):
            d = np.min([((coords - coords[c]) ** 2).sum(1) for c in centers], axis=0)
            centers.append(int(np.argmax(d)))
        cent = coords[centers].copy()
        for _ in range(50):
            labels = np.argmin(((coords[:, None] - cent[None]) ** 2).sum(-1), axis=1)
            new = np.stack([coords[labels == c].mean(0) if (labels == c).any() else cent[c] for c in range(num_clusters)])
            if np.allclose(new, cent):
                break
            cent = new
    reps = []
    for c in range(num_clusters):
        idx = np.where(labels == c)[0]
        avg = np.mean(distance_matrix[idx, :], axis=0)
        reps.append(int(idx[np.argmin(avg[idx])]))
    return reps


def rank_poses(x_pred: torch.Tensor, x_gt: torch.Tensor, align_weights: torch.Tensor, is_ligand_atom: torch.Tensor,
               num_clusters: int = 5):
    """Accepted poses [n,A,3] -> dict(order=ranked pose ids (global medoid first, redocking.py:410-418),
    rmsd=ligand RMSD to x_gt of the ranked poses, x_aligned, dist)."""
    x_al = weighted_rigid_align(x_gt[None].expand(x_pred.shape[0], -1, -1).contiguous(), x_pred, align_weights)
    lig = torch.nonzero(is_ligand_atom.to(x_pred.device) > 0).flatten().to(torch.int32)
    D, r = pairwise_ligand_rmsd(x_al, lig, x_gt)
    Dh, rh = D.cpu().numpy().astype(np.float64), r.cpu().numpy()
    n = len(Dh)
    if n > num_clusters:
        ids = get_representatives(Dh, num_clusters)
        first = get_representatives(Dh, 1)[0]
        if first in ids:
            ids.remove(first)
            ids = [first] + ids
        else:
            ids = [first] + ids[:num_clusters - 1]
    else:
        ids = list(range(n))
    return {"order": ids, "rmsd": [float(rh[i]) for i in ids], "x_aligned": x_al, "dist": D, "rmsd_all": r}
