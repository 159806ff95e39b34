"""Multi-round redocking driver - the caller of the hot path (reference redocking.py:156-356), without its file and
RDKit I/O.

What the reference does per system, and what is kept here:
  * rounds (redocking.py:181-183): round 0 samples freely; further rounds only with physics correction, each with
    the re-sampled MSA of that round (`batch_msa_feat[round]`, :188) and the template-projection branch of the
    sampler switched on (`align_ref_pos = round > 0`, `use_ref_mol_poses`, :283-299);
  * accept / reject (:303-317): the reference rebuilds the ligand with RDKit and compares chiral centres.  Here:
    `chirality=` (chirality.ChiralityReference: signed volumes of the stereocentres on the device, kernel pd_chirality)
    and / or an injected callable `accept_fn(x_pose_cpu [A,3]) -> bool`; default: accept.  Rejected poses go to a
    bounded deque (maxlen = max_samples, :164);
  * the adaptive threshold of the physics branch (:318-322): x1.15 after a round with any accepted pose, otherwise
    x0.7 with floor 1;
  * template pool for the next round (:323-335, and :293): accepted predicted ligands + the reference conformers
    closest to this round's poses under the sampler's own soft distance-difference metric (device, pd_template_match);
  * top-up with rejected poses when fewer than one round's worth was accepted (:336-337), alignment of every kept
    pose into the ground-truth frame with pocket weights (:341-342) and ranking (ranking.py, :357-423).
Conformer generation (ETKDG, :231-243) is an input (`ref_mol_poses [C,L,3]`).  `ref_mol` (an RDKit molecule, a
`physdock_amd.mmff.MMFFTerms` table, or any object with `sampler_kwargs={"relax_fn": ...}`) is handed to the sampler in
every round as the reference does (:292), which switches on the relaxation branch below the adaptive threshold; as in the
reference a molecule whose atom count differs from the crop's ligand is dropped and the ODE step scale becomes 1.5
(`ref_mol_num_error`, :195-196,292,296).
"""
from __future__ import annotations

from collections import deque
from typing import Callable, Dict, List, Optional

import torch

from . import ops
from .model import weighted_rigid_align


def ligand_atom_mask(batch: Dict[str, torch.Tensor]) -> torch.Tensor:
    """bool [A]: atoms of ligand tokens (redocking.py:189)"""
    return batch["is_ligand"][batch["atom_id_to_token_id"].long()].bool()


def pocket_align_weights(batch: Dict[str, torch.Tensor], use_pocket: bool = True) -> torch.Tensor:
    """per-atom weights of the final alignment, align_mode "pocket_ca" (redocking.py:197-201)"""
    idx = batch["atom_id_to_token_id"].long()
    w = (batch["s_mask"] * batch["is_protein"])[idx] if "s_mask" in batch and "is_protein" in batch \
        else 1.0 - batch["is_ligand"].float()[idx]
    if use_pocket:
        w = batch["pocket_res_feat"][idx] * w
    return w.float()


def template_scores(x_pred: torch.Tensor, ligand_idx: torch.Tensor, ref_mol_poses: torch.Tensor) -> torch.Tensor:
    """eps[b, c]: mean over ligand atom pairs of 1/4 sum_k sigmoid(|D_b - D_c| - {0.5, 1, 2, 4}) between the ligand of
    pose b and reference conformer c (redocking.py:326-331; same kernel as the in-loop template match, model.py:231-238)."""
    L_ = ops._lib.init()
    x = x_pred.float().contiguous()
    poses = ref_mol_poses.float().contiguous()
    B, A = x.shape[0], x.shape[1]
    C, L = poses.shape[0], poses.shape[1]
    assert ligand_idx.numel() == L, "reference conformers and ligand atoms differ in count"
    rd = torch.empty(C, L, L, device=x.device)
    ops.check(L_.pd_pose_dist(ops.ptr(poses), ops.ptr(rd), C, L, ops.stream()), "pd_pose_dist")
    eps = torch.empty(B, C, device=x.device)
    sel = torch.empty(B, dtype=torch.int32, device=x.device)
    ops.check(L_.pd_template_match(ops.ptr(x), ops.ptr(ligand_idx), ops.ptr(rd), None, None, ops.ptr(eps), ops.ptr(sel),
                                   B, A, L, C, ops.stream()), "pd_template_match")
    return eps


def select_reference_templates(x_pred: torch.Tensor, ligand_idx: torch.Tensor, ref_mol_poses: torch.Tensor, k: int) -> torch.Tensor:
    """indices of the k reference conformers closest to this round's poses: argsort of eps averaged over the poses
    (redocking.py:331-332)"""
    if k <= 0:
        return torch.empty(0, dtype=torch.long, device=x_pred.device)
    return torch.argsort(template_scores(x_pred, ligand_idx, ref_mol_poses).mean(0))[:k]


def next_gamma_factor(factor: float, any_accepted: bool) -> float:
    """redocking.py:318-322"""
    return factor * 1.15 if any_accepted else max(factor * 0.7, 1.0)


def _mol_num_atoms(ref_mol) -> Optional[int]:
    for attr in ("GetNumAtoms", "num_atoms"):
        v = getattr(ref_mol, attr, None)
        if v is not None:
            return int(v() if callable(v) else v)
    return None


def redock(model, batch: Dict[str, torch.Tensor], *, ref_mol=None, ref_mol_poses: Optional[torch.Tensor] = None,
           accept_fn: Optional[Callable[[torch.Tensor], bool]] = None, chirality=None, physics_correction: bool = False,
           max_samples: int = 5, max_rounds: int = 10, num_samples_per_round: int = 5, steps: int = 40,
           mmff_gamma_0_factor_start: float = 6.0, karras_noise_schedule_power: float = 1000, use_pocket: bool = True,
           align_weights: Optional[torch.Tensor] = None, ranking: bool = True, seed: Optional[int] = None,
           sampler_kwargs: Optional[dict] = None, infer_meta_data=None, reuse_conditioning: bool = True) -> dict:
    """One system through the reference's round loop (defaults = redocking.py:33-59).  `batch` holds device tensors
    as for `model.sample_diffusion`; with physics correction it may hold `batch_msa_feat [rounds,S,T,34]`.
    Returns dict(poses [n,A,3] in the ground-truth frame, accepted (count before the top-up), rounds (per-round log),
    gamma_factor, ranking (ranking.rank_poses output or None)); with `infer_meta_data` (the loader's per-system naming
    tables) also `pdb_blocks` / `receptor_pdb_blocks`: the `write_pdb_block` text of every kept pose (redocking.py:342-345),
    formatted on the device for the whole batch (pdbio.py).
    `reuse_conditioning`: rounds that see the SAME features (no `batch_msa_feat` re-sampling) share one run of the conditioning
    trunk - the reference recomputes it in every `sample_diffusion` call (model.py:179) with identical inputs and hence identical
    outputs; here round 0 returns its (a, ap, s, z) and later rounds take them through `conditioning=`.  Bit-identical poses."""
    if physics_correction and ref_mol_poses is None:
        raise ValueError("physics correction needs reference conformers (ref_mol_poses [C,L,3]); the reference generates "
                         "them with RDKit ETKDG (redocking.py:231-243), which this build does not include")
    batch = dict(batch)
    if ref_mol_poses is not None:            # the reference accepts host conformers (`.to(device)`, model.py:185)
        ref_mol_poses = ref_mol_poses.to(batch["x_gt"].device)
    is_lig = ligand_atom_mask(batch)
    ligand_idx = torch.nonzero(is_lig).flatten().to(torch.int32)
    accept: List[torch.Tensor] = []
    reject: deque = deque([], maxlen=max_samples)
    ligand_templates: List[torch.Tensor] = []
    reference_templates: List[torch.Tensor] = []
    factor = float(mmff_gamma_0_factor_start)
    log = []
    kw = dict(sampler_kwargs or {})
    n_mol = _mol_num_atoms(ref_mol) if ref_mol is not None else None
    ref_mol_num_error = ref_mol is None or (n_mol is not None and n_mol != int(is_lig.sum()))     # redocking.py:195-196
    reuse = bool(reuse_conditioning and getattr(model, "supports_conditioning_reuse", False))
    cond = None
    for rnd in range(max_rounds):
        if rnd > 0 and not physics_correction:
            break
        if rnd >= 1 and "batch_msa_feat" in batch:
            if rnd >= batch["batch_msa_feat"].shape[0]:
                raise ValueError(f"batch_msa_feat holds {batch['batch_msa_feat'].shape[0]} re-sampled MSAs, round {rnd} needs "
                                 "its own (the reference loads num_recycles = max_rounds of them, redocking.py:83)")
            batch["msa_feat"] = batch["batch_msa_feat"][rnd]
            cond = None                      # a re-sampled MSA: this round's trunk is its own
        templates = torch.stack(ligand_templates + reference_templates, 0) if rnd > 0 else None
        call = dict(num_sample=num_samples_per_round, steps=steps, mmff_gamma_0_factor=factor, align_ref_pos=rnd > 0,
                    ref_mol=None if ref_mol_num_error else ref_mol, ref_mol_poses=templates,
                    use_ref_mol_poses=rnd != 0 and physics_correction,
                    ode_step_scale_eta=1.5 if ref_mol_num_error else 1.0,
                    karras_noise_schedule_power=karras_noise_schedule_power)
        if seed is not None:
            call.update(seed=seed + rnd)
        call.update(kw)
        if reuse and "conditioning" not in call and "return_conditioning" not in call:
            if cond is not None:
                call.update(conditioning=cond)
            elif physics_correction and rnd + 1 < max_rounds and "batch_msa_feat" not in batch:
                call.update(return_conditioning=True)
        with torch.no_grad():
            x_pred = model.sample_diffusion(batch, **call)
        if isinstance(x_pred, tuple):
            x_pred, cond = x_pred
        if rnd + 1 >= max_rounds:            # no later round can take it: the shared conditioning (clones of a [A,c], ap [A,A,c], s,
            cond = None                      # z [T,T,128] - hundreds of MB at large crops, per StreamPool replica) is released here
        # accept / reject (redocking.py:303-317): on the device when a ChiralityReference is given (one kernel, one [B]
        # mask to the host), else through the injected per-pose callable (which needs the poses on the host)
        dev_ok = chirality.accept(x_pred).tolist() if (physics_correction and chirality is not None) else None
        x_cpu = x_pred.cpu() if (physics_correction and accept_fn is not None) else x_pred
        flags = []
        for b, (x, xc) in enumerate(zip(x_pred, x_cpu)):
            ok = True
            if dev_ok is not None:
                ok = bool(dev_ok[b])
            if ok and physics_correction and accept_fn is not None:
                ok = bool(accept_fn(xc))
            flags.append(ok)
            if ok:
                ligand_templates.append(x[is_lig])
                accept.append(x)
            else:
                reject.append(x)
        log.append({"round": rnd, "gamma_factor": factor, "accepted": int(sum(flags)), "sampled": len(flags),
                    "templates": 0 if templates is None else int(templates.shape[0])})
        if physics_correction:
            factor = next_gamma_factor(factor, any(flags))
            if len(accept) >= max_samples:
                cond = None                  # accept count reached: the shared conditioning is released with the loop
                break
            used = select_reference_templates(x_pred, ligand_idx, ref_mol_poses, max_samples - len(ligand_templates))
            reference_templates = [ref_mol_poses[i] for i in used.tolist()]
    n_accepted = len(accept)
    if len(accept) < num_samples_per_round:
        accept = accept + list(reject)
    poses = torch.stack(accept[:max_samples], 0)
    w = align_weights if align_weights is not None else pocket_align_weights(batch, use_pocket)
    x_gt = batch["x_gt"].float()
    aligned = weighted_rigid_align(x_gt[None].expand(poses.shape[0], -1, -1).contiguous(), poses, w)
    out = {"poses": aligned, "accepted": n_accepted, "rounds": log, "gamma_factor": factor, "ranking": None}
    if ranking:
        from .ranking import rank_poses
        out["ranking"] = rank_poses(poses, x_gt, w, is_lig)
    if infer_meta_data is not None:
        from .pdbio import PdbTemplate
        out["pdb_blocks"] = PdbTemplate(infer_meta_data).blocks(aligned)
        out["receptor_pdb_blocks"] = PdbTemplate(infer_meta_data, receptor_only=True).blocks(aligned)
    return out


def redock_many(model, systems, *, streams: Optional[int] = None, **common) -> List[dict]:
    """The loop over systems of the reference's drivers (`redocking.py:128-154`: one `redocking(...)` call per input system;
    `screening.py:100-116`: one receptor x many ligands) on ONE GPU.  `systems`: an iterable of feature dicts, or of
    `(batch, per_system_kwargs)` pairs (`ref_mol`, `ref_mol_poses`, `chirality`, `infer_meta_data` ... differ per system); `common`:
    keyword arguments of `redock` shared by all.  Results in input order.

    Rounds of few samples cannot fill an MI355X (20 samples per round, the drivers' setting: 70 % of the per-pose rate of a 64-sample
    call), and the systems are independent, so by default two of them are in flight on two HIP streams whenever a round has fewer
    than 32 samples (`parallel.StreamPool`: one model replica, stream and host thread each; poses are bit-identical to the
    one-at-a-time run, `tests/test_concurrent_streams_gpu.py`, `tests/test_configs_3_5_gpu.py`).  `streams=1` runs them one by one;
    rounds of 32 or more samples already fill the chip and run one by one unless `streams` says otherwise.  Across GPUs the same list
    is dealt out by `parallel.map_systems`."""
    items = [(s, {}) if isinstance(s, dict) else (s[0], dict(s[1])) for s in systems]
    n = streams if streams is not None else (2 if int(common.get("num_samples_per_round", 5)) < 32 else 1)
    on_gpu = bool(items) and items[0][0]["x_gt"].is_cuda and hasattr(model, "config")
    if n <= 1 or len(items) <= 1 or not on_gpu:
        return [redock(model, b, **dict(common, **kw)) for b, kw in items]
    from .parallel import StreamPool
    pool = common.pop("pool", None) or StreamPool.for_model(model, n=n)      # cached on the model: replicas are built once
    return pool.map(lambda m, it: redock(m, it[0], **dict(common, **it[1])), items)
