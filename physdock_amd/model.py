"""`PhysDock` - drop-in for the reference model class on the sampling hot path.

Mirrors ``PhysDock.models.model.PhysDock`` (reference models/model.py:55-68): same
constructor (a config from ``PhysDockConfig``), same parameter names (strict
``load_state_dict`` of a reference checkpoint works, see params.py), same
``sample_diffusion(...)`` / ``forward(batch)`` signatures and return values.  Everything
between "feature tensors on the device" and "coordinates on the device" runs in the HIP
kernels of libphysdock_hip.so; there is no PyTorch compute fallback - on a machine
without the built library or without a GPU the calls raise.

Extensions (keyword-only, not in the reference): ``noise=`` injects pre-drawn random
numbers (parity mode; same draw order as the reference, see oracle), ``seed=`` /
``sample_offset=`` select the on-device Philox streams (perf / multi-GPU mode),
``use_graph=`` replays the step loop from a hipGraph, ``relax_fn=`` / ``mmff_backend=`` choose how the ``ref_mol``
relaxation branch runs (physics.py), ``conditioning=`` / ``return_conditioning=`` share one trunk run between calls.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import threading
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import ops
from .engine import Engine, off
from .packing import PackedWeights
from .params import param_shapes


_CAPTURE_LOCK = threading.Lock()
#: PD_CHECK_FINITE=1: EVERY sample_diffusion call checks its poses for inf / NaN (one tiny reduction + a host sync) and raises;
#: without it the check still runs on the first call of every (shape, schedule, weights version) and on every eager call
_CHECK_FINITE = os.environ.get("PD_CHECK_FINITE") == "1"
#: PD_BOUND_CHECK=0 switches the first-call check of the fp16-format operand bounds off (engine.check_dit_bounds)
_BOUND_CHECK = os.environ.get("PD_BOUND_CHECK", "1") != "0"
#: missing step units whose kind has already run on the shape are recorded first and launched like cached ones (sample_diffusion, "the
#: loop as UNITS"); 0: every missing unit runs eagerly and is recorded behind the loop (round 6's first form; A/B knob)
PIPELINED_CAPTURE = os.environ.get("PD_PIPELINED_CAPTURE", "1") != "0"


def _register(root: nn.Module, name: str, tensor: torch.Tensor):
    parts = name.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, nn.Module())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


def karras_noise_schedule(num_steps=200, sigma_data=16, s_max=160, s_min=4 * 10e-4, p=7):
    """reference models/model.py:117-129; same fp32 torch op order (host side, p=1000 amplifies rounding)"""
    idx = torch.arange(num_steps, dtype=torch.float32)
    t = sigma_data * (s_max ** (1 / p) + idx / (num_steps - 1) * (s_min ** (1 / p) - s_max ** (1 / p))) ** p
    return torch.cat([t, torch.zeros_like(t[:1])])


class PhysDock(nn.Module):
    supports_conditioning_reuse = True      # sample_diffusion(conditioning=, return_conditioning=): driver.redock shares the trunk between rounds

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.num_augmentation_sample = config.model.num_augmentation_sample
        self.sigma_data = config.sigma_data
        for name, shape in param_shapes(config).items():
            # weights are expected to come from load_state_dict; start from zeros like an un-initialised buffer
            _register(self, name, torch.zeros(shape))
        self._packed: Optional[PackedWeights] = None
        self._engine: Optional[Engine] = None
        self._packed_versions = None
        self._graphs = {}                 # whole step-loop hipGraphs per (shape, schedule, physics plan), LRU-ordered (dict insertion order)
        self._units = {}                  # per-step head / tail hipGraphs (sample_diffusion: "the loop as UNITS"), LRU-ordered
        self.max_cached_graphs = 64       # a screening stream over one receptor needs one graph per ligand SIZE (18 - 44 atoms in the demo)
        self.max_cached_units = 64 * 96   # 40 heads + 40 tails (+ gathers) per shape
        self.unit_captures = self.whole_captures = 0     # counters (tests, bench.py): unit graphs / whole-loop graphs captured so far
        self.last_unit_misses = self.last_head_misses = 0
        self._warm_kinds = set()          # (shape key, unit kind) pairs whose workspace buffers exist: their units may be recorded before they run
        self._capture_stream = None
        self.last_capture_ms = None       # host time of the most recent graph capture + instantiation (bench.py reports it)
        #: workspace buffers are cached per shape (288 GB of HBM make re-allocation pointless for a stream of
        #: same-size crops); when systems of many different sizes pass through, the cache is dropped beyond this size
        self.workspace_limit_bytes = 160 * 2 ** 30
        self.register_load_state_dict_post_hook(lambda m, k: m._invalidate())

    # ------------------------------------------------------------------ plumbing
    def _invalidate(self):
        self._packed = None
        self._engine = None
        self._stream_pool = None          # parallel.StreamPool.for_model: replicas hold copies of the old weights
        self._drop_graphs()

    def _drop_graphs(self):
        for g in self._graphs.values():
            for ex in g["exec"] or ():
                ops._lib.lib().pd_graph_destroy(ex)
        for u in self._units.values():
            ops._lib.lib().pd_graph_destroy(u["exec"])
        self._graphs = {}
        self._units = {}
        self._warm_kinds = set()

    def release_workspace(self):
        """free every cached activation buffer and captured step-loop graph (they are rebuilt on the next call)"""
        self._drop_graphs()
        if self._engine is not None:
            torch.cuda.synchronize(self._engine.device)
            self._engine.ws.bufs.clear()

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._invalidate()
        return r

    def _param_versions(self):
        """sum of the parameters' in-place modification counters: the packed weights, their fp16 / bf16 splits and every static
        magnitude bound derive from the values at pack time, so an in-place update (optimizer step, `p.data.mul_`, ...) must
        rebuild them - a stale bound would let the fp16 operand format overflow silently.  (Writes through `p.data` bypass the
        counters by PyTorch's design: call `_invalidate()` after those, or run with PD_CHECK_FINITE=1.)"""
        return sum(p._version for p in self.parameters()) + sum(b._version for b in self.buffers())

    def engine(self, device) -> Engine:
        if self._engine is not None and self._packed_versions != self._param_versions():
            self._invalidate()               # a weight was modified in place since the engine was built
        if self._engine is None or self._engine.device != device:
            if device.type != "cuda":
                raise RuntimeError("physdock_amd.PhysDock runs its sampler on an MI355X (HIP) device only; "
                                   "there is no CPU path (the CPU oracle lives in oracle/ and is test-only)")
            params = {k: v for k, v in self.state_dict().items()}
            if any(v.device != device for v in params.values()):
                raise RuntimeError("model parameters and batch must be on the same device")
            self._packed = PackedWeights(params, self.config)
            self._engine = Engine(self._packed, self.config, device)
            self._packed_versions = self._param_versions()
        return self._engine

    @staticmethod
    def _prepare_batch(batch):
        """Boundary bookkeeping (layout only): int32 token start offsets for the pooling kernels, fp32
        contiguity, and - for un-padded real systems whose token / atom counts are not multiples of 4
        (reference feature_loader.py:982-983) - padding with MASKED tokens / atoms.  The reference's own
        masks (a_mask / ap_mask / z_mask, tensor_utils.py:642-646) make padded entries inert: masked keys
        get weight exp(-1e9) = 0, triangle operands are multiplied by the mask, pooling is per token."""
        if "_tok_start" in batch:
            return batch
        b = dict(batch)
        A, T = b["ref_pos"].shape[0], b["target_feat"].shape[0]
        b["_A_real"], b["_T_real"] = A, T
        # atoms: to a multiple of 64 for systems of DiT size - sample-major [B * A, C] activations then consist of whole 64-row tiles for
        # every sample count, which is what the fp16-format rows / tile kernels and the fused transition take (a ragged 1 803-atom system
        # at 20 samples ran 65 % slower than the LARGER 2 048-atom crop through row-remainder launches); <= 3 % more (masked) rows
        pa = (-A) % (64 if A >= 512 else 4)
        pt = (-T) % 4
        # (padded atoms belong to no token segment - token_id_to_chunk_sizes, hence _tok_start, covers the real atoms only, so the pooling
        #  kernels never see them - and point at token 0 where an index is needed; adding a padded TOKEN for them, as this did, turned a
        #  256-token system into 260 tokens: sample-major token rows with a remainder in every projection)
        if pa or pt:
            import torch.nn.functional as F
            dev = b["ref_pos"].device

            def pad(k, dims, value=0):
                v = b[k]
                spec = []
                for d in range(v.dim() - 1, -1, -1):
                    spec += [0, dims.get(d, 0)]
                b[k] = F.pad(v, spec, value=value)
            for k in ("ref_feat", "ref_pos", "a_mask", "x_gt"):
                pad(k, {0: pa})
            if "x_exists" in b:
                pad("x_exists", {0: pa})
            pad("ap_mask", {0: pa, 1: pa})
            b["ref_space_uid"] = torch.cat([b["ref_space_uid"], b["ref_space_uid"].max() + 1 +
                                            torch.arange(pa, device=dev, dtype=b["ref_space_uid"].dtype)])
            b["atom_id_to_token_id"] = torch.cat([b["atom_id_to_token_id"],
                                                  torch.zeros(pa, device=dev, dtype=b["atom_id_to_token_id"].dtype)])
            b["token_id_to_chunk_sizes"] = torch.cat([b["token_id_to_chunk_sizes"],
                                                      torch.zeros(pt, device=dev, dtype=b["token_id_to_chunk_sizes"].dtype)])
            for k in ("target_feat", "key_res_feat", "pocket_res_feat", "is_ligand"):
                pad(k, {0: pt})
            for k in ("token_bonds_feature", "rel_tok_feat", "templ_feat", "z_mask"):
                pad(k, {0: pt, 1: pt})
            pad("msa_feat", {1: pt})
            for k in ("asym_id", "sym_id", "entity_id"):
                b[k] = torch.cat([b[k], (b[k].max() + 1).expand(pt).to(b[k].dtype)])
            b["residue_index"] = torch.cat([b["residue_index"], torch.zeros(pt, device=dev, dtype=b["residue_index"].dtype)])
        chunk = b["token_id_to_chunk_sizes"]
        ts = torch.zeros(chunk.shape[0] + 1, dtype=torch.int32, device=chunk.device)
        ts[1:] = torch.cumsum(chunk, 0).to(torch.int32)
        b["_tok_start"] = ts
        # tokens per block of the fused downscale + pool kernel (csrc/pool.hip): as many consecutive tokens as are sure to hold <= 64 atoms
        mc = int(chunk.max()) if chunk.numel() else 0
        b["_pool_tpb"] = min(32, 64 // mc) if 0 < mc <= 64 else 0
        for k in ("ref_feat", "ref_pos", "a_mask", "ap_mask", "target_feat", "key_res_feat", "pocket_res_feat",
                  "token_bonds_feature", "rel_tok_feat", "msa_feat", "templ_feat", "z_mask", "x_gt", "is_ligand", "t_mask"):
            v = b[k]
            if v.dtype != torch.float32 or not v.is_contiguous():
                b[k] = v.float().contiguous()
        for k, dt in (("ref_space_uid", torch.int64), ("atom_id_to_token_id", torch.int64), ("residue_index", torch.int64),
                      ("asym_id", torch.int32), ("sym_id", torch.int32), ("entity_id", torch.int32)):
            if b[k].dtype != dt or not b[k].is_contiguous():
                b[k] = b[k].to(dt).contiguous()
        return b

    # ------------------------------------------------------------------ reference API
    def karras_noise_schedule(self, num_steps=200, sigma_data=16, s_max=160, s_min=4 * 10e-4, p=7):
        return karras_noise_schedule(num_steps, sigma_data, s_max, s_min, p)

    def _step_plan(self, steps, gamma_0, gamma_min, step_scale_eta, ode_step_scale_eta, mmff_gamma_0_factor,
                   align_ref_pos, p):
        """Host-side resolution of every per-step scalar and branch of reference model.py:211-281
        (all t-dependent branches depend only on the schedule, so none needs a device sync)."""
        sig = karras_noise_schedule(steps, p=p)
        sd = float(self.sigma_data)
        plan = []
        for i in range(steps):
            t_cur, t_next = sig[i], sig[i + 1]
            noisy = bool(t_cur > gamma_min)
            t_hat = t_cur * (gamma_0 + 1) if noisy else t_cur
            sdev = torch.sqrt(t_hat ** 2 - t_cur ** 2) if noisy else torch.zeros(())
            c_in = 1 / torch.sqrt(t_hat ** 2 + sd ** 2)
            c_noise = torch.log(t_hat / sd) / 4.0
            plan.append(dict(
                t_hat=float(t_hat), sdev=float(sdev), noisy=noisy, c_in=float(c_in), tau=float(t_hat * c_noise),
                c_skip=float(sd ** 2 / (sd ** 2 + t_hat ** 2)), c_out=float(sd * t_hat / torch.sqrt(sd ** 2 + t_hat ** 2)),
                dt=float(t_next - t_hat), eta=float(step_scale_eta if noisy else ode_step_scale_eta),
                align=bool(align_ref_pos and t_cur > gamma_min * mmff_gamma_0_factor),
                mmff=bool(t_cur <= gamma_min * mmff_gamma_0_factor)))
        return sig, plan

    @torch.no_grad()
    def sample_diffusion(
            self,
            batch: Dict[str, torch.Tensor],
            num_sample: int = 5,
            steps: int = 200,
            gamma_0: float = 0.8,
            gamma_min: float = 1.0,
            noise_scale_lambda: float = 1.003,
            step_scale_eta: float = 1.5,
            ode_step_scale_eta=1.0,
            ref_mol=None,
            ref_mol_poses=None,
            use_ref_mol_poses=False,
            mmff_gamma_0_factor=1.0,
            mmff_iters=5,
            align_ref_pos=True,
            karras_noise_schedule_power=7,
            *,
            noise=None,
            seed: int = 0,
            sample_offset: int = 0,
            use_graph: bool = True,
            conditioning=None,
            return_conditioning: bool = False,
            relax_fn=None,
            mmff_backend: str = "auto",
    ) -> torch.Tensor:
        """reference models/model.py:157-282.  Returns x_next [num_sample, A, 3] on the batch device.

        `ref_mol` switches on the relaxation branch (model.py:252-261) exactly as in the reference; how the relaxed
        ligand is produced is described in physics.py (device MMFF94 kernel, host RDKit, or an injected `relax_fn` with
        the signature of the reference's `get_next_step_pos`)."""
        from . import physics
        device = batch["x_gt"].device
        eng = self.engine(device)
        if eng.ws.nbytes() > self.workspace_limit_bytes:
            self.release_workspace()
        batch = self._prepare_batch(batch)
        L = ops._lib.init()
        ws = eng.ws
        B = num_sample
        A = batch["ref_pos"].shape[0]            # padded atom count
        A_real = batch["_A_real"]
        sp = ops.stream()
        if B <= 0:
            return torch.empty(0, A_real, 3, device=device)

        relaxer = physics.resolve_relaxer(ref_mol, relax_fn, mmff_backend)
        if ref_mol_poses is None and use_ref_mol_poses:      # model.py:188-203: ETKDG conformers of the reference molecule
            if not (ref_mol is not None and physics.have_rdkit() and physics._is_rdkit_mol(ref_mol)):
                raise RuntimeError("use_ref_mol_poses=True without ref_mol_poses needs an RDKit ref_mol (and RDKit) to embed "
                                   "conformers (reference model.py:188-203); pass ref_mol_poses [C,L,3]")
            n_lig_real = int((batch["is_ligand"][batch["atom_id_to_token_id"]][:A_real] > 0).sum())
            ref_mol_poses = physics.rdkit_ref_mol_poses(ref_mol, 512)[:, :n_lig_real]

        sig, plan = self._step_plan(steps, gamma_0, gamma_min, step_scale_eta, ode_step_scale_eta, mmff_gamma_0_factor,
                                    align_ref_pos, karras_noise_schedule_power)
        for p in plan:                                        # the elif of model.py:252 is only reachable with a molecule
            p["mmff"] = bool(p["mmff"] and not p["align"] and relaxer.kind != "none")

        # ---- per-call setup: conditioning trunk, hoisted biases, AdaLN tables
        a, ap, s, z = conditioning if conditioning is not None else eng.conditioning(batch)
        tau = ws.get("tau", steps)
        tau.copy_(torch.tensor([p["tau"] for p in plan], dtype=torch.float32))
        prep = eng.prepare_dit(a, ap, s, z, batch, tau, B=B)
        # First call of this engine (= these weights) on this schedule: check the fp16-format operand bounds against what the
        # denoiser's launches really see (engine.check_dit_bounds) - a violated bound raises, a uselessly loose one moves the
        # family to bf16 x 6 and the call is re-prepared.  Once per (weights, schedule); replays and later systems pay nothing.
        ck = (steps, float(sig[0]), float(sig[-2]), float(gamma_0), float(gamma_min))
        if _BOUND_CHECK and ops.F16_GEMM and ops.SPLIT_GEMM and ck not in eng._bounds_checked:
            if ck in eng._bounds_failed:         # a violated bound stays an exception on EVERY later call of this engine + schedule
                raise eng._bounds_failed[ck]
            try:
                switched = eng.check_dit_bounds(batch, a, s, prep, plan, B, float(self.sigma_data))
            except FloatingPointError as e:
                eng._bounds_failed[ck] = e
                raise
            eng._bounds_checked.add(ck)          # only a check that RETURNED counts
            if switched:
                self._drop_graphs()
                prep = eng.prepare_dit(a, ap, s, z, batch, tau, B=B)

        # Everything the step loop reads is staged in workspace buffers: a captured hipGraph replays raw addresses, so
        # no caller-owned or per-call temporary tensor may be referenced from inside the loop (a / s included: a graph
        # captured with the trunk's own outputs must stay valid when a later call passes conditioning=, and vice versa).
        def staged(name, t):
            buf = ws.get("loop:" + name, *t.shape, dtype=t.dtype)
            buf.copy_(t)
            return buf
        cond_out = (a, ap, s, z)
        a, s = staged("a", a), staged("s", s)
        batch = dict(batch)
        for k in ("a_mask", "atom_id_to_token_id", "_tok_start", "ref_pos"):
            batch[k] = staged(k, batch[k])
        lig_flag = batch["is_ligand"][batch["atom_id_to_token_id"]]          # index gather on metadata, once per call
        if A_real < A:
            lig_flag = lig_flag.clone()
            lig_flag[A_real:] = 0                                            # padded atoms carry a placeholder token index
        lig_w = ws.get("lig_w", A)
        lig_w.copy_(batch["a_mask"] * lig_flag)
        any_align = any(p["align"] for p in plan)
        any_mmff = any(p["mmff"] for p in plan)
        ref_dist = poses = lig_idx = atom_slot = tm_eps = None
        n_conf = n_lig = 0
        if (any_align and ref_mol_poses is not None) or any_mmff:
            lig_idx = torch.nonzero(lig_flag > 0).flatten().to(torch.int32)
            n_lig = int(lig_idx.numel())
            lig_idx = staged("lig_idx", lig_idx)
        if any_align:
            bref = ws.get("batch_ref_pos", B, A, 3)
            if ref_mol_poses is not None and ref_mol_poses.shape[1] == n_lig:   # mismatch: reference silently keeps ref_pos (model.py:229-243)
                poses = staged("poses", ref_mol_poses.to(device).float())
                n_conf = poses.shape[0]
                ref_dist = ws.get("ref_dist", n_conf, n_lig, n_lig)
                tm_eps = ws.get("tm_eps", B, n_conf)          # eps[b, c] scratch: lets the matching run conformer-parallel
                ops.check(L.pd_pose_dist(ops.ptr(poses), ops.ptr(ref_dist), n_conf, n_lig, sp), "pose_dist")
        if any_mmff:
            if n_lig == 0:
                raise ValueError("ref_mol given but the crop has no ligand atoms")
            x_ref = ws.get("x_ref", B, A, 3)
            if relaxer.kind == "host":
                slot = torch.full((A,), -1, dtype=torch.int32, device=device)
                slot[lig_idx.long()] = torch.arange(n_lig, dtype=torch.int32, device=device)
                atom_slot = staged("atom_slot", slot)
                lig_in = ws.get("lig_in", B, n_lig, 3)
                lig_out = ws.get("lig_out", B, n_lig, 3)
            else:
                mm = relaxer.terms.device_tables(device, n_lig)
                mm_ws = ws.get("mmff_ws", relaxer.terms.workspace_numel(B), dtype=torch.float64)

        # ---- random numbers: parity mode copies the caller's draws into fixed buffers
        x_a = ws.get("x_a", B, A, 3)
        x_hat = ws.get("x_hat", B, A, 3)
        x_den = ws.get("x_den", B, A, 3)
        x_proj = ws.get("x_proj", B, A, 3)
        n_noisy = sum(p["noisy"] for p in plan)
        if noise is not None:
            n_init = ws.get("n_init", B, A, 3, zero=True); n_init[:, :A_real].copy_(noise["init"])
            n_rot = ws.get("n_rot", steps, 4, B); n_rot.copy_(noise["rot_u"])
            n_tr = ws.get("n_trans", steps, B, 3); n_tr.copy_(noise["trans"])
            n_dif = ws.get("n_diffuse", max(n_noisy, 1), B, A, 3, zero=True)
            if n_noisy:
                n_dif[:n_noisy, :, :A_real].copy_(noise["diffuse"])
            seed_buf = None
        else:
            seed_buf = ws.get("seed", 1, dtype=torch.int64)
            seed_buf.fill_(int(seed))
        k_noisy = [sum(q["noisy"] for q in plan[:i]) for i in range(steps)]

        def step_head(i):
            """model.py:212-221: augmentation, noise injection, denoiser (the ligand read-out of a host relaxation: gather_fn below)"""
            p = plan[i]
            sp_ = ops.stream()
            if i == 0 and any_align:     # `batch_ref_pos = ref_pos[None].repeat(...)` (model.py:183): part of the replayed loop
                bref.copy_(batch["ref_pos"][None].expand(B, A, 3))
            if noise is not None:
                ru, tr = off(n_rot, i * 4 * B), off(n_tr, i * B * 3)
                nz = off(n_dif, k_noisy[i] * B * A * 3) if p["noisy"] else None
                sd_ptr = None
                src, x_scale = (n_init, float(sig[0])) if i == 0 else (x_a, 1.0)
            else:
                ru = tr = nz = None
                sd_ptr = ops.ptr(seed_buf)
                src, x_scale = x_a, 1.0
                if i == 0:
                    ops.check(L.pd_init_noise(ops.ptr(x_a), sd_ptr, sample_offset, float(sig[0]), B, A, sp_), "init_noise")
            ops.check(L.pd_augment(ops.ptr(src), x_scale, ops.ptr(batch["a_mask"]), ru, tr, nz, float(noise_scale_lambda),
                                   p["sdev"], sd_ptr, i, sample_offset, ops.ptr(x_hat), B, A, sp_), "augment")
            eng.af3_dit(batch, x_hat, x_den, a, s, prep, B, p, row=i)

        def step_tail(i):
            """model.py:223-281: physics correction and Euler update"""
            p = plan[i]
            sp_ = ops.stream()
            if p["align"]:
                if poses is not None:
                    ops.check(L.pd_template_match(ops.ptr(x_den), ops.ptr(lig_idx), ops.ptr(ref_dist), ops.ptr(poses),
                                                  ops.ptr(bref), ops.ptr(tm_eps), None, B, A, n_lig, n_conf, sp_), "template_match")
                target, tstride = bref, A * 3
            elif p["mmff"]:
                if relaxer.kind == "host":
                    ops.check(L.pd_ligand_scatter(ops.ptr(x_ref), ops.ptr(x_den), ops.ptr(lig_out), ops.ptr(atom_slot),
                                                  B, A, n_lig, sp_), "ligand_scatter")
                else:
                    relaxer.terms.launch_relax(mm, x_den, lig_idx, x_ref, mm_ws, B, A, int(mmff_iters), sp_)
                target, tstride = x_ref, A * 3
            else:
                ops.check(L.pd_euler(ops.ptr(x_hat), ops.ptr(x_den), None, None, p["t_hat"], p["eta"], p["dt"],
                                     ops.ptr(x_a), B, A, sp_), "euler")
                return
            ops.check(L.pd_kabsch_align(ops.ptr(x_den), ops.ptr(batch["a_mask"]), ops.ptr(target), tstride, ops.ptr(lig_w),
                                        ops.ptr(x_proj), B, A, sp_), "kabsch")
            ops.check(L.pd_euler(ops.ptr(x_hat), ops.ptr(x_den), ops.ptr(x_proj), ops.ptr(lig_w), p["t_hat"], p["eta"],
                                 p["dt"], ops.ptr(x_a), B, A, sp_), "euler")

        # ---- the loop as UNITS (round 6): one unit = the launches of one step's head (augmentation + denoiser: ~115 kernels, the
        #      expensive part) or of one step's tail (physics correction + Euler update: 1 - 4 kernels).  A head depends on the shape,
        #      the schedule and the step index only; what the drivers change between calls - the physics threshold
        #      (`mmff_gamma_0_factor` x 1.15 / x 0.7 per round, redocking.py:318-322), the template pool size, the relaxation - lives in
        #      the tails.  Every unit is captured as its own hipGraph, keyed by exactly what its launches depend on, so a call with a new
        #      threshold or pool replays all 40 heads and the unchanged tails and runs (then captures) only the few tails that changed.
        #      A whole schedule seen twice is additionally captured as ONE graph (segments around a host relaxation), as before.
        sched_id = (steps, float(sig[0]), float(sig[-2]), float(gamma_0), float(gamma_min), float(karras_noise_schedule_power))
        common = (B, A, batch["target_feat"].shape[0], batch["_A_real"], batch["_T_real"], sched_id, noise is not None,
                  float(noise_scale_lambda), sample_offset)
        host_relax_steps = relaxer.kind == "host"
        pool_sig = poses is not None and (n_conf, n_lig)
        relax_sig = (relaxer.kind, relaxer.kind == "device" and relaxer.terms.signature(), int(mmff_iters), n_lig)

        def gather_fn(i):
            ops.check(L.pd_ligand_gather(ops.ptr(x_den), ops.ptr(lig_idx), ops.ptr(lig_in), B, A, n_lig, ops.stream()), "ligand_gather")

        units = []                   # (key, [(fn, i), ...], host break after this unit?)
        for i, p in enumerate(plan):
            units.append(((common, "H", i, p["noisy"], p["sdev"], p["t_hat"], i == 0 and any_align), [(step_head, i)], False))
            if p["mmff"] and host_relax_steps:
                units.append(((common, "G", i, n_lig), [(gather_fn, i)], True))
            kind = ("align", pool_sig) if p["align"] else (("mmff", relax_sig) if p["mmff"] else ("plain",))
            units.append(((common, "T", i, p["t_hat"], p["eta"], p["dt"], kind), [(step_tail, i)], False))
        segments, cur = [], []
        for ukey, fns, brk in units:
            cur.extend(fns)
            if brk:
                segments.append(cur)
                cur = []
        segments.append(cur)

        def run_fns(fns):
            for fn, i in fns:
                fn(i)

        def host_relax():
            lig_out.copy_(relaxer(lig_in.clone(), int(mmff_iters)).to(device=device, dtype=torch.float32))

        def capture(fn_lists, sync=True):
            """record (not run) each launch list as one hipGraph; one capture at a time per process: objects driven from several host
            threads (parallel.StreamPool) replay concurrently, but two overlapping captures make unrelated launches of the other
            thread fail.  sync=False (units recorded in the middle of a call, below): no device synchronisation around the recording -
            nothing is enqueued on the recording stream, and the launch stream keeps executing the units issued before"""
            import time as _time
            with _CAPTURE_LOCK:
                if sync:
                    torch.cuda.synchronize()
                t_cap = _time.perf_counter()
                execs = []
                cap = self._capture_stream if getattr(self, "_capture_stream", None) is not None else torch.cuda.Stream()
                self._capture_stream = cap
                with torch.cuda.stream(cap):
                    for fns in fn_lists:
                        ops.check(L.pd_graph_begin(ops.stream()), "graph_begin")
                        run_fns(fns)
                        ex = C.c_void_p()
                        ops.check(L.pd_graph_end(ops.stream(), C.byref(ex)), "graph_end")
                        execs.append(ex)
                if sync:
                    torch.cuda.synchronize()
                    self.last_capture_ms = 1e3 * (_time.perf_counter() - t_cap)
                else:
                    self.last_capture_ms = (self.last_capture_ms or 0.0) + 1e3 * (_time.perf_counter() - t_cap)
            return execs

        fresh = not use_graph                 # did any launch of this call run eagerly (= was not replayed from a checked capture)?
        entry = None
        if use_graph:
            # (the REAL atom / token counts are launch arguments - reduction bounds - of the captured kernels: two systems that pad
            #  to the same shape must not share a graph)
            key = (common, tuple(u[0] for u in units))
            entry = self._graphs.get(key)
            if entry is not None:
                self._graphs[key] = self._graphs.pop(key)          # LRU order
        if entry is not None and entry["exec"] is not None:
            for k, g in enumerate(entry["exec"]):
                ops.check(L.pd_graph_launch(g, sp), "graph_launch")
                if k + 1 < len(segments):
                    host_relax()
        elif not use_graph:
            for k, seg in enumerate(segments):
                run_fns(seg)
                if k + 1 < len(segments):
                    host_relax()
        else:
            # replay the units that exist; run the others eagerly - they produce this call's result AND allocate their workspace
            # buffers, so the capture below (which only records) needs no launch of its own
            # A missing unit whose KIND (head / gather / tail of one physics branch) has already run once on this shape - earlier in this
            # call or in an earlier one - finds every workspace buffer it touches allocated: it is RECORDED first (no execution: ~12 us
            # per launch on the host against ~35 us for an eager launch) and then launched like a cached one, while the GPU still works on
            # the units issued before.  Only the first unit of a kind runs eagerly (it allocates).  First call of a new shape at 20 samples:
            # 225 -> 179 ms against 164 ms cached; at 64 samples 392 -> 334 ms against 333 ms (profiles/r06_new_shape_call.txt).
            missing, recorded = [], 0
            terms = relaxer.terms if relaxer.kind == "device" else None
            if PIPELINED_CAPTURE:
                self.last_capture_ms = 0.0
            for ukey, fns, brk in units:
                u = self._units.get(ukey)
                kind = (common, ukey[1], ukey[-1][0] if ukey[1] == "T" else None)
                if u is None and PIPELINED_CAPTURE and kind in self._warm_kinds:
                    ex = capture([fns], sync=False)[0]
                    u = self._units[ukey] = {"exec": ex, "terms": terms if ukey[1] == "T" and ukey[-1][0] == "mmff" else None}
                    self.unit_captures += 1
                    recorded += 1
                    fresh = True                                   # (its first execution: the finite check below applies)
                if u is not None:
                    self._units[ukey] = self._units.pop(ukey)      # LRU order
                    ops.check(L.pd_graph_launch(u["exec"], sp), "graph_launch")
                else:
                    fresh = True
                    run_fns(fns)
                    missing.append((ukey, fns))
                    self._warm_kinds.add(kind)
                if brk:
                    host_relax()
            self.last_unit_misses = len(missing) + recorded
            self.last_head_misses = sum(1 for k, _ in missing if k[1] == "H")
            if missing:
                # the captured launches hold raw device addresses: a unit keeps the MMFF table object whose tables it captured alive
                # (a later call with an EQUAL table - same signature, e.g. rebuilt from the same RDKit molecule - replays against
                # these tables, not against its own freshly built and soon freed ones)
                for (ukey, _), ex in zip(missing, capture([fns for _, fns in missing], sync=not PIPELINED_CAPTURE)):
                    self._units[ukey] = {"exec": ex, "terms": terms if ukey[1] == "T" and ukey[-1][0] == "mmff" else None}
                self.unit_captures += len(missing)
                while len(self._units) > self.max_cached_units:
                    L.pd_graph_destroy(self._units.pop(next(iter(self._units)))["exec"])
            if entry is None:
                entry = self._graphs[key] = {"exec": None, "terms": relaxer.terms if relaxer.kind == "device" else None}
            else:
                # the second call of the same schedule on the same shape: the whole N-step loop as ONE graph from here on
                # (recorded without a device synchronisation while the GPU executes the units launched above: a synchronising capture
                #  stalled every stream of the device - the other replica of a StreamPool too: two systems at a time fell from 3.1 to 2.7
                #  ligands/s when the promotion landed in the measured region)
                entry["exec"] = capture(segments, sync=not PIPELINED_CAPTURE)
                entry["terms"] = relaxer.terms if relaxer.kind == "device" else None      # the tables THIS capture recorded
                self.whole_captures += 1
            while len(self._graphs) > self.max_cached_graphs:
                old = self._graphs.pop(next(iter(self._graphs)))
                for ex in old["exec"] or ():
                    L.pd_graph_destroy(ex)
        out = x_a[:, :A_real].clone()
        # Finite check: ALWAYS on the first call of a (shape, schedule, weights version) - the eager pass that precedes a graph capture,
        # and every eager call - and on every call with PD_CHECK_FINITE=1 (one tiny reduction + a host sync; replays stay sync-free).
        # A violated magnitude bound (weights changed behind the engine's back through `.data`, a caller-supplied bound that does not
        # hold) overflows the fp16 operand format to inf / NaN: that must be an exception, not a pose.
        if (_CHECK_FINITE or fresh) and not bool(torch.isfinite(out).all()):
            if use_graph:
                self._drop_graphs()          # the loop captured from this pass would replay the same overflow
            raise FloatingPointError("sample_diffusion produced non-finite coordinates: an fp16-format operand bound was violated "
                                     "(rebuild the engine after changing weights; ops.F16_GEMM / F16_ATTN = False to confirm)")
        if return_conditioning:
            return out, tuple(t.clone() for t in cond_out)
        return out

    @torch.no_grad()
    def forward(self, batch):
        """reference models/model.py:99-115 (training-time forward; kept for API completeness):
        conditioning -> 48 noised copies (per-sample noise level) -> denoiser -> distogram logits."""
        device = batch["x_gt"].device
        eng = self.engine(device)
        if eng.ws.nbytes() > self.workspace_limit_bytes:
            self.release_workspace()
        batch = self._prepare_batch(batch)
        L = ops._lib.init()
        ws = eng.ws
        B = self.num_augmentation_sample
        A, T = batch["ref_pos"].shape[0], batch["target_feat"].shape[0]
        A_real, T_real = batch["_A_real"], batch["_T_real"]
        sd = float(self.sigma_data)
        a, ap, s, z = eng.conditioning(batch)
        # augmentation_diffuse (model.py:87-97): per-sample noise levels; RNG on the host generator
        t_hat = (torch.exp(torch.normal(0, 1, (B,)) * 1.5 - 1.2) * sd).to(device)
        x_noisy = ws.get("fw_xn", B, A, 3)
        x_noisy.copy_(batch["x_gt"][None] + torch.normal(0, 1, (B, A, 3)).to(device) * t_hat[:, None, None])
        x_hat = ws.get("fw_xhat", B, A, 3)
        rot = torch.rand(4, B).to(device).contiguous()
        tr = torch.normal(0, 1, (B, 3)).to(device).contiguous()
        ops.check(L.pd_augment(ops.ptr(x_noisy), 1.0, ops.ptr(batch.get("x_exists", batch["a_mask"])), ops.ptr(rot), ops.ptr(tr),
                               None, 1.0, 0.0, None, 0, 0, ops.ptr(x_hat), B, A, ops.stream()), "augment")
        th = t_hat.cpu()
        scal = {"c_in": (1 / torch.sqrt(th ** 2 + sd ** 2)).to(device), "c_skip": (sd ** 2 / (sd ** 2 + th ** 2)).to(device),
                "c_out": (sd * th / torch.sqrt(sd ** 2 + th ** 2)).to(device)}
        tau = (th * (torch.log(th / sd) / 4.0)).to(device)
        prep = eng.prepare_dit(a, ap, s, z, batch, tau, per_sample=True, B=B)
        x_den = ws.get("fw_xden", B, A, 3)
        eng.af3_dit(batch, x_hat, x_den, a, s, prep, B, scal, row=0, per_sample=True)
        pd = eng.lin(z, "linear_distogram", T * T).reshape(T, T, -1)[:T_real, :T_real]
        return {"x_denoised": x_den[:, :A_real].clone(), "x_hat": x_hat[:, :A_real].clone(), "t_hat": t_hat,
                "p_distogram": pd + pd.transpose(0, 1)}


def weighted_rigid_align(x_pred, x_gt, weights):
    """reference utils/tensor_utils.py:724-778: returns x_gt moved onto x_pred (second argument moves)."""
    L = ops._lib.init()
    xp = x_pred.float().contiguous()
    B, A = xp.shape[0], xp.shape[1]
    xg = x_gt.float().contiguous()
    out = torch.empty_like(xp)
    ops.check(L.pd_kabsch_align(ops.ptr(xp), None, ops.ptr(xg), 0 if xg.dim() == 2 else A * 3,
                                ops.ptr(weights.float().contiguous()), ops.ptr(out), B, A, ops.stream()), "kabsch")
    return out
