"""CPU oracle for the PhysDock sampler hot path  --  TEST INFRASTRUCTURE ONLY.

A plain PyTorch (CPU, fp32) restatement of ``PhysDock.sample_diffusion`` and
everything under it, written as pure functions over a flat state dict ``P`` that
uses the reference's parameter names.  It is the checker for the HIP path:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it; the product package ``physdock_amd`` never does.

Pinning: ``tests/test_oracle_golden.py`` checks every function here against the
golden vectors in ``tests/golden/`` that ``tools/make_golden.py`` captured by
importing the reference itself in the build container (the reference ships no
tests or golden vectors of its own).  NOT pinned ("parity unpinned"):
the RDKit MMFF94 relaxation sub-step (reference model.py:26-52,252-261) - RDKit
is not available, so that branch is not restated here.

Each function cites the reference lines it follows (paths relative to the
reference checkout).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
HEAD = 32


# --------------------------------------------------------------------------- primitives
def linear(P, name, x):
    """PhysDock/models/primitives/linear.py:146-161 (fp32 path = F.linear)."""
    return F.linear(x, P[name + ".weight"], P.get(name + ".bias"))


def rms_norm(P, name, x, eps):
    """primitives/rms_norm.py:14-19."""
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * P[name + ".weight"]


def layer_norm(P, name, x, eps):
    """primitives/layer_norm.py:5 (nn.LayerNorm, affine)."""
    return F.layer_norm(x, x.shape[-1:], P[name + ".weight"], P[name + ".bias"], eps)


def ada_ln_zero(P, name, x, t, eps):
    """primitives/adaptive_layer_norm_zero.py:18-21; t:[B,256], x:[B,N,C] -> (x_mod, gate[B,1,C])."""
    mod = linear(P, name + ".linear", F.silu(t[..., None, :]))
    shift, scale, gate = mod.chunk(3, dim=-1)
    xn = F.layer_norm(x, x.shape[-1:], None, None, eps)
    return xn * (1 + scale) + shift, gate


def feed_forward(P, name, x):
    """primitives/feed_forward.py:30-31 (SwiGLU, no biases)."""
    return linear(P, name + ".w2", F.silu(linear(P, name + ".w1", x)) * linear(P, name + ".w3", x))


def transition(P, name, x, eps):
    """primitives/transitions.py:15-18."""
    return feed_forward(P, name + ".feed_forward", rms_norm(P, name + ".ffn_norm", x, eps))


def dit_transition(P, name, x, t, eps):
    """primitives/transitions.py:27-30."""
    xn, gate = ada_ln_zero(P, name + ".ffn_norm", x, t, eps)
    return feed_forward(P, name + ".feed_forward", xn) * gate


def timestep_embeddings(P, name, tau):
    """primitives/timestep_embeddings.py:35-86,156-166: cos|sin(256), shift 0, then MLP."""
    half = 128
    freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half).to(tau.dtype)
    arg = tau[:, None] * freq[None]
    emb = torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)
    h = F.silu(linear(P, name + ".timestep_embedder.linear_1", emb))
    return linear(P, name + ".timestep_embedder.linear_2", h)


def attn_mask_bias(mask, inf):
    """utils/tensor_utils.py:642-646: 0 where mask != 0, -inf_cfg elsewhere."""
    return torch.where(mask == 0, torch.full_like(mask, -inf), torch.zeros_like(mask))


def _heads(x, lead):
    """[..., N, H*32] -> [..., H, N, 32]"""
    return x.reshape(*x.shape[:-1], -1, HEAD).transpose(-2, -3)


def _sdpa(q, k, v, bias):
    s = q @ k.transpose(-1, -2) / math.sqrt(HEAD)
    if bias is not None:
        s = s + bias
    return torch.softmax(s, dim=-1) @ v


def _merge(o):
    o = o.transpose(-2, -3)
    return o.reshape(*o.shape[:-2], -1)


def attention_pair_bias(P, name, s, z, z_mask, inf, eps, norm_name="norm_s"):
    """primitives/attentions.py:32-53 (single) and :76-97 (MSA rows: s has a leading row dim).
    Raw (un-squashed) gate."""
    sn = rms_norm(P, f"{name}.{norm_name}", s, eps)
    zn = rms_norm(P, name + ".norm_z", z, eps)
    q, k, v = (_heads(linear(P, f"{name}.linear_{c}", sn), 0) for c in "qkv")
    g = linear(P, name + ".linear_g", sn)
    bias = linear(P, name + ".linear_z", zn).permute(2, 0, 1) + attn_mask_bias(z_mask, inf)[None]
    o = _merge(_sdpa(q, k, v, bias))
    return linear(P, name + ".linear_o", o) * g


def msa_column_attention(P, name, m, eps):
    """primitives/attentions.py:117-136: attention along the MSA-row axis, no mask, no bias."""
    mt = m.transpose(-2, -3)
    mn = rms_norm(P, name + ".norm_m", mt, eps)
    q, k, v = (_heads(linear(P, f"{name}.linear_{c}", mn), 0) for c in "qkv")
    g = linear(P, name + ".linear_g", mn)
    o = _merge(_sdpa(q, k, v, None))
    return (linear(P, name + ".linear_o", o) * g).transpose(-2, -3)


def outer_product_mean(P, name, m, eps):
    """primitives/outer_product_mean.py:23-31: SUM over MSA rows, norm after projection."""
    mn = rms_norm(P, name + ".norm_in", m, eps)
    q = linear(P, name + ".linear_q", mn)
    k = linear(P, name + ".linear_k", mn)
    T = m.shape[-2]
    outer = torch.einsum("bic,bjd->ijcd", q, k).reshape(T, T, -1)
    return rms_norm(P, name + ".norm_out", linear(P, name + ".linear_o", outer), eps)


def triangle_update(P, name, z, z_mask, eps, transpose):
    """primitives/attentions.py:157-171."""
    if transpose:
        z = z.transpose(-2, -3)
    zn = rms_norm(P, name + ".norm_in", z, eps)
    q = linear(P, name + ".linear_qx", zn) * torch.sigmoid(linear(P, name + ".linear_q", zn)) * z_mask[..., None]
    k = linear(P, name + ".linear_kx", zn) * torch.sigmoid(linear(P, name + ".linear_k", zn)) * z_mask[..., None]
    g = torch.sigmoid(linear(P, name + ".linear_g", zn))
    o = torch.einsum("ijc,Ijc->iIc", q, k)
    o = linear(P, name + ".linear_z", rms_norm(P, name + ".norm_out", o, eps)) * g
    return o.transpose(-2, -3) if transpose else o


def triangle_attention(P, name, z, z_mask, inf, eps, transpose):
    """primitives/attentions.py:194-217: rows are the batch; bias shared by all rows; raw gate."""
    if transpose:
        z = z.transpose(-2, -3)
    zn = rms_norm(P, name + ".norm", z, eps)
    q, k, v = (_heads(linear(P, f"{name}.linear_{c}", zn), 0) for c in "qkv")
    g = linear(P, name + ".linear_g", zn)
    bias = linear(P, name + ".linear_z", zn).permute(2, 0, 1)[None] + attn_mask_bias(z_mask, inf)[None, None]
    o = _merge(_sdpa(q, k, v, bias))
    o = linear(P, name + ".linear_o", o) * g
    return o.transpose(-2, -3) if transpose else o


def dit_attention(P, name, bs, z, t, z_mask, inf, eps):
    """primitives/attentions.py:241-265 (norm_z is nn.LayerNorm with its default eps 1e-5, :232)."""
    xn, gate = ada_ln_zero(P, name + ".norm_s", bs, t, eps)
    zn = layer_norm(P, name + ".norm_z", z, 1e-5)
    q, k, v = (_heads(linear(P, f"{name}.linear_{c}", xn), 0) for c in "qkv")
    q = rms_norm(P, name + ".norm_q", q, eps)
    k = rms_norm(P, name + ".norm_k", k, eps)
    bias = linear(P, name + ".linear_z", zn).permute(2, 0, 1)[None] + attn_mask_bias(z_mask, inf)[None, None]
    o = _merge(_sdpa(q, k, v, bias))
    return linear(P, name + ".linear_o", o) * gate


# --------------------------------------------------------------------------- blocks
def triangle_block(P, name, z, z_mask, inf, eps):
    """layers/transformers.py:48-54 (also the z half of Evoformer/Pairformer blocks)."""
    z = z + triangle_update(P, name + ".triangle_row_update", z, z_mask, eps, False)
    z = z + triangle_update(P, name + ".triangle_col_update", z, z_mask, eps, True)
    z = z + triangle_attention(P, name + ".triangle_row_attention", z, z_mask, inf, eps, False)
    z = z + triangle_attention(P, name + ".triangle_col_attention", z, z_mask, inf, eps, True)
    z = z + transition(P, name + ".pair_transition", z, eps)
    return z


def evoformer_block(P, name, m, z, z_mask, inf, eps):
    """layers/transformers.py:85-95."""
    m = m + attention_pair_bias(P, name + ".msa_row_attention", m, z, z_mask, inf, eps, "norm_m")
    m = m + msa_column_attention(P, name + ".msa_col_attention", m, eps)
    m = m + transition(P, name + ".msa_transition", m, eps)
    z = z + outer_product_mean(P, name + ".opm", m, eps)
    return m, triangle_block(P, name, z, z_mask, inf, eps)


def pairformer_block(P, name, s, z, z_mask, inf, eps):
    """layers/transformers.py:124-132."""
    z = triangle_block(P, name, z, z_mask, inf, eps)
    s = s + attention_pair_bias(P, name + ".attention", s, z, z_mask, inf, eps)
    s = s + transition(P, name + ".transition", s, eps)
    return s, z


def atom_block(P, name, a, ap, ap_mask, inf, eps):
    """layers/transformers.py:19-22."""
    a = a + attention_pair_bias(P, name + ".attention", a, ap, ap_mask, inf, eps)
    return a + transition(P, name + ".transition", a, eps)


def dit_block(P, name, bs, z, t, z_mask, inf, eps):
    """layers/transformers.py:155-159."""
    bs = bs + dit_attention(P, name + ".attention", bs, z, t, z_mask, inf, eps)
    return bs + dit_transition(P, name + ".transition", bs, t, eps)


def _nblocks(P, prefix):
    n = 0
    while any(k.startswith(f"{prefix}.{n}.") for k in P):
        n += 1
    return n


# --------------------------------------------------------------------------- conditioning trunk
def segment_mean_pool(u, chunk_sizes):
    """layers/transformers.py:205-212 / diffusion_conditioning.py:168-176: cumsum + diff, / (n+1e-3)."""
    cs = torch.cumsum(u, dim=-2)
    inds = torch.cumsum(chunk_sizes, dim=-1) - 1
    val = cs[..., inds, :]
    x = torch.cat([val[..., 0:1, :], torch.diff(val, dim=-2)], dim=-2)
    return x / (chunk_sizes[:, None] + 1e-3)


def one_hot_nearest(x, bins):
    """utils/tensor_utils.py:78-82."""
    am = torch.argmin(torch.abs(x[..., None] - bins), dim=-1)
    return F.one_hot(am, num_classes=len(bins)).float()            # cast to the weight dtype by the caller


def rel_pos_features(batch):
    """layers/diffusion_conditioning.py:65-92 -> [T,T,115]."""
    asym, sym, ent = batch["asym_id"], batch["sym_id"], batch["entity_id"]
    res = batch["residue_index"]
    r_max, s_max = 32, 2
    chain_same = asym[:, None] == asym[None, :]
    ent_same = ent[:, None] == ent[None, :]
    d_res = torch.clamp(res[:, None] - res[None, :] + r_max, 0, 2 * r_max)
    d_res = torch.where(chain_same, d_res, 2 * r_max + 1)
    f_pos = one_hot_nearest(d_res, torch.arange(0, 2 * r_max + 2))
    d_ch = torch.clamp(sym[:, None] - sym[None, :] + s_max, 0, 2 * s_max)
    d_ch = torch.where(chain_same | ~ent_same, 2 * s_max + 1, d_ch)
    f_ch = one_hot_nearest(d_ch, torch.arange(0, 2 * s_max + 2))
    dt = batch["rel_tok_feat"].dtype
    return torch.cat([f_pos.to(dt), batch["rel_tok_feat"], ent_same[..., None].to(dt), f_ch.to(dt)], dim=-1)


def atom_embedder(P, name, batch, inf, eps):
    """layers/diffusion_conditioning.py:110-128."""
    ref_pos, uid = batch["ref_pos"], batch["ref_space_uid"]
    d = ref_pos[:, None, :] - ref_pos[None, :, :]                 # (reference casts to fp32 here; inputs are fp32)
    v = (uid[:, None] == uid[None, :]).to(ref_pos.dtype)[..., None]
    a = linear(P, name + ".linear_c", batch["ref_feat"])
    p = linear(P, name + ".linear_p", d) * v
    p = p + linear(P, name + ".linear_d", 1 / (1 + torch.norm(d, dim=-1)[..., None])) * v
    p = p + linear(P, name + ".linear_v", v) * v
    ra = F.relu(a)
    ap = linear(P, name + ".linear_c_l", ra)[:, None, :] + linear(P, name + ".linear_c_m", ra)[None, :, :] + p
    ap = ap + feed_forward(P, name + ".ffn", ap)
    for b in range(_nblocks(P, name + ".atom_transformer.blocks")):
        a = atom_block(P, f"{name}.atom_transformer.blocks.{b}", a, ap, batch["ap_mask"], inf, eps)
    return a, ap


def template_pair_embedder(P, name, batch, z, inf, eps):
    """layers/diffusion_conditioning.py:38-50 (norm_in uses RMSNorm's default eps 1e-6, :24)."""
    tf = batch["templ_feat"]
    asym = batch["asym_id"]
    chain_same = (asym[None] == asym[:, None]).to(tf.dtype)
    mask = batch["z_mask"] * tf[..., 39] * chain_same
    u = linear(P, name + ".linear_in", rms_norm(P, name + ".norm_in", z, 1e-6)) + linear(P, name + ".linear_templ_feat", tf)
    for b in range(_nblocks(P, name + ".triangleformer.blocks")):
        u = triangle_block(P, f"{name}.triangleformer.blocks.{b}", u, mask, inf, eps)
    return linear(P, name + ".linear_out", F.relu(rms_norm(P, name + ".norm_out", u, eps))) * batch["t_mask"]


def token_embedder(P, name, batch, a, inf, eps, return_parts=False, s_pool=None):
    """layers/diffusion_conditioning.py:178-202.  ``s_pool`` (diagnostics only, tools/pool_noise_cpu.py): a replacement for
    the pooled atom activations (the output of the reference's `downscale`, :168-176)."""
    chunk = batch["token_id_to_chunk_sizes"]
    z_mask = batch["z_mask"]
    s = segment_mean_pool(F.silu(linear(P, name + ".linear_a", a)), chunk) if s_pool is None else s_pool
    s = s + linear(P, name + ".linear_target_feat", batch["target_feat"]) \
          + linear(P, name + ".linear_key_res_feat", batch["key_res_feat"]) \
          + linear(P, name + ".linear_pocket_res_feat", batch["pocket_res_feat"][..., None])
    z = linear(P, name + ".linear_s_i", s)[:, None, :] + linear(P, name + ".linear_s_j", s)[None, :, :] \
        + linear(P, name + ".rel_pos_embedder.linear", rel_pos_features(batch)) \
        + linear(P, name + ".linear_bonds", batch["token_bonds_feature"][..., None])
    m = linear(P, name + ".linear_msa_feat", batch["msa_feat"]) + linear(P, name + ".linear_s_input", s)
    parts = {"s0": s, "z0": z, "m0": m}
    for b in range(_nblocks(P, name + ".evoformer.blocks")):
        m, z = evoformer_block(P, f"{name}.evoformer.blocks.{b}", m, z, z_mask, inf, eps)
    parts["z_evo"] = z
    z = z + template_pair_embedder(P, name + ".template_pair_embedder", batch, z, inf, eps)
    parts["z_templ"] = z
    s = linear(P, name + ".linear_m", m[0]) + linear(P, name + ".linear_s", s)
    for b in range(_nblocks(P, name + ".pairformer.blocks")):
        s, z = pairformer_block(P, f"{name}.pairformer.blocks.{b}", s, z, z_mask, inf, eps)
    if return_parts:
        return s, z, parts
    return s, z


def diffusion_conditioning(P, batch, inf=1e9, eps=1e-8, name="diffusion_conditioning", s_pool=None):
    """layers/diffusion_conditioning.py:232-238 -> (a, ap, s, z)."""
    a2t = batch["atom_id_to_token_id"]
    a, ap = atom_embedder(P, name + ".atom_embedder", batch, inf, eps)
    s, z = token_embedder(P, name + ".token_embedder", batch, a, inf, eps, s_pool=s_pool)
    a = a + linear(P, name + ".linear_s", rms_norm(P, name + ".norm_s", s, eps))[a2t]
    ap = ap + linear(P, name + ".linear_z", rms_norm(P, name + ".norm_z", z, eps))[a2t][:, a2t]
    return a, ap, s, z


# --------------------------------------------------------------------------- denoiser
def af3_dit(P, batch, x_hat, t_hat, a, ap, s, z, sigma_data=16.0, inf=1e9, eps=1e-8, name="dit"):
    """layers/transformers.py:218-262 (precond, atom enc, pool, token DiT, un-pool, atom dec, denoise)."""
    chunk, a2t = batch["token_id_to_chunk_sizes"], batch["atom_id_to_token_id"]
    c_in = 1 / torch.sqrt(t_hat[:, None, None] ** 2 + sigma_data ** 2)
    c_noise = torch.log(t_hat / sigma_data) / 4.0
    ba = linear(P, name + ".linear_x", x_hat * c_in) + a[None]
    t = timestep_embeddings(P, name + ".time_embedder", t_hat * c_noise)
    for b in range(_nblocks(P, name + ".atom_dit_encoder.blocks")):
        ba = dit_block(P, f"{name}.atom_dit_encoder.blocks.{b}", ba, ap, t, batch["ap_mask"], inf, eps)
    bs = segment_mean_pool(F.silu(linear(P, name + ".linear_downscale", ba)), chunk) + s[None]
    for b in range(_nblocks(P, name + ".token_dit.blocks")):
        bs = dit_block(P, f"{name}.token_dit.blocks.{b}", bs, z, t, batch["z_mask"], inf, eps)
    ba = ba + linear(P, name + ".linear_upscale", bs)[:, a2t]
    for b in range(_nblocks(P, name + ".atom_dit_decoder.blocks")):
        ba = dit_block(P, f"{name}.atom_dit_decoder.blocks.{b}", ba, ap, t, batch["ap_mask"], inf, eps)
    c_skip = (sigma_data ** 2 / (sigma_data ** 2 + t_hat ** 2))[:, None, None]
    c_out = (sigma_data * t_hat / torch.sqrt(sigma_data ** 2 + t_hat ** 2))[:, None, None]
    r = linear(P, name + ".linear_r", layer_norm(P, name + ".norm_r", ba, eps))
    return c_skip * x_hat + c_out * r


# --------------------------------------------------------------------------- sampler
def confidence_module(P, batch, s, z, x_pred, inf=1e9, eps=1e-8, name="confidence_module"):
    """layers/confidence_module.py:56-88 -> p_pae [T,T,c_pae], p_pde [T,T,c_pde], p_plddt [A,c_plddt]."""
    ctr = batch["token_id_to_centre_atom_id"]
    a2t = batch["atom_id_to_token_id"]
    xc = x_pred[0, ctr, :]
    z = z + linear(P, name + ".linear_s_i", s)[..., None, :] + linear(P, name + ".linear_s_j", s)[..., None, :, :]
    d = torch.norm(xc[..., None, :] - xc[..., None, :, :], dim=-1, keepdim=True)
    v_bins = torch.linspace(3.375, 24.375, 13).type(z.dtype)
    b = torch.argmin(torch.abs(d[..., None] - v_bins), dim=-1)              # utils/tensor_utils.py:673-686
    onehot = torch.zeros(d.shape[:-1] + (13,), dtype=z.dtype).scatter_(-1, b, 1)
    z = z + linear(P, name + ".linear_d", onehot)
    for blk in range(_nblocks(P, name + ".pairformer.blocks")):
        s, z = pairformer_block(P, f"{name}.pairformer.blocks.{blk}", s, z, batch["z_mask"], inf, eps)
    z = z + z.transpose(-2, -3)
    p_pae = linear(P, name + ".linear_pae", z)
    p_pde = linear(P, name + ".linear_pde", z)
    a = linear(P, name + ".linear_s_a", s)[a2t]
    ap = linear(P, name + ".linear_z_a", torch.norm(x_pred[0][None] - x_pred[0][:, None], dim=-1)[..., None])
    a1 = a
    for blk in range(_nblocks(P, name + ".atom_transformer.blocks")):
        a1 = atom_block(P, f"{name}.atom_transformer.blocks.{blk}", a1, ap, batch["ap_mask"], inf, eps)
    a = a + a1
    return p_pae, p_pde, linear(P, name + ".linear_plddt", a)


def karras_noise_schedule(num_steps=200, sigma_data=16, s_max=160, s_min=4 * 10e-4, p=7):
    """models/model.py:117-129, same torch fp32 op order (p=1000 amplifies rounding)."""
    idx = torch.arange(num_steps, dtype=torch.float32)
    t = sigma_data * (s_max ** (1 / p) + idx / (num_steps - 1) * (s_min ** (1 / p) - s_max ** (1 / p))) ** p
    return torch.cat([t, torch.zeros_like(t[:1])])


def sphere_point(u_phi, u_theta):
    """utils/tensor_utils.py:545-562 given the two uniform draws."""
    phi = u_phi * 2 * torch.pi
    theta = torch.acos(u_theta * 2 - 1)
    return torch.stack([torch.cos(phi) * torch.sin(theta), torch.sin(phi) * torch.sin(theta), torch.cos(theta)], dim=-1)


def rotation_from_uniforms(u):
    """utils/tensor_utils.py:565-573; u: [4, B] = (phi0, theta0, phi1, theta1) draws in call order."""
    e0 = sphere_point(u[0], u[1])
    e1 = sphere_point(u[2], u[3])
    e1 = e1 - e0 * (e1 * e0).sum(dim=-1, keepdim=True)
    e1 = e1 / torch.norm(e1, dim=-1, keepdim=True)
    e2 = torch.cross(e0, e1, dim=-1)
    return torch.stack([e0, e1, e2], dim=-2)


def centre_random_augmentation(x, x_exists, u, trans):
    """utils/tensor_utils.py:576-586 with the random draws passed in (u:[4,B], trans:[B,3])."""
    mean = torch.sum(x * x_exists[None, :, None], dim=-2, keepdim=True) / torch.sum(x_exists)
    R = rotation_from_uniforms(u)
    return torch.einsum("bij,bkj->bki", R, x - mean) + trans[:, None, :]


def weighted_rigid_align(x_pred, x_gt, weights):
    """utils/tensor_utils.py:744-778: returns x_gt moved onto x_pred (second argument moves)."""
    x_pred, x_gt, weights = x_pred.float(), x_gt.float(), weights.float()      # reference: fp32 under autocast-off
    if x_gt.dim() == 2:
        x_gt = x_gt[None]
    wsum = weights.sum()
    mu_p = (x_pred * weights[None, :, None]).sum(-2) / wsum
    mu_g = (x_gt * weights[None, :, None]).sum(-2) / wsum
    xp, xg = x_pred - mu_p[:, None, :], x_gt - mu_g[:, None, :]
    H = torch.einsum("bij,bik->bjk", xg * weights[None, :, None], xp)
    U, _, Vh = torch.linalg.svd(H)
    Fm = torch.eye(3)
    Fm[-1, -1] = -1
    R = U @ Vh
    R = torch.where((torch.det(R) < 0)[:, None, None], U @ Fm @ Vh, R).transpose(-1, -2)
    return torch.einsum("bij,bkj->bki", R, xg) + mu_p[:, None, :]


def template_epsilon(ligand_pos, ref_dist):
    """models/model.py:233-239: eps[b,c] = mean_ij 1/4 sum_k sigmoid(|D_b - D_c| - {.5,1,2,4})."""
    dist = torch.norm(ligand_pos[:, :, None] - ligand_pos[:, None], dim=-1)
    delta = (dist[:, None] - ref_dist[None]).abs()
    e = 0.25 * (torch.sigmoid(-0.5 + delta) + torch.sigmoid(-1 + delta) + torch.sigmoid(-2 + delta) + torch.sigmoid(-4 + delta))
    return e.mean(dim=[-1, -2])


def template_reselect(x_pred_ligand, ref_mol_poses, k):
    """redocking.py:326-335: mean eps over samples per conformer, argsort, first k."""
    ref_dist = torch.norm(ref_mol_poses[:, :, None] - ref_mol_poses[:, None], dim=-1)
    e = template_epsilon(x_pred_ligand, ref_dist).mean(dim=0)
    return torch.argsort(e)[:k], e


def sample_diffusion(P, batch, noise, num_sample=5, steps=200, gamma_0=0.8, gamma_min=1.0,
                     noise_scale_lambda=1.003, step_scale_eta=1.5, ode_step_scale_eta=1.0,
                     ref_mol_poses=None, mmff_gamma_0_factor=1.0, align_ref_pos=True,
                     karras_noise_schedule_power=7, sigma_data=16.0, inf=1e9, eps=1e-8,
                     conditioning=None, return_trajectory=False, ref_mol=None, relax_fn=None, mmff_iters=5):
    """models/model.py:157-282 with every random draw supplied by ``noise`` (parity mode):

    noise = {"init":[B,A,3], "rot_u":[steps,4,B], "trans":[steps,B,3], "diffuse":[n_noisy,B,A,3]}
    (draw order of the reference: model.py:148; per step tensor_utils.py:549-557 x2, :582; model.py:77).
    ``ref_mol`` / ``relax_fn``: the relaxation branch (model.py:252-261).  Its tensor ops and control flow are restated
    here and pinned by golden set G8 (reference run with `get_next_step_pos` patched to a deterministic torch function);
    the relaxation itself is the callable ``relax_fn(ref_mol, ligand_pos [B,L,3], mmff_iters) -> [B,L,3]`` - RDKit's own
    MMFF94 arithmetic (model.py:26-52) stays parity-unpinned (RDKit is not installed here; see oracle/mmff_oracle.py).
    """
    x_exists = batch["a_mask"]
    lig_w = batch["is_ligand"][batch["atom_id_to_token_id"]]
    is_lig = lig_w.bool()
    batch_ref_pos = batch["ref_pos"][None].repeat(num_sample, 1, 1)
    ref_dist = None
    if ref_mol_poses is not None:
        ref_dist = torch.norm(ref_mol_poses[:, :, None] - ref_mol_poses[:, None], dim=-1)
    a, ap, s, z = conditioning if conditioning is not None else diffusion_conditioning(P, batch, inf, eps)
    sigmas = karras_noise_schedule(steps, p=karras_noise_schedule_power)
    x_next = sigmas[0] * noise["init"]
    n_noisy = 0
    traj = []
    for i in range(steps):
        t_cur, t_next = sigmas[i], sigmas[i + 1]
        x_cur = centre_random_augmentation(x_next, x_exists, noise["rot_u"][i], noise["trans"][i])
        if t_cur > gamma_min:
            t_hat = torch.full([num_sample], float(t_cur * (gamma_0 + 1)))
            x_hat = x_cur + noise_scale_lambda * noise["diffuse"][n_noisy] * \
                torch.sqrt(t_hat ** 2 - t_cur ** 2)[:, None, None]
            n_noisy += 1
        else:
            t_hat = torch.full([num_sample], float(t_cur))
            x_hat = x_cur
        x_den = af3_dit(P, batch, x_hat, t_hat, a, ap, s, z, sigma_data, inf, eps)
        if align_ref_pos and t_cur > gamma_min * mmff_gamma_0_factor:
            w = x_exists * lig_w
            if ref_dist is not None and ref_dist.shape[-1] == int(is_lig.sum()):
                e = template_epsilon(x_den[:, is_lig], ref_dist)
                batch_ref_pos[:, is_lig] = ref_mol_poses[torch.argmin(e, dim=-1)]
            lig_den = weighted_rigid_align(x_den * x_exists[..., None], batch_ref_pos, w)
            d_lig = (x_hat - lig_den) / t_hat[:, None, None] * w[None, :, None]
            d_cur = (x_hat - x_den) / t_hat[:, None, None] * (1 - w[None, :, None]) + d_lig
        elif ref_mol is not None and t_cur <= gamma_min * mmff_gamma_0_factor:        # model.py:252-261
            w = x_exists * lig_w
            x_ref = x_den.clone()
            x_ref[:, is_lig] = relax_fn(ref_mol, x_den[:, is_lig], mmff_iters)
            lig_den = weighted_rigid_align(x_den * x_exists[..., None], x_ref, w)
            d_lig = (x_hat - lig_den) / t_hat[:, None, None] * w[None, :, None]
            d_cur = (x_hat - x_den) / t_hat[:, None, None] * (1 - w[None, :, None]) + d_lig
        else:
            d_cur = (x_hat - x_den) / t_hat[:, None, None]
        dt = (t_next - t_hat)[:, None, None]
        eta = step_scale_eta if t_cur > gamma_min else ode_step_scale_eta
        x_next = x_hat + eta * dt * d_cur
        if return_trajectory:
            traj.append(x_next.clone())
    return (x_next, traj) if return_trajectory else x_next
