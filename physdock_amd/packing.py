"""Weight re-packing done once at load time (pure layout changes, plus folding a LayerNorm
affine / the AdaLN "+1" into the projection that follows them).  Runs on whatever device
the parameters live on; results are cached until the parameters change."""
from __future__ import annotations

import math

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


def pad_k(W):
    """[N,K] -> [N, ceil4(K)] zero padded so every row start is 16-byte aligned."""
    N, K = W.shape
    Kp = (K + 3) // 4 * 4
    if Kp == K:
        return W.contiguous()
    out = torch.zeros(N, Kp, dtype=W.dtype, device=W.device)
    out[:, :K] = W
    return out


def split3_rows(W):
    """[N,K] fp32 -> [3][N][Kp] bf16 with W = hi + mid + lo exactly (round-to-nearest-even splits; the residuals are exact
    in fp32), Kp = K rounded up to a multiple of 32, zero padded."""
    N, K = W.shape
    Kp = (K + 31) // 32 * 32
    out = torch.zeros(3, N, Kp, dtype=torch.bfloat16, device=W.device)
    r = W.float()
    for part in range(3):
        h = r.to(torch.bfloat16)
        out[part, :, :K] = h
        r = r - h.float()
    return out


def split3_bf16(W):
    """The B operand of csrc/gemm_split.hip: the three bf16 parts of W [N,K] (split3_rows), stored FRAGMENT-MAJOR:
    [3][ceil(N/32)][Kp/16][64][8], element (n, k) of a part at lane 32 * ((k % 16) // 8) + n % 32, slot k % 8 of the
    (n // 32, k // 16) block - i.e. every 1 KB block is exactly one wave's v_mfma_f32_32x32x16_bf16 B operand (row n % 32
    = lane & 31, eight consecutive k at 8 * (lane >> 5)), so the kernel fetches it with one coalesced 16-byte load per lane."""
    rows = split3_rows(W)
    _, N, Kp = rows.shape
    Np = (N + 31) // 32 * 32
    if Np != N:
        rows = torch.cat([rows, torch.zeros(3, Np - N, Kp, dtype=rows.dtype, device=rows.device)], 1)
    v = rows.reshape(3, Np // 32, 32, Kp // 16, 2, 8).permute(0, 1, 3, 4, 2, 5)
    return v.contiguous()


def split2_f16(W, rows_per_scale=1):
    """The B operand of csrc/gemm_f16.hip: W [N,K] times a per-row power of two (max_k |W[n,k]| w_scale[n] in [2^14, 2^15)),
    split into two fp16 parts (hi = fp16(w'), lo = fp16(w' - hi): 22 significand bits), stored FRAGMENT-MAJOR like split3_bf16:
    [2][ceil(N/32)][Kp/16][64][8].  Returns (parts, w_inv [N] fp32 = 1 / w_scale).  Rows of zeros get scale 1.
    rows_per_scale = 32: ONE scale per 32-row fragment tile (its largest row decides) - for kernels that undo the scale with a scalar
    per tile (csrc/tri_attn.hip); w_inv then repeats the tile's value for its rows."""
    W = W.float()
    N, K = W.shape
    Kp = (K + 31) // 32 * 32
    amax = W.abs().amax(1)
    if rows_per_scale > 1:
        assert N % rows_per_scale == 0
        amax = amax.reshape(-1, rows_per_scale).amax(1, keepdim=True).expand(-1, rows_per_scale).reshape(-1)
    # w_scale = 2^(14 - floor(log2(amax))): frexp gives amax = m 2^e with m in [0.5, 1) -> floor(log2) = e - 1
    e = torch.frexp(amax)[1]
    w_scale = torch.where(amax > 0, torch.ldexp(torch.ones_like(amax), 15 - e), torch.ones_like(amax))
    Ws = W * w_scale[:, None]
    hi = Ws.to(torch.float16)
    lo = (Ws - hi.float()).to(torch.float16)
    rows = torch.zeros(2, N, Kp, dtype=torch.float16, device=W.device)
    rows[0, :, :K], rows[1, :, :K] = hi, lo
    Np = (N + 31) // 32 * 32
    if Np != N:
        rows = torch.cat([rows, torch.zeros(2, Np - N, Kp, dtype=rows.dtype, device=rows.device)], 1)
    v = rows.reshape(2, Np // 32, 32, Kp // 16, 2, 8).permute(0, 1, 3, 4, 2, 5)
    return v.contiguous(), (1.0 / w_scale).contiguous()


class PackedWeights:
    """Device-resident fp32 views of a reference-named state dict + cached packed forms."""

    def __init__(self, params, config, equalise=True):
        self.p = {k: v.detach().float().contiguous() for k, v in params.items()}
        self.cfg = config
        self.cache = {}
        self.equalised = self._equalise() if equalise else {}

    @staticmethod
    def _pow2_ratio(norms):
        """per-channel power of two r that brings norms / r within a factor sqrt(2) of the median norm - for channels at least a
        factor ~3 (2^1.5) away from it; the ordinary spread of a weight matrix's rows is left alone (r = 1, also for zero rows)"""
        nz = norms[norms > 0]
        if nz.numel() == 0:
            return torch.ones_like(norms)
        e = torch.round(torch.log2(norms.clamp_min(1e-30) / nz.median())).clamp(-40, 40)
        e = torch.where(e.abs() >= 2, e, torch.zeros_like(e))
        return torch.where(norms > 0, torch.exp2(e), torch.ones_like(norms))

    def _equalise(self):
        """Channel equalisation of producer / consumer weight pairs by POWERS OF TWO (exact in fp32, the network function is
        unchanged): value channels  linear_v row k / r_k, linear_o column k x r_k  (attention is linear in v), and SwiGLU channels
        w3 row n / r_n, w2 column n x r_n  (h_n = silu(w1_n y) (w3_n y) is linear in w3_n), with r = the power of two nearest to
        (row norm / median row norm), resp. (||w1_n|| ||w3_n|| / median).  A trained checkpoint with a few large value / hidden
        channels then presents operands v, o, h whose channels have comparable magnitude, so ONE power-of-two operand scale per
        launch (the two-part fp16 format, csrc/common.h) costs no precision on the ordinary channels.  Weights whose rows are
        already balanced - the seeded parity weights - get r = 1 everywhere and are left bit-identical.
        Returns {weight name: r} for the rows that were touched (diagnostics)."""
        done = {}
        for name in sorted(self.p):
            if name.endswith(".linear_v.weight"):
                pre = name[:-len(".linear_v.weight")]
                o = pre + ".linear_o.weight"
                if o not in self.p or self.p[o].shape[1] != self.p[name].shape[0]:
                    continue
                r = self._pow2_ratio(self.p[name].double().norm(dim=1).float())
            elif name.endswith(".w3.weight"):
                pre = name[:-len(".w3.weight")]
                o, w1 = pre + ".w2.weight", pre + ".w1.weight"
                if o not in self.p or w1 not in self.p or self.p[o].shape[1] != self.p[name].shape[0]:
                    continue
                r = self._pow2_ratio((self.p[w1].double().norm(dim=1) * self.p[name].double().norm(dim=1)).float())
            else:
                continue
            if bool((r != 1).any()):
                self.p[name] = (self.p[name] / r[:, None]).contiguous()
                self.p[o] = (self.p[o] * r[None, :]).contiguous()
                done[name] = r
        return done

    def __getitem__(self, name):
        return self.p[name]

    def w3(self, W, K=None):
        """cached bf16 x 3 split of a packed weight matrix [N, ld] (only its first K columns when ld is a padded K)"""
        if not isinstance(W, torch.Tensor) or W.dim() != 2:
            return None
        key = ("w3", W.data_ptr(), tuple(W.shape), K)
        v = self.cache.get(key)
        if v is None:
            v = (W, split3_bf16(W if K is None else W[:, :K]))       # holding W keeps its address from being reused
            self.cache[key] = v
        return v[1]

    def norm_bound(self, w, b, K):
        """device scalar sqrt(K) max|w| + max|b| (w / b None: 1 / 0): upper bound of a normalised row times a static affine"""
        key = ("normbound", None if w is None else w.data_ptr(), None if b is None else b.data_ptr(), K)

        def mk():
            wm = 1.0 if w is None else float(w[:K].abs().max())
            bm = 0.0 if b is None else float(b[:K].abs().max())
            dev = (w if w is not None else b).device if (w is not None or b is not None) else next(iter(self.p.values())).device
            return (w, b, torch.tensor([(math.sqrt(K) * wm + bm) * 1.0001], dtype=torch.float32, device=dev))
        return self._c(key, mk)[2]

    def w2(self, W, K=None):
        """cached two-part fp16 split (+ inverse row scales) of a packed weight matrix, see split2_f16"""
        if not isinstance(W, torch.Tensor) or W.dim() != 2:
            return None
        key = ("w2", W.data_ptr(), tuple(W.shape), K)
        v = self.cache.get(key)
        if v is None:
            v = (W,) + split2_f16(W if K is None else W[:, :K])
            self.cache[key] = v
        return v[1], v[2]

    def _c(self, key, fn):
        v = self.cache.get(key)
        if v is None:
            v = fn()
            self.cache[key] = v
        return v

    def linear(self, name):
        def mk():
            W = self.p[name + ".weight"]
            N, K = W.shape
            Wp = pad_k(W)
            return (Wp, self.p.get(name + ".bias"), N, K, Wp.shape[1])
        return self._c(("lin", name), mk)

    def qkvg(self, prefix):
        def mk():
            W = torch.cat([self.p[f"{prefix}.linear_{c}.weight"] for c in "qkvg"], 0).contiguous()
            C = W.shape[1]
            b = torch.cat([torch.zeros(3 * C, device=W.device), self.p[prefix + ".linear_g.bias"]]).contiguous()
            return (W, b)
        return self._c(("qkvg", prefix), mk)

    def qkv_g(self, prefix):
        """q|k|v rows of the packed q|k|v|g projection (a view: the first 3 C rows) and the separate gate projection (W_g, b_g)"""
        W, b = self.qkvg(prefix)
        C3 = 3 * (W.shape[0] // 4)
        return W[:C3], b[:C3], self.p[prefix + ".linear_g.weight"], self.p[prefix + ".linear_g.bias"]

    def qkv_folded_w2(self, prefix, norm_weight):
        """two-part fp16 split (parts, w_inv) of the q | k | v projection [3 C][C] with the RMSNorm gain folded into its columns
        (Wf[n][c] = W[n][c] w[c]), ONE power-of-two scale per (projection, head) tile of 32 rows: the operand of pd_tri_attention, which
        reads z / rms(z) without the gain and undoes the weight scale with a scalar per tile"""
        def mk():
            Wf = (self.qkv(prefix) * norm_weight[None, :]).contiguous()
            return (Wf,) + split2_f16(Wf, rows_per_scale=32)
        v = self._c(("qkv_folded_w2", prefix, norm_weight.data_ptr()), mk)
        return v[1], v[2]

    def attn_static_bounds_host(self, prefix, norm_weight):
        """[|q|, |k|, |v|] rigorous upper bounds of a trunk attention whose projections (no bias) read an RMS- / LayerNorm-ed row
        times the static gain `norm_weight`: |W_n . (x^ w)| <= ||W_n w||_2 ||x^||_2 and ||x^||_2 <= sqrt(C) (Cauchy-Schwarz;
        weights only, so it holds for any input) - the precondition of the two-part fp16 attention format.  fp32-rounded host
        floats: exactly the numbers the device copy (attn_static_bounds) holds."""
        def mk():
            w = norm_weight.double()
            out = []
            for c in "qkv":
                W = self.p[f"{prefix}.linear_{c}.weight"].double()
                out.append(float((W * w[None, :]).norm(dim=1).max()) * math.sqrt(W.shape[1]) * 1.0001)
            return [float(v) for v in torch.tensor(out, dtype=torch.float32)]
        return self._c(("attn_bounds_host", prefix, norm_weight.data_ptr()), mk)

    def attn_static_bounds(self, prefix, norm_weight):
        """device floats [3] of attn_static_bounds_host"""
        return self._c(("attn_bounds", prefix, norm_weight.data_ptr()), lambda: torch.tensor(
            self.attn_static_bounds_host(prefix, norm_weight), dtype=torch.float32, device=norm_weight.device))

    def glu_hidden_bound(self, prefix, norm_weight):
        """device scalar: rigorous upper bound of |silu(W1 y) (W3 y)| for y = x^ w, ||x^||_2 <= sqrt(C) (RMS- / LayerNorm-ed row
        times the static gain): |silu(a)| <= |a|, so |h_n| <= ||W1_n w||_2 ||W3_n w||_2 C - weights only, any input"""
        def mk():
            w = norm_weight.double()
            W1, W3 = self.p[prefix + ".w1.weight"].double(), self.p[prefix + ".w3.weight"].double()
            b = float(((W1 * w[None, :]).norm(dim=1) * (W3 * w[None, :]).norm(dim=1)).max()) * W1.shape[1] * 1.0001
            return torch.tensor([b], dtype=torch.float32, device=norm_weight.device)
        return self._c(("glu_h_bound", prefix, norm_weight.data_ptr()), mk)

    def qkv(self, prefix):
        return self._c(("qkv", prefix), lambda: torch.cat(
            [self.p[f"{prefix}.linear_{c}.weight"] for c in "qkv"], 0).contiguous())

    def headnorm(self, prefix):
        return self._c(("hn", prefix), lambda: torch.stack(
            [self.p[prefix + ".norm_q.weight"], self.p[prefix + ".norm_k.weight"]]).contiguous())

    def glu(self, prefix):
        def mk():
            W1, W3 = self.p[prefix + ".w1.weight"], self.p[prefix + ".w3.weight"]
            W, _ = pack_glu(W1, W3)
            return (pad_k(W), W1.shape[0])
        return self._c(("glu", prefix), mk)

    def tri_qk(self, prefix):
        def mk():
            g = lambda n: (self.p[f"{prefix}.linear_{n}.weight"], self.p[f"{prefix}.linear_{n}.bias"])
            (Wqx, bqx), (Wq, bq), (Wkx, bkx), (Wk, bk) = g("qx"), g("q"), g("kx"), g("k")
            Wa, ba = pack_glu(Wqx, Wq, bqx, bq)
            Wb, bb = pack_glu(Wkx, Wk, bkx, bk)
            return (torch.cat([Wa, Wb], 0).contiguous(), torch.cat([ba, bb]).contiguous())
        return self._c(("triqk", prefix), mk)

    def tri_qk_bounds(self, prefix, norm_weight):
        """device floats [2]: rigorous upper bounds of the triangle update's gated operands q = (W_qx zn + b) sigmoid(.) mask and k
        (same with kx): |sigmoid| <= 1, mask in [0, 1], zn = x^ w with ||x^||_2 <= sqrt(C) - weights only, any input"""
        def mk():
            w = norm_weight.double()
            out = []
            for c in ("qx", "kx"):
                W, b = self.p[f"{prefix}.linear_{c}.weight"].double(), self.p[f"{prefix}.linear_{c}.bias"].double()
                out.append(float(((W * w[None, :]).norm(dim=1) * math.sqrt(W.shape[1]) + b.abs()).max()) * 1.0001)
            return torch.tensor(out, dtype=torch.float32, device=norm_weight.device)
        return self._c(("triqk_bounds", prefix, norm_weight.data_ptr()), mk)

    def bias_w(self, prefix, norm_name):
        """linear_z weights of an attention with the gain of the norm in front folded in: Wf[h][k] = w[k] W[h][k]"""
        return self._c(("biasw", prefix, norm_name), lambda: (
            self.p[prefix + ".linear_z.weight"] * self.p[f"{prefix}.{norm_name}.weight"][None, :]).contiguous())

    def relpos_T(self):
        name = "diffusion_conditioning.token_embedder.rel_pos_embedder.linear.weight"
        return self._c(("relposT",), lambda: self.p[name].t().contiguous())

    def _dit_blocks(self, kind):
        dt = self.cfg.model.dit
        if kind == "atom":
            return [f"dit.atom_dit_encoder.blocks.{b}" for b in range(dt.no_blocks_atom)] + \
                   [f"dit.atom_dit_decoder.blocks.{b}" for b in range(dt.no_blocks_atom)]
        return [f"dit.token_dit.blocks.{b}" for b in range(dt.no_blocks_dit)]

    def dit_bias(self, kind):
        """Concatenated linear_z of every DiT block with that block's LayerNorm affine folded in:
        Wz.(xhat*w + b) = (Wz*w).xhat + Wz.b            (attentions.py:232,242,254)"""
        def mk():
            Ws, bs = [], []
            for blk in self._dit_blocks(kind):
                Wz = self.p[blk + ".attention.linear_z.weight"]
                w, b = self.p[blk + ".attention.norm_z.weight"], self.p[blk + ".attention.norm_z.bias"]
                Ws.append(Wz * w[None, :])
                bs.append((Wz * b[None, :]).sum(1))
            W = pad_k(torch.cat(Ws, 0))
            return (W, torch.cat(bs).contiguous(), W.shape[0])
        return self._c(("ditbias", kind), mk)

    def dit_qk_bounds_host(self, kind):
        """(q bound, k bound) of a DiT family: per-head RMS norm => |q| <= sqrt(32) max|gain_q|; ONE pair for all blocks of the family
        (the maximum over its blocks), so that every block's attention launch derives the same operand scales and the hoisted bias
        tiles of the family carry one pre-scale (ops.attn_bias_prescale).  fp32-rounded host floats = the device table's entries."""
        def mk():
            up = 1.0001
            hq = max(float(self.headnorm(blk + ".attention")[0].abs().max()) for blk in self._dit_blocks(kind))
            hk = max(float(self.headnorm(blk + ".attention")[1].abs().max()) for blk in self._dit_blocks(kind))
            return tuple(float(v) for v in torch.tensor([math.sqrt(32.0) * hq * up, math.sqrt(32.0) * hk * up], dtype=torch.float32))
        return self._c(("ditqk", kind), mk)

    def dit_bound_consts(self, kind):
        """[blocks][4] = (q bound, k bound, 0, 0): the weight-only entries of the activation bounds pd_dit_bounds derives for the
        two-part fp16 operand format (csrc/sampler.hip); the v / h bounds come from dit_bound_weights and the step's AdaLN row"""
        def mk():
            qb, kb = self.dit_qk_bounds_host(kind)
            rows = [[qb, kb, 0.0, 0.0] for _ in self._dit_blocks(kind)]
            return torch.tensor(rows, dtype=torch.float32, device=self.p[self._dit_blocks(kind)[0] + ".attention.linear_v.weight"].device)
        return self._c(("ditbound", kind), mk)

    def dit_bound_weights(self, kind):
        """([blocks][C + 2 hidden][C] fp32, hidden): per DiT block the rows of linear_v, w1 and w3 as the projections use them (after
        the channel equalisation) - pd_dit_bounds bounds |v_n| and |h_n| per output row against the step's AdaLN modulation"""
        def mk():
            blocks = []
            for blk in self._dit_blocks(kind):
                blocks.append(torch.cat([self.p[blk + ".attention.linear_v.weight"], self.p[blk + ".transition.feed_forward.w1.weight"],
                                         self.p[blk + ".transition.feed_forward.w3.weight"]], 0))
            hidden = self.p[self._dit_blocks(kind)[0] + ".transition.feed_forward.w1.weight"].shape[0]
            return (torch.stack(blocks).contiguous(), hidden)
        return self._c(("ditboundw", kind), mk)

    def adaln(self, kind):
        """All AdaLN-Zero projections of one DiT family stacked: per block
        [attention.norm_s (shift|scale|gate), transition.ffn_norm (shift|scale|gate)], with +1 folded
        into the scale bias so the table holds (shift, 1+scale, gate)   (adaptive_layer_norm_zero.py:19-20)"""
        def mk():
            Ws, bs = [], []
            for blk in self._dit_blocks(kind):
                for sub in (".attention.norm_s.linear", ".transition.ffn_norm.linear"):
                    W, b = self.p[blk + sub + ".weight"], self.p[blk + sub + ".bias"].clone()
                    C = W.shape[0] // 3
                    b[C:2 * C] += 1.0
                    Ws.append(W)
                    bs.append(b)
            return (torch.cat(Ws, 0).contiguous(), torch.cat(bs).contiguous())
        return self._c(("adaln", kind), mk)
