"""Thin Python launch wrappers over the C ABI (one function per launcher).

These add no arithmetic: they fill the argument structs from tensor metadata and call
libphysdock_hip.so on the current torch stream.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import threading

import torch

from . import _lib
from ._lib import (ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SILU, OUT_BIASFRAG, OUT_OPM, OUT_ROWMAJOR,
                   OUT_TRANSPOSED, AttnArgs, GemmArgs, check, ptr, stream)

RMS, LN = 0, 1

_CONST = {}
_CONST_LOCK = threading.Lock()


def const_vec(val, n):
    """Cached device vector of a constant (ones / zeros for an absent norm gain / shift).  The cache is shared by every
    host thread / HIP stream of the process (parallel.StreamPool): a vector is published only after the fill has
    completed on the device, so a launch on another stream can never read it half-written."""
    key = (float(val), torch.cuda.current_device())
    v = _CONST.get(key)
    if v is None or v.numel() < n:
        with _CONST_LOCK:
            v = _CONST.get(key)
            if v is None or v.numel() < n:
                v = torch.full((max(n, 4096),), float(val), device="cuda", dtype=torch.float32)
                torch.cuda.current_stream().synchronize()
                _CONST[key] = v
    return v


_PRESPLIT_OK = {}


def presplit_supported(M, N, K, *, glu=0, hn=False, f16=False, gate=False, per_group_rows=0, gstride=0):
    """Does the library take a pre-split A operand for this projection - [3][M][K] bf16 (pd_gemm_args.A3), or with f16=True
    [2][M][K] fp16 (A2, csrc/gemm_f16.hip)?  Asked of the library itself (pd_gemm_variant: tile-count threshold, alignment and
    epilogue rules live in csrc/gemm_split.hip / gemm_f16.hip), so a differently tuned build changes the answer here, not an
    error in pd_gemm."""
    key = (M, N, K, int(glu), bool(hn), bool(f16), bool(gate), int(per_group_rows), int(gstride))
    r = _PRESPLIT_OK.get(key)
    if r is None:
        a = GemmArgs()
        a.A = a.W = a.Y = 1 << 20                        # never dereferenced by the query: aligned placeholders
        if f16:
            a.W2 = a.w_inv = a.a_amax = a.A2 = 1 << 20
        else:
            a.W3 = a.A3 = 1 << 20
        a.M, a.N, a.K = M, N, K
        a.lda, a.ldw, a.ldy = K, K, (N // 2 if glu else N)
        a.batch, a.glu, a.out_scale = 1, int(glu), 1.0
        if hn:
            a.hn_w, a.hn_cols, a.hn_split = 1 << 20, 0, 32
        if gate:                                         # gate x (acc + bias) + residual: one gate row for all rows, or - per_group_rows
            a.res = a.mul = 1 << 20                      # (training-time forward: one AdaLN row per sample) - one per row group
            a.ldres, a.mul_rows_per_group = N, (per_group_rows or M)
            a.mul_gstride = gstride if per_group_rows else 0
        r = _lib.init().pd_gemm_variant(C.byref(a)) >= (2000000 if f16 else 1000000)
        _PRESPLIT_OK[key] = r
    return r


_KV2_OK = {}
_INLINE_STATS_OK = {}


def kv2_supported(M, Cdim, *, a2, per_group_rows=0):
    """Does the q|k|v projection of a DiT block ([M, C] -> [M, 3 C], head-norm epilogue) take the fp16-format kernel AND write
    k | v pre-split for the attention kernel (pd_gemm_args.Y2)?  Asked of the library (pd_gemm_variant), as presplit_supported."""
    key = (M, Cdim, bool(a2), int(per_group_rows))
    r = _KV2_OK.get(key)
    if r is None:
        a = GemmArgs()
        a.A = a.W = a.Y = 1 << 20
        a.W2 = a.w_inv = a.a_amax = a.Y2 = a.y2_amax = 1 << 20
        if a2:
            a.A2 = 1 << 20
        else:
            a.stats = a.pro_w = a.pro_b = 1 << 20
            a.pro_rows_per_group, a.pro_gstride = int(per_group_rows), (6 * Cdim if per_group_rows else 0)
        a.M, a.N, a.K = M, 3 * Cdim, Cdim
        a.lda, a.ldw, a.ldy = Cdim, Cdim, 3 * Cdim
        a.batch, a.out_scale = 1, 1.0
        a.hn_w, a.hn_cols, a.hn_split = 1 << 20, 2 * Cdim, Cdim
        a.y2_col0, a.ldy2 = Cdim, 4 * Cdim
        r = _lib.init().pd_gemm_variant(C.byref(a)) >= 2000000
        _KV2_OK[key] = r
    return r


_ROWS_OK = {}


def rows_inline_supported(M, N, K, *, hn=False, y2=False, glu=0, per_group_rows=0, gstride=0):
    """Does the library run this norm-prologue projection on one of the fp16-format ROWS kernels (whole rows resident in LDS, own
    statistics: csrc/gemm_f16.hip gemm_f16_rows_kernel at K = 128, gemm_f16_wrows_kernel at K = 512)?  Asked of the library
    (pd_gemm_variant with stats_inline), as presplit_supported."""
    key = (M, N, K, bool(hn), bool(y2), int(glu), int(per_group_rows), int(gstride))
    r = _ROWS_OK.get(key)
    if r is None:
        a = GemmArgs()
        a.A = a.W = a.Y = a.W2 = a.w_inv = a.a_amax = a.pro_w = a.pro_b = 1 << 20
        a.M, a.N, a.K = M, N, K
        a.lda, a.ldw, a.ldy = K, K, (N // 2 if glu else N)
        a.batch, a.glu, a.out_scale = 1, int(glu), 1.0
        a.stats_inline, a.stats_eps = 2, 1e-5
        a.pro_rows_per_group, a.pro_gstride = int(per_group_rows), int(gstride)
        if hn:
            a.hn_w, a.hn_cols, a.hn_split = 1 << 20, 2 * (N // 3), N // 3
        if y2:
            a.Y2 = a.y2_amax = 1 << 20
            a.y2_col0, a.ldy2 = N // 3, 4 * (N // 3)
        v = _lib.init().pd_gemm_variant(C.byref(a))
        r = v >= 2000000 and (v % 1000000) // 100000 >= 3
        _ROWS_OK[key] = r
    return r


def gemm(A, W, Y, M, N, K, *, lda=None, ldw=None, ldy=None, batch=1, sA=0, sW=0, sY=0,
         a_kmajor=False, w_kmajor=False, stats=None, pro_w=None, pro_b=None, pro_rows_per_group=0,
         pro_gstride=0, pro_act=ACT_NONE, rowscale_acc=None, bias=None, sBias=0, hn_w=None, hn_cols=0,
         hn_split=32, hn_eps=0.0, act=ACT_NONE, glu=0, rowscale=None, maskadd=None, maskval=0.0,
         mul=None, ldmul=0, mul_rows_per_group=0, mul_gstride=0, out_scale=1.0, res=None, ldres=0,
         res_row_mod=0, sRes=0, out_mode=OUT_ROWMAJOR, T1=0, T2=0, frag_transpose=False, W3=None, ksplit_ws=None, A3=None,
         W2=None, a_amax=None, A2=None, Y2=None, y2_amax=None, y2_col0=0, stats_inline=None):
    """Y = epilogue(prologue(A) @ W^T); see include/physdock_hip.h pd_gemm_args.
    A/W/Y and the optional operands may be tensors or raw device addresses (ints).
    stats_inline=(mode, eps) with stats = an (uninitialised) [M, 2] scratch tensor: the row statistics of A are the launch's own
    business - computed inside the kernel when pd_gemm would run it on the fp32 streaming kernel anyway (few samples, the trunk's
    small tracks: one launch less), by a pd_rowstats launch into `stats` first otherwise."""
    def P(x):
        return x if (x is None or isinstance(x, int)) else ptr(x)
    n_out = N // 2 if glu else N
    if stats is not None:
        if pro_w is None:
            pro_w = const_vec(1.0, K)
        if pro_b is None:
            pro_b = const_vec(0.0, K)
    a = GemmArgs()
    a.A, a.W, a.Y = P(A), P(W), P(Y)
    a.W3 = W3.data_ptr() if (W3 is not None and SPLIT_GEMM) else None
    a.A3 = A3.data_ptr() if A3 is not None else None
    if W2 is not None and a_amax is not None and F16_GEMM and SPLIT_GEMM:
        # two-part fp16 operands (csrc/gemm_f16.hip): W2 = (parts, w_inv) of packing.split2_f16, a_amax = device scalar bound of |A'|
        a.W2, a.w_inv = W2[0].data_ptr(), W2[1].data_ptr()
        a.a_amax = a_amax if isinstance(a_amax, int) else ptr(a_amax)
        a.A2 = A2.data_ptr() if A2 is not None else None
        if Y2 is not None:       # head-norm epilogue: columns >= y2_col0 (k | v) written pre-split for pd_attention (K2 / V2)
            a.Y2, a.y2_col0 = Y2.data_ptr(), int(y2_col0)
            a.y2_amax = y2_amax if isinstance(y2_amax, int) else ptr(y2_amax)
            a.ldy2 = int(Y2.shape[-1])
    elif A2 is not None or Y2 is not None:
        raise ValueError("A2 / Y2 (pre-split fp16 operands) need W2 and a_amax")
    a.M, a.N, a.K = M, N, K
    a.lda = lda if lda is not None else (M if a_kmajor else K)
    a.ldw = ldw if ldw is not None else (N if w_kmajor else K)
    a.ldy = ldy if ldy is not None else (M if out_mode == OUT_TRANSPOSED else n_out)
    a.batch, a.sA, a.sW, a.sY = batch, sA, sW, sY
    a.a_kmajor, a.w_kmajor = int(a_kmajor), int(w_kmajor)
    a.stats, a.pro_w, a.pro_b = P(stats), P(pro_w), P(pro_b)
    a.pro_rows_per_group, a.pro_gstride, a.pro_act = pro_rows_per_group, pro_gstride, pro_act
    a.rowscale_acc, a.bias, a.sBias = P(rowscale_acc), P(bias), sBias
    a.hn_w, a.hn_cols, a.hn_split, a.hn_eps = P(hn_w), hn_cols, hn_split, hn_eps
    a.act, a.glu = act, glu
    a.rowscale, a.maskadd, a.maskval = P(rowscale), P(maskadd), maskval
    a.mul, a.ldmul, a.mul_rows_per_group, a.mul_gstride = P(mul), ldmul, mul_rows_per_group, mul_gstride
    a.out_scale = out_scale
    a.res, a.ldres, a.res_row_mod, a.sRes = P(res), (ldres or n_out), res_row_mod, sRes
    a.out_mode, a.T1, a.T2, a.frag_transpose = out_mode, T1, T2, int(frag_transpose)
    if ksplit_ws is not None and KSPLIT_GEMM:
        a.ksplit_ws, a.ksplit_ws_bytes = ptr(ksplit_ws), ksplit_ws.numel() * 4
    if stats_inline is not None:
        mode_, eps_ = stats_inline
        key = (M, N, K, a.lda, int(glu), hn_w is not None, mul is not None, res is not None, pro_rows_per_group > 0, out_mode, act, pro_act,
               a.W2 is not None, a.W3 is not None, a.ksplit_ws is not None, bool(a_kmajor), batch)
        key = key + (INLINE_STATS, F16_ROWS, a.A2 is not None, a.a_amax is not None)
        # everything else the library's choice depends on: epilogue operands, the alignment class of A and of the
        # prologue vectors, the group stride, and the run-time switches a test or a lab run may flip
        key = key + (bias is not None, Y2 is not None, int(hn_split or 0), int(pro_gstride or 0), (a.A or 0) & 15, (a.pro_w or 0) & 15,
                     (a.pro_b or 0) & 15, F16_GEMM, SPLIT_GEMM, KV_PRESPLIT, KSPLIT_GEMM, F16_WIDE_ROWS)
        ok = _INLINE_STATS_OK.get(key)
        if ok is None:
            ok = False
            if F16_ROWS and F16_GEMM and a.W2 and a.a_amax and not a.A2:
                # rows of 128 channels (the atom q | k | v projection): the fp16-format rows kernel keeps whole rows in LDS and
                # computes their statistics itself (csrc/gemm_f16.hip: gemm_f16_rows_kernel)
                saved = a.stats
                a.stats, a.stats_inline, a.stats_eps = None, (1 if mode_ == RMS else 2), float(eps_)
                ok = _lib.init().pd_gemm_variant(C.byref(a)) >= 2000000
                a.stats, a.stats_inline = saved, 0
            if not ok:
                v = _lib.init().pd_gemm_variant(C.byref(a))              # with the statistics as an operand: which kernel takes it?
                if INLINE_STATS and 5000 <= v % 10000 and 0 <= v < 1000000:   # the fp32 streaming kernel: it can compute them itself
                    a.stats, a.stats_inline, a.stats_eps = None, (1 if mode_ == RMS else 2), float(eps_)
                    ok = _lib.init().pd_gemm_variant(C.byref(a)) >= 0
            _INLINE_STATS_OK[key] = ok
        if ok:
            a.stats, a.stats_inline, a.stats_eps = None, (1 if mode_ == RMS else 2), float(eps_)
        else:
            a.stats_inline = 0
            a.stats = P(stats)
            rowstats(A, stats, M, K, ldx=a.lda, mode=mode_, eps=eps_)
    if GEMM_HOOK is not None and not getattr(_TLS, "no_hooks", False):
        return GEMM_HOOK(a, lambda: check(_lib.init().pd_gemm(C.byref(a), stream()), "pd_gemm"))
    check(_lib.init().pd_gemm(C.byref(a), stream()), "pd_gemm")


def lab_set_trace(kind, buf):
    """Lab builds only (PD_LAB=1 python -m physdock_amd.build --force): point the in-kernel phase trace of the GEMM
    ("gemm": int64 tensor of 64*4*5*64 entries, tools/gemm_trace.py) or attention kernel ("attn": 16*4*4*64,
    tools/attn_trace.py) at `buf` (None: off).  The production library has no trace code and no such symbol."""
    fn = getattr(_lib.init(), f"pd_lab_set_{kind}_trace", None)
    if fn is None:
        raise RuntimeError("phase traces need a lab build: PD_LAB=1 python -m physdock_amd.build --force")
    fn.argtypes = [C.c_void_p]
    check(fn(buf.data_ptr() if buf is not None else None), "pd_lab_set_trace")


#: use the pre-split bf16 x 6 contraction (csrc/gemm_split.hip) where a launcher is given split weights; False forces the
#: fp32-MFMA kernels everywhere (A/B comparisons in the tests and the bench)
SPLIT_GEMM = True

#: normalise + split the activations of wide-row projections once (pd_norm_split -> pd_gemm_args.A3) instead of in the GEMM's staging
PRESPLIT_GEMM = True
PRESPLIT_QKV = True       # also for q|k|v (12 column blocks): +0.4 % on top of the SwiGLU projection's +1.0 %
PRESPLIT_MIN_C_F16 = 256  # narrowest rows that take the pre-split path when the operand format is two fp16 parts (atom rows,
                          # C = 128: the split pass costs 43 us per launch against 16 us of statistics - measured -0.8 % end to end)

#: launches on the fp32 streaming kernel compute the row statistics of their norm prologue themselves (gemm(stats_inline=)): one
#: pd_rowstats launch less in front of every small projection (B = 1: ~38 per step).  Correct (tests/test_gemm_ksplit_gpu.py) and
#: OFF: the statistics pass at the head of every block (two dependent sweeps over its rows) costs the consuming kernels more
#: than the 4.6 us launch it saves - token SwiGLU 11.2 -> 22.2 us, q|k|v 8.7 -> 13.5 us, B = 1 call 88.5 -> 104.4 ms
INLINE_STATS = False

#: projections of 128-channel rows with a norm prologue (the atom q | k | v of a DiT block) on the fp16-format ROWS kernel: a block
#: keeps its 128 rows, normalised and split once, in LDS for all column tiles and computes their statistics itself (no pd_rowstats
#: launch, one pass over the rows instead of four)
F16_ROWS = True
#: ... and the token-level q | k | v and SwiGLU up-projections (K = 512) on the wide-rows kernel instead of pd_norm_split2 + the tile kernel
F16_WIDE_ROWS = True
F16_WIDE_ROWS_GLU = True

#: K-split of launches that cannot fill the chip (few samples) when the caller hands pd_gemm a scratch buffer
KSPLIT_GEMM = True

#: the same switch for the attention kernel (csrc/attn_split.hip vs csrc/attention.hip)
SPLIT_ATTN = True

#: two-part fp16 operands (three partial products) for attention launches whose caller supplies magnitude bounds (f16_amax=)
F16_ATTN = True
#: linear_downscale + SiLU + token pooling of the denoiser as one kernel (csrc/pool.hip); False: pd_gemm + pd_segment_pool
FUSED_POOL = True

#: chip-filling fp16-format launches on the software-pipelined kernel (csrc/attn_pipe.hip) when the bias was produced pre-scaled
#: (bias_prescale=); False: attn_parts_kernel (csrc/attn_f16.hip) with an unscaled bias - callers must then not pre-scale
PIPE_ATTN = True
#: the same for GEMM launches that carry fp16-split weights and a bound of |A| (W2=, a_amax=)
F16_GEMM = True
#: GEMMs with a norm prologue and a static gain / shift derive the bound themselves (sqrt(K) max|w| + max|b|): the trunk
F16_NORM_BOUND = True
#: the atom-level DiT transition (C = 128, hidden = 384) as one kernel with the hidden activations in LDS
FUSED_TRANSITION = True
TRANSITION_HOOK = None
#: ... and the trunk's pair transition (RMSNorm, static gain; [T*T][128] rows) on the same kernel
FUSED_TRUNK_TRANSITION = True
#: the tail of the TriangleUpdate (gate projection, norm of the einsum output, K = 32 projection, gate, residual) in one launch
FUSED_TRI_TAIL = True
#: ... and the tail of the TriangleAttention (gate projection, linear_o, gate, residual; the projection in front shrinks to q|k|v).
#: Round 4 measured it slower behind a q|k|v GEMM (45.4 + 39.6 us against 50.5 (q|k|v|g) + 24.8 us) and left it off; with the projection
#: inside the attention block (FUSED_TRI_ATTN, round 6) the gate has no projection to ride on, and the tail (one launch) against gate
#: GEMM + linear_o GEMM (two) measures 22.9 - 23.3 ms against 23.4 ms per trunk pass: on.  PD_FUSED_TRI_ATTN_TAIL=0 switches it off.
FUSED_TRI_ATTN_TAIL = os.environ.get("PD_FUSED_TRI_ATTN_TAIL", "1") != "0"
#: the triangle einsum on the two-part fp16 format (csrc/tri_mul.hip) instead of 32 batched fp32-MFMA GEMMs: "row" = the outgoing
#: form only (15.8 vs 16.1 us at T = 256, error vs float64 6.8e-8 vs 1.0e-7 rms); the incoming form's transposing LDS scatter
#: measured 23.4 vs 16.6 us and stays on the k-major fp32 kernel; True = both forms
F16_TRI_MUL = "row"
#: fp16-parts attention launches write their output already split for the projection that follows (pd_attn_args.O2 -> A2)
ATTN_SPLIT_OUT = True
#: the q|k|v projection writes k | v already scaled and split for the fp16-parts attention kernel (pd_gemm_args.Y2 -> pd_attn_args.K2 / V2):
#: the 8 query blocks of a (sample, head) stage K / V tiles with copies instead of re-splitting them
KV_PRESPLIT = True
#: trunk attention (triangle, MSA row / column, pair-biased single / atom attention) on the fp16-parts kernel with STATIC bounds
#: of q, k, v from the projection weights and the norm gain (packing.attn_static_bounds); False: bf16 x 6 as in round 2
F16_TRUNK_ATTN = True
#: ... and the projections that consume the attention output (|o| <= max|v|) and the SwiGLU hidden activations of the trunk's
#: transitions (packing.glu_hidden_bound) on the fp16-format GEMM
F16_TRUNK_GEMM = True

#: optional profiling hook: GEMM_HOOK(args_struct, launch_fn) (bench.py brackets launches with HIP events)
GEMM_HOOK = None
ATTN_HOOK = None
#: per-thread switch: `_TLS.no_hooks = True` hides the hooks from the launches of THIS host thread (engine.check_dit_bounds runs on
#: StreamPool worker threads while other threads sample: it must not touch the process-wide hook variables)
_TLS = threading.local()


def rowstats(x, stats, M, Cdim, *, ldx=None, kmajor=False, mode=RMS, eps=1e-8):
    xp = x if isinstance(x, int) else ptr(x)
    check(_lib.init().pd_rowstats(xp, ptr(stats), M, Cdim, ldx if ldx is not None else (M if kmajor else Cdim),
                                  int(kmajor), mode, eps, stream()), "pd_rowstats")


def norm_split(x, out3, M, Cdim, *, ldx=None, mode=RMS, eps=1e-8, w=None, b=None, rows_per_group=0, gstride=0):
    """out3 [3, M, C] bf16 = error-free split of (x - mean) rstd w[g] + b[g] (pd_norm_split): the pre-split A operand of gemm(A3=)"""
    def P(t):
        return t if (t is None or isinstance(t, int)) else ptr(t)
    check(_lib.init().pd_norm_split(P(x), ldx if ldx is not None else Cdim, M, Cdim, mode, eps, P(w), P(b), rows_per_group, gstride,
                                    ptr(out3), stream()), "pd_norm_split")


def norm_split2(x, out2, M, Cdim, a_amax, *, ldx=None, mode=RMS, eps=1e-8, w=None, b=None, rows_per_group=0, gstride=0):
    """out2 [2, M, C] fp16 = (hi, lo) split of ((x - mean) rstd w[g] + b[g]) 2^e, e from the device scalar bound a_amax
    (pd_norm_split2): the pre-split A operand of gemm(A2=)"""
    def P(t):
        return t if (t is None or isinstance(t, int)) else ptr(t)
    check(_lib.init().pd_norm_split2(P(x), ldx if ldx is not None else Cdim, M, Cdim, mode, eps, P(w), P(b), rows_per_group, gstride,
                                     P(a_amax), ptr(out2), stream()), "pd_norm_split2")


def transition_f16(x, M, Cdim, hidden, *, shift, scale1p, gate, W13, W2, y_amax, h_amax, eps, rows_per_group=0, gstride=0, rms=False):
    """fused atom-level DiT transition (pd_transition_f16); returns False when the library does not cover the shape.
    shift / scale1p / gate / y_amax / h_amax: device addresses or tensors; W13 / W2: (parts, w_inv) of packing.split2_f16"""
    def P(t):
        return t if (t is None or isinstance(t, int)) else ptr(t)
    a = _lib.TransitionArgs()
    a.x, a.M, a.C, a.hidden = ptr(x), M, Cdim, hidden
    a.shift, a.scale1p, a.gate = P(shift), P(scale1p), P(gate)
    a.rows_per_group, a.gstride, a.eps = rows_per_group, gstride, eps
    a.W13, a.w13_inv, a.W2, a.w2_inv = W13[0].data_ptr(), W13[1].data_ptr(), W2[0].data_ptr(), W2[1].data_ptr()
    a.y_amax, a.h_amax = P(y_amax), P(h_amax)
    a.rms = int(rms)
    def launch():
        rc = _lib.init().pd_transition_f16(C.byref(a), stream())
        if rc != -3:
            check(rc, "pd_transition_f16")
        return rc != -3
    if TRANSITION_HOOK is not None:          # profiling hook (bench.py): brackets the launch
        return TRANSITION_HOOK(a, launch)
    return launch()


def tri_tail(z, o, M, Cdim, Co, *, w_in, w_out, eps, Wg, bg, Wz, bz, zn_amax, on_amax, mode=0):
    """tail of a TriangleUpdate in one launch (pd_tri_tail): z += sigmoid(Wg RMSNorm(z) + bg) * (Wz RMSNorm(o) + bz); Wg / Wz:
    (parts, w_inv) of packing.split2_f16.  Returns False when the library does not cover the shape."""
    a = _lib.TriTailArgs()
    a.z, a.o, a.M, a.C, a.Co = ptr(z), ptr(o), M, Cdim, Co
    a.w_in, a.w_out, a.eps = ptr(w_in), (ptr(w_out) if w_out is not None else None), eps
    a.Wg, a.wg_inv, a.bg = Wg[0].data_ptr(), Wg[1].data_ptr(), ptr(bg)
    a.Wz, a.wz_inv, a.bz = Wz[0].data_ptr(), Wz[1].data_ptr(), ptr(bz)
    a.zn_amax, a.on_amax = (zn_amax if isinstance(zn_amax, int) else ptr(zn_amax)), (on_amax if isinstance(on_amax, int) else ptr(on_amax))
    a.mode = int(mode)
    rc = _lib.init().pd_tri_tail(C.byref(a), stream())
    if rc != -3:
        check(rc, "pd_tri_tail")
    return rc != -3


#: TriangleAttention with the q | k | v projection inside the attention block (csrc/tri_attn.hip, round 6); False: projection GEMM + pd_attention
FUSED_TRI_ATTN = os.environ.get("PD_FUSED_TRI_ATTN", "1") != "0"


def tri_z2_numel(T):
    """fp16 elements of the split normalised rows pd_pair_bias_split writes for pd_tri_attention: [T][ceil(T/32)][8][2][64][8]"""
    return T * ((T + 31) // 32) * 8 * 2 * 64 * 8


def pair_bias_split(x, Wf, frag, T, z2, *, stats_out=None, maskadd=None, maskval=0.0, out_scale=1.0, transpose=False, eps=1e-8,
                    zn_amax):
    """pd_pair_bias for a TriangleAttention (C = 128, H = 4, RMS) that also writes the normalised rows, scaled and split, in
    pd_tri_attention's fragment order into z2 (int16 / fp16 tensor of tri_z2_numel(T) elements, zeroed once by the caller)"""
    if T % 4 != 0:
        return False
    check(_lib.init().pd_pair_bias_split(ptr(x), ptr(Wf), None, ptr(stats_out), ptr(maskadd), maskval, out_scale, ptr(frag), T,
                                         int(transpose), eps, z2.data_ptr(), float(zn_amax), stream()), "pd_pair_bias_split")
    return True


def tri_attention(z2, W2, bias, o, T, Treal, Cdim, nheads, *, transpose, bias_prescale, bias_nk, qkv_amax, zn_amax):
    """TriangleAttention up to the attention output in one launch (pd_tri_attention): q | k | v projection of the normalised rows z2
    (pair_bias_split's output) -> biased attention.  W2: (parts, w_inv) of packing.split2_f16(rows_per_scale=32) of the [3 C][C]
    projection with the norm gain folded in.  Returns False when the library does not cover the shape."""
    a = _lib.TriAttnArgs()
    a.z2, a.W2, a.w_inv = z2.data_ptr(), W2[0].data_ptr(), W2[1].data_ptr()
    a.bias, a.bias_prescale, a.bias_nk, a.o = ptr(bias), float(bias_prescale), int(bias_nk), ptr(o)
    a.T, a.Treal, a.C, a.nheads, a.transpose = T, Treal, Cdim, nheads, int(bool(transpose))
    a.zn_amax, a.qkv_amax, a.scale = float(zn_amax), (qkv_amax if isinstance(qkv_amax, int) else ptr(qkv_amax)), 1.0 / math.sqrt(32.0)
    rc = _lib.init().pd_tri_attention(C.byref(a), stream())
    if rc != -3:
        check(rc, "pd_tri_attention")
    return rc != -3


def tri_mul(q, k, o, T, Treal, nch, ch_stride, *, transpose, q_amax, k_amax):
    """triangle-multiplication einsum on the two-part fp16 format (pd_tri_mul); q / k / o: tensors or raw device addresses.
    Returns False when the library does not cover the shape."""
    def P(x):
        return x if isinstance(x, int) else ptr(x)
    a = _lib.TriMulArgs()
    a.q, a.k, a.o = P(q), P(k), P(o)
    a.T, a.Treal, a.nch, a.ch_stride, a.transpose = T, Treal, nch, ch_stride, int(transpose)
    a.q_amax, a.k_amax = P(q_amax), P(k_amax)
    rc = _lib.init().pd_tri_mul(C.byref(a), stream())
    if rc != -3:
        check(rc, "pd_tri_mul")
    return rc != -3


def rownorm(x, y, M, Cdim, *, res=None, w=None, b=None, mode=RMS, eps=1e-8, act=ACT_NONE):
    check(_lib.init().pd_rownorm(ptr(x), ptr(y), ptr(res), ptr(w), ptr(b), M, Cdim, mode, eps, act, stream()),
          "pd_rownorm")


#: (C, H) combinations pd_pair_bias is instantiated for, and the ones where it beats rowstats + GEMM on MI355X
#: (tools/kbench.py --pair-bias: z H=4 10 vs 26 us, H=8 14 vs 27, H=16 22 vs 29; ap H=4 101 vs 549 us, H=24 354 vs 639 us)
PAIR_BIAS_SHAPES = {(128, 4), (128, 8), (128, 16), (16, 4), (16, 24)}
PAIR_BIAS_FASTER = {(128, 4), (128, 8), (128, 16), (16, 4), (16, 24)}


def pair_bias(x, Wf, frag, T1, T2, Cdim, H, *, c2=None, stats_out=None, maskadd=None, maskval=0.0, out_scale=1.0,
              transpose=False, mode=RMS, eps=1e-8, only_if_faster=False):
    """fragment-layout attention bias of x [T1*T2, C] in one streaming pass (see pd_pair_bias); returns False when the shape
    is not covered - or, with only_if_faster, not a win - (the caller then uses rowstats + gemm)"""
    if (Cdim, H) not in (PAIR_BIAS_FASTER if only_if_faster else PAIR_BIAS_SHAPES) or T2 % 4 != 0:
        return False
    check(_lib.init().pd_pair_bias(ptr(x), ptr(Wf), ptr(c2), ptr(stats_out), ptr(maskadd), maskval, out_scale, ptr(frag), T1, T2,
                                   Cdim, H, int(transpose), mode, eps, stream()), "pd_pair_bias")
    return True


def attn_split_ws_numel(nbatch, nq, nk, nheads):
    """floats of scratch that let pd_attention split the key range of a launch too small to fill the chip (0: no split)"""
    blocks = nbatch * nheads * ((nq + 127) // 128)
    nit = (nk + 63) // 64
    if blocks >= 512 or nit < 8:
        if os.environ.get("PD_ATTN_TAIL") != "1":
            return 0
        # lab builds with -DPD_ATTN_TAIL=1 (tools/ab_tail.sh): scratch for the key-split TAIL round of a chip-filling pipelined launch -
        # the arithmetic of csrc/attention.hip attn_tail (measured: no gain, NOTES round 5; the shipped library never asks for it)
        bps = nheads * ((nq + 255) // 256)
        if nq <= 128 or bps > 256 or 512 % bps:
            return 0
        per = 512 // bps
        bt = nbatch % per
        if nbatch < per or bt == 0 or bt * bps > 256:
            return 0
        s = min(4, 512 // (bt * bps), nit // 4)
        return s * bt * nq * nheads * 34 if s >= 2 else 0
    s = min(8, 1024 // blocks, nit // 4)
    return s * nbatch * nq * nheads * 34 if s >= 2 else 0


def unsplit_f16_attention(variant):
    """pd_attention_variant id of an UNSPLIT fp16-parts launch (2000 + waves): the form that can write O2 / read K2, V2
    (2000 + waves + 100 * chunks is the key-split form for a handful of samples)"""
    return variant >= 2000 and variant % 1000 < 100


def attention(Q, K, V, O, *, nq, nk, nbatch, nheads, q_strides, k_strides, v_strides, o_strides, bias=None,
              scale=1.0 / math.sqrt(32.0), ws=None, bias_nk=0, f16_amax=None, O2=None, query_only=False,
              KV2=None, kv2_strides=None, bias_prescale=0.0):
    """strides = (batch_stride, seq_stride) in floats; Q/K/V/O tensors or raw addresses.  ws: optional float scratch
    tensor (attn_split_ws_numel) enabling key-split launches for small grids.  bias_nk: key count the bias buffer was laid
    out for (the padded count when nk is the real one)."""
    def P(x):
        return x if (x is None or isinstance(x, int)) else ptr(x)
    a = AttnArgs()
    a.Q, a.K, a.V, a.O = P(Q), P(K), P(V), P(O)
    a.O2 = P(O2)
    a.nq, a.nk, a.nbatch, a.nheads = nq, nk, nbatch, nheads
    a.q_bs, a.q_ss = q_strides
    a.k_bs, a.k_ss = k_strides
    a.v_bs, a.v_ss = v_strides
    a.o_bs, a.o_ss = o_strides
    a.bias = P(bias)
    a.scale = scale
    a.bias_nk = bias_nk
    a.bias_prescale = (float(bias_prescale) if bias is not None else 0.0) if PIPE_ATTN else -1.0
    a.fp32_mfma = 0 if SPLIT_ATTN else 1
    if f16_amax is not None and F16_ATTN:      # (max|q|, max|k|, max|v|) upper bounds: three floats by value, or a device tensor [3]
        a.f16x3 = 1
        if isinstance(f16_amax, torch.Tensor):
            a.f16_amax = ptr(f16_amax)
        elif isinstance(f16_amax, int):          # raw device address of three floats
            a.f16_amax = f16_amax
        else:
            a.f16_q_amax, a.f16_k_amax, a.f16_v_amax = (float(v) for v in f16_amax)
    if KV2 is not None:      # [rows][4 C] fp16: k | v as written by gemm(Y2=) (groups of 4 dims: 4 high parts, 4 low parts)
        Cc = nheads * 32
        a.K2, a.V2 = KV2.data_ptr(), KV2.data_ptr() + 4 * Cc          # v follows k inside a row: 2 C fp16 elements = 4 C bytes
        a.kv2_bs, a.kv2_ss = kv2_strides
    if ws is not None:
        a.ws, a.ws_bytes = ptr(ws), ws.numel() * 4
    if query_only:            # the kernel pd_attention would pick (pd_attention_variant): >= 2000 = fp16-parts kernel
        return _lib.init().pd_attention_variant(C.byref(a))
    if ATTN_HOOK is not None and not getattr(_TLS, "no_hooks", False):
        return ATTN_HOOK(a, lambda: check(_lib.init().pd_attention(C.byref(a), stream()), "pd_attention"))
    check(_lib.init().pd_attention(C.byref(a), stream()), "pd_attention")


def attn_bias_prescale(q_amax, k_amax, scale=1.0 / math.sqrt(32.0)):
    """the power of two a bias producer folds into out_scale so that the pipelined fp16-format attention kernel (csrc/attn_pipe.hip)
    can take the bias tile as the initial value of its score accumulator: the product of the q and k operand scales the kernel
    derives from the same bounds (pd_attention_bias_prescale_log2: host arithmetic identical to the device's)"""
    return 2.0 ** _lib.lib().pd_attention_bias_prescale_log2(float(q_amax), float(k_amax), float(scale))


def bias_frag_numel(nheads, nq, nk):
    return nheads * ((nq + 31) // 32) * ((nk + 31) // 32) * 1024


def bias_to_frag(bias):
    """[H,nq,nk] dense bias (CPU or device tensor) -> fragment layout, scaled by log2(e).
    Test helper mirroring the address map of pd_gemm's PD_OUT_BIASFRAG store."""
    H, nq, nk = bias.shape
    nqt, nkt = (nq + 31) // 32, (nk + 31) // 32
    pad = torch.zeros(H, nqt * 32, nkt * 32, dtype=bias.dtype, device=bias.device)
    pad[:, :nq, :nk] = bias * _lib.LOG2E
    # [H, qt, q5, kt, k5] with k5 = 8*g + 4*hh + e ; frag index = g*256 + (q5 + 32*hh)*4 + e
    x = pad.reshape(H, nqt, 32, nkt, 4, 2, 4)            # q5, kt, g, hh, e
    x = x.permute(0, 1, 3, 4, 5, 2, 6)                    # H, qt, kt, g, hh, q5, e
    return x.reshape(-1).contiguous()
