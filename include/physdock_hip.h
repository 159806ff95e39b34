/* physdock_hip.h - C ABI of libphysdock_hip.so (MI355X / gfx950).
 *
 * The reference (KexinZhangResearch/PhysDock) has no FFI layer: its hot path
 * `PhysDock.sample_diffusion` (PhysDock/models/model.py:157-282) is PyTorch all the way
 * down to ATen.  This header is the boundary this build introduces *below* the Python
 * class `physdock_amd.PhysDock` (which mirrors the reference class, model.py:55-68):
 * every entry point replaces the ATen op sequence named in its comment.
 *
 * Conventions: plain device pointers (fp32 unless noted) and sizes; `stream` is a
 * hipStream_t; return value 0 = ok, negative = error (PD_ERR_*).  No allocation, no
 * synchronisation and no global mutable state inside any launcher, so every call is
 * legal inside hipStreamBeginCapture / EndCapture (hipGraph).
 */
#ifndef PHYSDOCK_HIP_H
#define PHYSDOCK_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define PD_ABI_VERSION 9

enum { PD_OUT_ROWMAJOR = 0, PD_OUT_TRANSPOSED = 1, PD_OUT_OPM = 2, PD_OUT_BIASFRAG = 3 };

/* ---- pd_gemm: Y = epilogue(prologue(A) . W^T) ---------------------------------------
 * replaces F.linear (primitives/linear.py:161) together with the norm in front of it
 * (rms_norm.py:14-19, nn.LayerNorm, adaptive_layer_norm_zero.py:18-21), the SwiGLU /
 * sigmoid gates (feed_forward.py:30-31, attentions.py:161-163), the residual add that
 * follows (transformers.py:20-21,49-53,157-158) and the two einsums
 * (attentions.py:164, outer_product_mean.py:28).                                       */
typedef struct pd_gemm_args {
    const float* A;          /* [M,K] row-major (lda) or, if a_kmajor, [K,M] */
    const float* W;          /* [N,K] row-major (ldw) or, if w_kmajor, [K,N] */
    const void* W3;          /* optional: W pre-split into three bf16 parts, fragment-major [3][ceil(N/32)][Kp/16][64][8] with Kp =
                                32*ceil(K/32), zero padded: element (n,k) at lane 32*(k%16/8) + n%32, slot k%8 (packing.split3_bf16)
                                (w = hi + mid + lo exactly).  When given and the problem is one of the full-tile row-major
                                shapes, the contraction runs as six bf16 MFMAs per block with fp32 accumulation
                                (csrc/gemm_split.hip: at least the accuracy of the fp32 MFMA, 2.67x its peak rate); W must
                                still be valid (ragged row remainders and ineligible shapes use it).  NULL: fp32 MFMA.   */
    const void* A3;          /* optional: A already normalised and split into three bf16 parts [3][M][K] (pd_norm_split); K % 32
                                == 0, no prologue fields.  Only the split-operand kernel reads it: a launch that cannot run there
                                (no W3, ragged rows, too few tiles) returns PD_ERR_UNSUPPORTED instead of using the raw A.      */
    float* Y;
    int M, N, K;
    int lda, ldw, ldy;
    int batch;               /* blockIdx.z batches with strides sA/sW/sY (floats) */
    long long sA, sW, sY;
    int a_kmajor, w_kmajor;
    /* prologue on A: a' = act((a - mean[m]) * rstd[m] * pro_w[k] + pro_b[k]); stats = [M][2] */
    const float* stats;
    const float* pro_w;      /* NULL -> 1 */
    const float* pro_b;      /* NULL -> 0 */
    int pro_rows_per_group;  /* >0: pro_w/pro_b row = (m / rows_per_group) * pro_gstride */
    int pro_gstride;
    int pro_act;             /* PD_ACT_* applied to A after the affine (also without stats) */
    /* epilogue, in this order */
    const float* rowscale_acc;   /* [M]  acc *= rowscale_acc[m]                              */
    const float* bias;           /* [N]  (+ batch * sBias)                                   */
    long long sBias;
    const float* hn_w;           /* per-head RMSNorm over each 32-column tile with base < hn_cols:
                                    weight row = (tile_base / hn_split), hn_w = [rows][32]    */
    int hn_cols, hn_split;
    float hn_eps;
    int act;                     /* PD_ACT_*; ignored when glu != 0                          */
    int glu;                     /* 1: silu(a)*b   2: a*sigmoid(b); columns packed per 64 as [a(32) | b(32)] */
    const float* rowscale;       /* [M]  v *= rowscale[m]                                    */
    const float* maskadd;        /* [M]  v += maskval where maskadd[m] == 0                  */
    float maskval;
    const float* mul;            /* gate: v *= mul[m*ldmul + n]  or per row group            */
    int ldmul;
    int mul_rows_per_group;      /* >0: v *= mul[(m / rows_per_group) * mul_gstride + n]     */
    int mul_gstride;
    float out_scale;             /* v *= out_scale (0 -> 1)                                  */
    const float* res;            /* v += res[(m % res_row_mod) * ldres + n] (may alias Y)    */
    int ldres;
    int res_row_mod;
    long long sRes;
    int out_mode;                /* PD_OUT_*                                                 */
    int T1, T2;                  /* OPM: T2 = tokens; BIASFRAG: rows m = (i,j), i<T1, j<T2   */
    int frag_transpose;          /* BIASFRAG: query = j, key = i                             */
    int vecA, vecW, vecY;        /* set by the launcher                                      */
    /* optional K-split scratch (ABI 4): float workspace.  Given it, a launch whose full tiles cannot fill the chip (few
       samples) is cut along K into `ksplit` parts per tile: a first launch stores every part's partial accumulators here, a
       second one adds them IN FIXED ORDER (bit-reproducible, no floating-point atomics) and runs the epilogue.  NULL: never.  */
    void* ksplit_ws;
    long long ksplit_ws_bytes;
    int ksplit;                  /* set by the launcher                                      */
    /* ABI 5: two-part fp16 operand format (csrc/gemm_f16.hip: three fp16 MFMAs per block instead of six bf16 ones).  Used when
       all of W2, w_inv and a_amax are given and the launch is a chip-filling full-tile row-major problem; otherwise the launch
       falls through to W3 / fp32 as before.  The caller guarantees |A'[m,k]| <= *a_amax for the A the contraction sees (after
       the prologue): a larger element overflows fp16.  pd_dit_bounds derives such bounds for the DiT blocks.                  */
    const void* W2;              /* W[n,:] * w_scale[n] as two fp16 parts (hi, lo), fragment-major [2][ceil(N/32)][Kp/16][64][8]
                                    (packing.split2_f16); w_scale[n] = power of two that brings max_k |W[n,k]| into [2^14, 2^15)      */
    const float* w_inv;          /* [N]  1 / w_scale[n]                                                                        */
    const void* A2;              /* optional: A already normalised, modulated, scaled by the power of two pd_gemm derives from
                                    *a_amax, and split into two fp16 parts [2][M][K] (pd_norm_split2); K % 32 == 0, no prologue  */
    const float* a_amax;         /* device scalar: upper bound of |A'|                                                         */
    /* ABI 6, head-norm epilogue of the fp16-format kernel only (the q|k|v projection of a DiT block): output columns
       n >= y2_col0 (k and v) are NOT stored to Y but written ALREADY SCALED AND SPLIT for the attention kernel
       (pd_attn_args.K2 / V2) into Y2, fp16 elements, row stride ldy2 = 2 * (N - y2_col0): inside a row every group of four
       columns c = n - y2_col0 = 4 g + e occupies eight consecutive elements - the four high parts at 8 g + e, the four low
       parts at 8 g + 4 + e - so that one 16-byte load yields both parts of four values.  A value is multiplied by the power
       of two pd_attention derives from the bound y2_amax[c / hn_split] (device floats: max|k|, max|v|) before it is split.
       Any launch that cannot honour Y2 fails with PD_ERR_UNSUPPORTED - it is never silently ignored.                      */
    void* Y2;
    const float* y2_amax;
    int y2_col0, ldy2;
    /* ABI 7: row statistics computed INSIDE the consuming kernel: stats == NULL and stats_inline = 1 (RMSNorm) or 2 (LayerNorm,
       two passes: mean, then centred squares - the arithmetic of pd_rowstats) with stats_eps; pro_w / pro_b / groups as with
       stats.  Two kernels do this: the fp32 streaming kernel (csrc/gemm_stream.hip: launches too small to fill the chip - few
       samples, the trunk's single / MSA tracks): a block re-reads its rows (K floats each, L2-resident) before its main loop,
       which saves the separate pd_rowstats launch - latency, not bytes; and the fp16-format ROWS kernel (csrc/gemm_f16.hip,
       gemm_f16_rows_kernel: K == 128, W2 / w_inv / a_amax given, whole 64- / 128-row tiles, plain or head-norm epilogue): a
       block keeps its rows, normalised and split once, in LDS for every column tile - one pass over A instead of one per
       column tile plus the statistics pass.  Anything else answers PD_ERR_UNSUPPORTED.                                     */
    int stats_inline;
    float stats_eps;
} pd_gemm_args;
int pd_gemm(const pd_gemm_args* args, void* stream);
/* id of the kernel instantiation pd_gemm would launch for these arguments (for profiling);
 * id % 10000 >= 5000: gemm_stream_kernel<id % 10, (id / 10000) % 10, Tile> (csrc/gemm_stream.hip) takes it,
 * Tile = (id / 100000) % 10: 0 -> <128,128,2>, 1 -> <64,64,2>, 2 -> <128,64,4>; id >= 1000000: the split-operand
 * kernel gemm_split_kernel<...> (csrc/gemm_split.hip) with the same template arguments; id >= 2000000: gemm_f16_kernel<...>
 * (csrc/gemm_f16.hip, two-part fp16 operands; tile field (id % 1000000) / 100000 = 3 / 4: gemm_f16_rows_kernel<pro, EPI, 128 / 64>) */
int pd_gemm_variant(const pd_gemm_args* args);

/* ---- pd_rowstats: per-row (mean, rstd) for the GEMM prologue --------------------------
 * mode 0: RMS  -> (0, rsqrt(mean(x^2)+eps));  mode 1: LayerNorm -> (mean, rsqrt(var+eps)).
 * x is [M,C] (ldx) or, if kmajor, [C,M] (ldx = M stride).                               */
int pd_rowstats(const float* x, float* stats, int M, int C, int ldx, int kmajor, int mode, float eps, void* stream);

/* ---- pd_rownorm: y = [res +] act(norm(x) * w + b) (standalone normalisation)            */
int pd_rownorm(const float* x, float* y, const float* res, const float* w, const float* b,
               int M, int C, int mode, float eps, int act, void* stream);

/* ---- pd_norm_split: rows normalised, modulated and split for pd_gemm_args.A3 -----------------
 * out3 [3][M][C] bf16 (hi, mid, lo: a' = hi + mid + lo exactly) of a'[m,k] = (x[m,k] - mean_m) rstd_m w[g][k] + b[g][k],
 * g = m / rows_per_group (0: one row of w / b for all; NULL w / b: 1 / 0) - the expression of pd_gemm's norm prologue
 * (rms_norm.py:14-19, adaptive_layer_norm_zero.py:16-21) evaluated ONCE per element instead of once per column block of
 * the GEMM that consumes it.  C % 32 == 0.                                                                            */
int pd_norm_split(const float* x, int ldx, int M, int C, int mode, float eps, const float* w, const float* b,
                  int rows_per_group, int gstride, void* out3, void* stream);
/* pd_norm_split2 (ABI 5): the same rows times the power of two derived from the device scalar *a_amax (an upper bound of |a'|),
 * as TWO fp16 parts out2 [2][M][C] (hi, lo) - the pre-split A operand pd_gemm_args.A2 of csrc/gemm_f16.hip.                    */
int pd_norm_split2(const float* x, int ldx, int M, int C, int mode, float eps, const float* w, const float* b,
                   int rows_per_group, int gstride, const float* a_amax, void* out2, void* stream);

/* ---- pd_transition_f16 (ABI 5): the atom-level DiT transition in one launch --------------------------------------------------
 * x[m,:] += gate[g] * W2 . ( silu(W1 y) * (W3 y) ),  y = scale1p[g] * LayerNorm(x[m,:]) + shift[g],  g = m / rows_per_group
 * (transitions.py:27-30 with adaptive_layer_norm_zero.py:16-21 and feed_forward.py:30-31): row statistics, SwiGLU projection
 * and down-projection of 64 whole rows per block, the hidden activations stay in LDS.  C = 128 and hidden = 384 only, M % 64
 * == 0 and M >= 2048 (PD_ERR_UNSUPPORTED otherwise: run pd_rowstats + two pd_gemm).  W13 / W2 are the two-part fp16
 * fragment-major forms of the packed [a(32) | b(32)] SwiGLU weights [2 hidden][C] and of W2 [C][hidden] with their inverse row
 * scales (packing.split2_f16); y_amax / h_amax are device scalars bounding |y| and |silu(a) b| (pd_dit_bounds).
 * args == NULL: one-time set-up (dynamic LDS limit), called by pd_init.                                                        */
typedef struct pd_transition_args {
    float* x;                    /* [M][C], updated in place */
    int M, C, hidden;
    const float* shift; const float* scale1p; const float* gate;      /* [C] each (+ g * gstride)                      */
    int rows_per_group, gstride; /* 0: one row of shift / scale / gate for all rows                                        */
    float eps;
    const void* W13; const float* w13_inv; const void* W2; const float* w2_inv;
    const float* y_amax; const float* h_amax;
    int rms;                     /* ABI 7: 1 = RMSNorm instead of LayerNorm (no mean subtraction): with shift = zeros, scale1p = the norm
                                    gain and gate = ones this is the trunk's pair Transition (transitions.py:15-18) on [T*T][128] rows  */
} pd_transition_args;
int pd_transition_f16(const pd_transition_args* args, void* stream);

/* ---- pd_tri_tail (ABI 7): the tail of the trunk's TriangleUpdate in one launch ------------------------------------------------
 * z[m,:] += sigmoid(W_g RMSNorm(z[m,:]) w_in + b_g) * (W_z RMSNorm(o[:,m]) w_out + b_z)     (attentions.py:163,170-171)
 * for the channel-major einsum output o [Co][M] (attentions.py:164): replaces the gate projection, the column statistics of o and
 * the K = Co projection with gate + residual (three launches, a 33 MB gate tensor) - HBM sees z in, o in, z out.  C = 128 and
 * Co = 32 only (PD_ERR_UNSUPPORTED otherwise).  W_g [C][C] / W_z [C][Co] in the two-part fp16 fragment-major form with inverse
 * row scales (packing.split2_f16); zn_amax / on_amax: device scalars bounding the normalised rows times their gains
 * (sqrt(C) max|w_in|, sqrt(Co) max|w_out|).  args == NULL: one-time set-up, called by pd_init.                               */
typedef struct pd_tri_tail_args {
    float* z;                    /* [M][C], updated in place */
    const float* o;              /* [Co][M] */
    int M, C, Co;
    const float* w_in; const float* w_out;       /* norm gains [C], [Co] */
    float eps;
    const void* Wg; const float* wg_inv; const float* bg;
    const void* Wz; const float* wz_inv; const float* bz;
    const float* zn_amax; const float* on_amax;
    int mode;                    /* 0: TriangleUpdate tail as above.  1: TriangleAttention tail (attentions.py:204,212-213):
                                    z[m,:] += (W_g RMSNorm(z[m,:]) w_in + b_g) * (W_z o[m,:] + b_z) with the attention output o [M][C]
                                    (row-major, Co = C, |o| <= *on_amax = the v bound), a RAW gate, w_out unused              */
} pd_tri_tail_args;
int pd_tri_tail(const pd_tri_tail_args* args, void* stream);

/* ---- pd_tri_attention (ABI 9): TriangleAttention up to the attention output with the q | k | v projection INSIDE the attention
 * block (reference primitives/attentions.py:194-211; csrc/tri_attn.hip).  Replaces, per instance, pd_gemm (RMSNorm prologue, q | k | v)
 * + pd_attention:  o[b, r, 32 h + d] = sum_k softmax_k( q[b,r,h,:] . k[b,k,h,:] / sqrt(32) + bias[h, r, k] ) v[b,k,h,d]  with
 * q | k | v = (z[b, r, :] / rms(z[b, r, :])) . Wf^T, Wf = the [3 C][C] projection with the norm gain folded in (Wf[n][c] = W[n][c] w[c]).
 * z2: the normalised rows, scaled, split into two fp16 parts and in fragment order, as pd_pair_bias_split writes them (same zn_amax,
 * same transpose flag); o: [T][T][C] in the pair tensor's own layout - transpose == 0: batch b = first index, sequence r = second;
 * 1: the other way round (the column variant - nothing is transposed in memory).  W2 / w_inv = packing.split2_f16(Wf, rows_per_scale =
 * 32) (two fp16 parts, fragment-major, ONE power-of-two scale per 32-row tile; w_inv[n] = its inverse); bias: fragment layout for
 * (nq = T, nk = bias_nk) ALREADY multiplied by bias_prescale = the power of two pd_attention_bias_prescale_log2 derives from
 * qkv_amax[0..1]; qkv_amax: device floats [3] bounding |q|, |k|, |v|; zn_amax: host float bounding |z / rms| (sqrt(C));
 * Treal: real key count (keys >= Treal are masked).  C = 128, nheads = 4, T <= 256, T % 4 == 0; else PD_ERR_UNSUPPORTED.        */
typedef struct pd_tri_attn_args {
    const void* z2;
    const void* W2; const float* w_inv;
    const float* bias; float bias_prescale; int bias_nk;
    float* o;
    int T, Treal, C, nheads, transpose;
    float zn_amax; const float* qkv_amax;
    float scale;                 /* 1 / sqrt(32) */
} pd_tri_attn_args;
int pd_tri_attention(const pd_tri_attn_args* args, void* stream);
int pd_tri_attn_args_size(void);

/* ---- pd_tri_mul (ABI 7): the triangle-multiplication einsum (attentions.py:164) on the two-part fp16 format ----------------------
 * transpose == 0:  o[c,i,I] = sum_{j < Treal} q[c,i,j] k[c,I,j];   transpose == 1:  o[c,a,b] = sum_{j < Treal} k[c,j,a] q[c,j,b]
 * for nch channel planes [T][T] (plane stride ch_stride floats) of q, k, o; q_amax / k_amax: device scalars bounding |q|, |k|
 * (the gated projection's linear part: ||W_n w||_2 sqrt(C) + |b_n|).  T % 4 == 0.                                              */
typedef struct pd_tri_mul_args {
    const float* q; const float* k; float* o;
    int T, Treal, nch;
    long long ch_stride;
    int transpose;
    const float* q_amax; const float* k_amax;
} pd_tri_mul_args;
int pd_tri_mul(const pd_tri_mul_args* args, void* stream);

/* ---- pd_pair_bias: attention pair bias in one streaming pass (pairbias.hip) -------------------
 * frag = fragment layout of [ (norm(x) . Wf^T + c2 + maskadd ? 0 : maskval) * out_scale ] for x [T1*T2, C] (C = 16 or 128),
 * Wf [H][C] = projection weights with the norm gain folded in (Wf[h][k] = w[k] W[h][k]), c2 [H] = projection of the norm
 * shift (NULL: 0), H in {4, 8, 16} for C = 128 and {4, 24} for C = 16; mode 0 RMS / 1 LayerNorm.  Replaces linear_z(norm_z(z))
 * of attentions.py:38-41,82-85,200-203,246,254 (= pd_rowstats + pd_gemm PD_OUT_BIASFRAG) with one read of x; stats_out
 * (optional, [T1*T2][2]) receives the (mean, rstd) pairs for the projection GEMM that follows.  T2 % 4 == 0.              */
int pd_pair_bias(const float* x, const float* Wf, const float* c2, float* stats_out, const float* maskadd, float maskval,
                 float out_scale, float* frag, int T1, int T2, int C, int H, int frag_transpose, int mode, float eps,
                 void* stream);
/* pd_pair_bias_split (ABI 9): pd_pair_bias for the TriangleAttention (x [T*T][128], H = 4, RMS) that ALSO writes x / rms(x) times the
 * power-of-two operand scale of zn_amax (= sqrt(C) 1.0001), split into two fp16 parts, in pd_tri_attention's fragment order:
 * z2 [T batches][ceil(T/32) row tiles][8 k-steps][2 parts][64 lanes][8] halves (batch / row = the pair indices, swapped when
 * frag_transpose).  Rows beyond T of the last tile are not written: zero the buffer once.                                         */
int pd_pair_bias_split(const float* x, const float* Wf, const float* c2, float* stats_out, const float* maskadd, float maskval,
                       float out_scale, float* frag, int T, int frag_transpose, float eps, void* z2, float zn_amax, void* stream);

/* ---- pd_attention: O = softmax(Q K^T * scale + bias) V, head width 32 ------------------
 * replaces F.scaled_dot_product_attention at attentions.py:48,92,130,211,259.
 * Q/K/V/O element (b, i, h, d) at ptr[b*bs + i*ss + h*32 + d].  bias is in the fragment
 * layout [H][ceil(nq/32)][ceil(nk/32)][4][64][4] and already multiplied by log2(e)
 * (written by pd_gemm PD_OUT_BIASFRAG); NULL = no bias.  bias is shared by all batches. */
typedef struct pd_attn_args {
    const float* Q; const float* K; const float* V; float* O;
    int nq, nk, nbatch, nheads;
    long long q_bs, q_ss, k_bs, k_ss, v_bs, v_ss, o_bs, o_ss;
    const float* bias;
    float scale;             /* 1/sqrt(32) */
    int bias_nk;             /* key count the bias buffer was laid out for (pd_gemm PD_OUT_BIASFRAG's T2, i.e. the PADDED
                                count when nk is the real one); 0 -> nk                                                  */
    int fp32_mfma;           /* 1: keep both contractions on v_mfma_f32_32x32x2_f32 (csrc/attention.hip); 0 (default): launches
                                that fill the chip run on the bf16 matrix pipe with 3-way split operands at fp32 accuracy
                                (csrc/attn_split.hip)                                                                      */
    float* ws;               /* optional scratch (16-byte aligned) for key-split launches, see below; may be NULL       */
    long long ws_bytes;
    int nsplit;              /* set by the launcher                                                                      */
    /* ABI 5: two-part fp16 operand format for the split-operand kernel (csrc/attn_split.hip, NP = 2): q, k, v are multiplied by
       powers of two derived from UPPER BOUNDS of their magnitudes and split into (hi, lo) fp16 parts; three partial products per
       block instead of six.  The bounds must hold (a larger element overflows fp16): either by value, or in device memory
       (f16_amax[0..2] = max|q|, max|k|, max|v|, read by the kernel - graph-capturable, no host round trip).                   */
    int f16x3;               /* 1: use the fp16 format (needs the bounds below); 0: bf16 x 6 (any fp32 input)              */
    float f16_q_amax, f16_k_amax, f16_v_amax;
    const float* f16_amax;   /* optional device array [3]; overrides the by-value bounds                                  */
    void* O2;                /* optional, f16x3 launches only: instead of O, write the output ALREADY SPLIT for the projection that
                                follows - two fp16 parts [2][nbatch*nq][nheads*32] of o times the power of two derived from the v
                                bound (|o| <= max|v|), i.e. pd_gemm_args.A2 with a_amax = &f16_amax[2]; rows are (batch, query)
                                in order, so it needs o_bs = nq * o_ss and o_ss = nheads * 32.  16-byte aligned.              */
    /* ABI 6, f16x3 launches only: K and V already scaled (by the powers of two this kernel derives from f16_amax[1] /
       f16_amax[2]) and split into two fp16 parts by the producing projection (pd_gemm_args.Y2 layout): the high parts of dims
       4 g .. 4 g + 3 of (b, key, h) at K2[b * kv2_bs + key * kv2_ss + 64 h + 8 g ..], the low parts four elements further
       (fp16 elements; 16-byte aligned).  The staging of a key tile is then a copy - every query block of a (batch, head) no
       longer re-splits the same K and V.  K / V (fp32) are ignored when K2 / V2 are given; both or neither.               */
    const void* K2;
    const void* V2;
    long long kv2_bs, kv2_ss;
    /* ABI 7, f16x3 launches only: > 0 = the bias fragments were produced ALREADY MULTIPLIED by this power of two - the product of
       the q and k operand scales the kernel derives from f16_amax[0..1] (pd_attention_bias_prescale_log2 computes its exponent
       from the same bounds on the host; the producer folds it into out_scale).  The pipelined kernel (csrc/attn_pipe.hip) then
       takes the bias tile as the INITIAL VALUE of the score accumulator - no bias add.  0 with a bias: the launch stays on
       attn_parts_kernel (csrc/attn_f16.hip), which adds an unscaled bias; < 0: keep the launch on attn_parts_kernel (A/B runs).   */
    float bias_prescale;
    /* ABI 8: rows of one part plane of O2 when the launch covers only a slice of the samples that share the O2 buffer (the low parts
       sit o2_rows * nheads * 32 elements behind the high parts); 0: nbatch * nq.  pd_attention sets it for its own sub-launches.    */
    long long o2_rows;
} pd_attn_args;
/* Launches that cannot fill the chip (nbatch * nheads * ceil(nq/128) < 512 blocks) with a long key range are split into
 * up to 8 key chunks when ws holds nsplit * nbatch * nq * nheads * 34 floats; a second kernel merges the chunks. */
int pd_attention(const pd_attn_args* args, void* stream);
/* waves per block (4 or 8 = template argument of attn_kernel) pd_attention picks for these arguments; 4 + 100 * nsplit for
 * a key-split launch; 1000 + waves for attn_split_kernel<waves>, 2000 + waves for attn_parts_kernel<waves, 2>, 2000 + 4 +
 * 100 * nsplit for a key-split launch on attn_parts_kernel<4, 2, false, true> (profiling) */
int pd_attention_variant(const pd_attn_args* args);
/* ABI 8 (lab builds with -DPD_ATTN_TAIL=1 only; the shipped library returns 0: measured no gain): a chip-filling launch of the
 * pipelined kernel (variant 3000 +) whose last round of 256-query blocks would be less than half
 * full - 20 samples x 4 heads x 2048 atoms are 640 blocks for 512 slots: the second round runs on a quarter of the chip - hands the
 * samples of that round to the key-split form of attn_parts_kernel instead (their key range cut in `*nsplit` chunks, merged by the
 * combine kernel), so that the tail costs a fraction of a round.  Needs ws (nsplit * tail samples * nq * nheads * 34 floats).  Returns
 * the number of tail samples pd_attention would treat that way (0: none) and the chunk count.                                    */
int pd_attention_tail(const pd_attn_args* args, int* nsplit);
/* log2 of the power of two a bias producer folds into out_scale for f16x3 launches with bias_prescale (ABI 7): the q and k operand
 * scales of the fp16 format for these bounds (scale = 1/sqrt(32)); host arithmetic identical to the kernel's.  3000 + waves =
 * attn_pipe_kernel<waves, ., .> in pd_attention_variant's numbering.                                                        */
int pd_attention_bias_prescale_log2(float q_amax, float k_amax, float scale);

/* ---- pair-representation / pooling kernels (pair.hip) ----------------------------------
 * pd_atom_pair_init : ap = cl_l + cm_m + v*(Wp.d + Wd/(1+|d|) + Wv)   (diffusion_conditioning.py:116-124)
 * pd_pair_gather_add: ap[l,m] += zt[a2t[l], a2t[m]]                    (diffusion_conditioning.py:237)
 * pd_pair_init_z    : z = s_i + s_j + RelPos + bonds                   (diffusion_conditioning.py:65-94,187-189)
 * pd_segment_pool   : token mean of contiguous atom rows (/(n+1e-3))   (transformers.py:205-212)
 * pd_unpool_add     : ba[b,l] += us[b, a2t[l]]                         (transformers.py:214-216)
 * pd_gather_rows_add: y[r] += x[idx[r]]                                (diffusion_conditioning.py:236)
 * pd_axpby          : out = a*sa + b*(sb_ptr ? sb_ptr[0]*sb : sb)
 * pd_template_mask  : z_mask * templ_feat[...,D-1] * same_chain        (diffusion_conditioning.py:41-42)
 * Index tensors keep the loader's dtypes: int64 (uid, a2t, residue_index) / int32 (asym, sym, entity). */
/* pd_atom_pair_ffn  : ap += W2 . (silu(W1 ap) * (W3 ap)) in one pass, c_ap = 16 / hidden = 128 only (other shapes:
 *                      PD_ERR_UNSUPPORTED, use two pd_gemm)         (diffusion_conditioning.py:125-126, feed_forward.py:26-31) */
int pd_atom_pair_ffn(float* ap, const float* W1, const float* W3, const float* W2, long long rows, int c_ap, int hidden,
                     void* stream);
int pd_atom_pair_init(const float* pos, const long long* uid, const float* cl, const float* cm, const float* Wp,
                      const float* Wd, const float* Wv, float* ap, int A, int c_ap, void* stream);
int pd_pair_gather_add(float* ap, const float* zt, const long long* a2t, int A, int T, int c_ap, void* stream);
int pd_pair_init_z(const float* si, const float* sj, const float* WT, const float* wb, const int* asym, const int* sym,
                   const int* ent, const long long* res, const float* rel_tok_feat, const float* bonds, float* z, int T,
                   int CZ, void* stream);
int pd_segment_pool(const float* u, const int* tok_start, const float* add, float* out, int B, int A, int T, int C,
                    void* stream);
int pd_unpool_add(float* ba, const float* us, const long long* a2t, int B, int A, int T, int C, void* stream);
/* pd_downscale_pool (ABI 8): linear_downscale + SiLU + token mean pooling + s of the denoiser in one launch (reference
 * layers/transformers.py:205-212; csrc/pool.hip): out[b,t,:] = sum_{atoms l of t} silu(W ba[b,l,:] + bias) / (n_t + 1e-3) + add[t,:].
 * ba [B][A][128]; W2 / w_inv = the two fp16 parts of W [N][128] (fragment-major) and its inverse row scales (packing.split2_f16) -
 * the A operand's power-of-two scale is the block's own: the maximum of the tile it stages; tok_start [T + 1]; tpb tokens per block
 * (1..32) with the caller's guarantee that tpb consecutive tokens hold at most 64 atoms (the table is device memory: a block that
 * finds more writes NaN into its tokens' rows instead of a wrong mean).  PD_ERR_UNSUPPORTED: other shapes.                         */
int pd_downscale_pool(const float* ba, const void* W2, const float* w_inv, const float* bias, const int* tok_start, const float* add,
                      float* out, int B, int A, int T, int Cin, int N, int tpb, void* stream);
int pd_gather_rows_add(float* y, const float* x, const long long* idx, int R, int C, void* stream);
int pd_axpby(float* out, const float* a, float sa, const float* b, const float* sb_ptr, float sb, long long n, void* stream);
/* pd_template_feat  : templ_feat [T,T,no_bins+1] = [distogram bins of the pseudo-beta distance | mask] * mask, mask = z_mask *
 *                      protein_i * protein_j (feature_loader.py:944-968 inference branch; tensor_utils.py:689-703); lower =
 *                      the no_bins fp32 bin edges linspace(3.25, 50.75, no_bins)^2 computed by the host (slice) */
int pd_template_feat(const float* x, const long long* pseudo_beta_atom, const float* z_mask, const float* is_protein,
                     const float* lower, float* out, int T, int no_bins, void* stream);
int pd_template_mask(const float* z_mask, const float* templ_feat, const int* asym, float* out, int T, int D, void* stream);
/* ConfidenceModule entry / exit passes (confidence.hip; reference models/layers/confidence_module.py:56-88):
 * pd_confidence_pair_init: out[i,j,:] = z[i,j,:] + si[i,:] + sj[j,:] + WdT[bin(|xc_i - xc_j|), :], xc = x[centre[.]], bin =
 *                          nearest of linspace(3.375, 24.375, 13) (first on ties), WdT = linear_d.weight^T [13][C]   (:68-72)
 * pd_pair_symmetrize     : out[i,j,:] = z[i,j,:] + z[j,i,:] (out != z)                                             (:75)
 * pd_atom_dist_embed     : ap[i,j,:] = |x_i - x_j| w + b, Linear(1, c_ap)                                           (:80)     */
int pd_confidence_pair_init(const float* z, const float* si, const float* sj, const float* WdT, const float* x,
                            const long long* centre, float* out, int T, int C, void* stream);
int pd_pair_symmetrize(const float* z, float* out, int T, int C, void* stream);
int pd_atom_dist_embed(const float* x, const float* w, const float* b, float* ap, int A, int C, void* stream);

/* ---- feature tensorisation + PDB writer (features.hip) ------------------------
 * The steps either side of the sampler: FeatureLoader.transform (feature_loader.py:970-998) and
 * FeatureLoader.write_pdb_block (:1230-1283).  Raw per-system arrays in, model feature tensors out; poses in, PDB bytes out.
 * pd_target_feat    : [one_hot(restype, n_class) | profile | deletion_mean] -> [T, n_class + n_profile + 1]      (:805-809)
 * pd_msa_feat       : rows inds[] of (msa, deletion_matrix) -> [n_rows_out, T, n_class + 2] =
 *                     [one_hot | clamp(del,0,1) | atan(del/3) * two_over_pi]; two_over_pi = the host's fp32 2/(2 acos 0) (:813-826)
 * pd_outer_mask     : out[i,j] = m[i] m[j]  (z_mask, ap_mask)                                                     (:982-983)
 * pd_chain_contacts : per chain pair p = (pairs[2p], pairs[2p+1]) (atoms chain_start[c] .. chain_start[c+1]): closest atom pair
 *                     under |xa-xb| + (1 - mask_a mask_b) 1000, first in (a,b) order on ties; below `threshold` the two atoms'
 *                     tokens are set to 1 in between[T,T] (symmetric; caller zeroes it); min_out / arg_out [n_pairs] optional  (:882-900)
 * pd_pdb_format     : out[b, n, 0:81] = tmpl[n, 0:81] with columns 31-54 replaced by x[b, atom[n], 0:3] as "%8.3f" (Python
 *                     float formatting of the fp32 value, ties to even, "-0.000" kept); *overflow counts values that do not fit
 *                     the field (printed as '*')                                                                  (:1259-1270) */
int pd_target_feat(const long long* restype, const float* profile, const float* deletion_mean, float* out, int T, int n_class,
                   int n_profile, void* stream);
int pd_msa_feat(const long long* msa, const float* deletion_matrix, const long long* inds, float two_over_pi, float* out,
                int n_rows_out, int T, int n_class, void* stream);
int pd_outer_mask(const float* m, float* out, int N, void* stream);
int pd_chain_contacts(const float* x, const float* a_mask, const int* chain_start, const int* pairs, int n_pairs,
                      const long long* a2t, float threshold, float* between, int T, float* min_out, long long* arg_out,
                      void* stream);
int pd_pdb_format(const float* x, const unsigned char* tmpl, const int* atom, unsigned char* out, int* overflow, int B, int A,
                  int N, void* stream);

/* ---- per-step sampler kernels (sampler.hip) ------------------------------------------------
 * pd_augment       : centre_random_augmentation + noise injection      (tensor_utils.py:576-586, model.py:70-85)
 *                    parity mode: rot_u[4][B], trans[B][3], noise[B][A][3]; perf mode: Philox(seed, sample0+b, step)
 * pd_init_noise    : x0 = sigma_0 * N(0,1) from Philox                  (model.py:148)
 * pd_precond       : ba = Wx.(x_hat*c_in) + bx + a                      (transformers.py:218-223)
 * pd_denoise       : x_den = c_skip*x_hat + c_out*Wr.LN(ba)             (transformers.py:228-233)
 * pd_kabsch_align  : weighted_rigid_align (moves x_gt onto x_pred)      (tensor_utils.py:724-778)
 * pd_template_match: eps metric, argmin, write template into ref_pos   (model.py:231-241; redocking.py:326-335);
 *                    with eps_out [B][Cn] given the metric runs one workgroup per (conformer, sample), same values
 * pd_pose_dist     : pairwise distances of conformers                   (model.py:186)
 * pd_euler         : d_cur mix + Euler step                             (model.py:245-281)
 * pd_timestep_embed: sinusoidal embedding                               (timestep_embeddings.py:64-81)   */
int pd_augment(const float* x, float x_scale, const float* mask, const float* rot_u, const float* trans,
               const float* noise, float lambda, float sdev, const unsigned long long* seed, int step, int sample0,
               float* out, int B, int A, void* stream);
int pd_init_noise(float* x, const unsigned long long* seed, int sample0, float sigma0, int B, int A, void* stream);
int pd_precond(const float* x_hat, float c_in, const float* c_in_b, const float* Wx, const float* bx, const float* a,
               float* ba, int B, int A, int C, void* stream);
int pd_denoise(const float* ba, const float* x_hat, const float* nw, const float* nb, const float* Wr, float eps,
               float c_skip, float c_out, const float* cs_b, const float* co_b, float* x_den, int B, int A, int C,
               void* stream);
int pd_kabsch_align(const float* x_pred, const float* pred_mask, const float* x_gt, long long gt_bstride, const float* w,
                    float* out, int B, int A, void* stream);
int pd_template_match(const float* x, const int* lig_idx, const float* ref_dist, const float* poses, float* batch_ref_pos,
                      float* eps_out, int* sel_out, int B, int A, int L, int Cn, void* stream);
int pd_pose_dist(const float* poses, float* D, int Cn, int L, void* stream);
/* pairwise RMSD matrix of n poses over the atoms idx[0..L) (NULL: first L atoms) and, if ref != NULL, the RMSD of
 * every pose to ref: the device half of the ranking step (redocking.py:357-423)                  */
int pd_pairwise_rmsd(const float* x, const int* idx, const float* ref, float* D, float* rmsd_ref, int n, int A, int L,
                     void* stream);
int pd_euler(const float* x_hat, const float* x_den, const float* x_proj, const float* w, float t_hat, float eta, float dt,
             float* x_next, int B, int A, void* stream);
int pd_timestep_embed(const float* tau, float* emb, int n, void* stream);
/* pd_dit_bounds (ABI 5; tightened in ABI 8): rigorous magnitude bounds of a DiT block's activations for the two-part fp16 operand
 * format, from the AdaLN table and the weights - no activation is looked at: tab [nrows][ld] holds per DiT block
 * (shift | 1 + scale | gate) x (attention, transition), C channels each; consts [nblocks][4] = (q bound, k bound, -, -);
 * wstack [nblocks][C + 2 hidden][C] = the block's (linear_v | w1 | w3) rows as its projections use them; vh [nrows][nblocks][2]
 * scratch.  out [nrows][nblocks][8] = (|q|, |k|, |v| = |o|, |y|, |y'|, |h|, 0, 0) upper bounds.  ABI 8: |v| and |h| are per-output-row
 * Cauchy-Schwarz bounds with the step's own modulation inside the norm (sqrt(C) ||W_n o (1 + scale)||_2 + |W_n . shift|), maximised
 * over n - a large AdaLN gain or weight row no longer loosens the bound of every channel (csrc/sampler.hip for the derivation;
 * reference adaptive_layer_norm_zero.py:16-21, attentions.py:241-265, transitions.py:27-30).                                     */
int pd_dit_bounds(const float* tab, int nrows, int ld, int nblocks, int C, int hidden, const float* consts, const float* wstack,
                  float* vh, float* out, void* stream);
/* chirality accept / reject of B poses without leaving the device (replaces the per-pose RDKit rebuild + R/S comparison of
 * redocking.py:264-281,303-317): centres[nc][4] = (centre atom, three neighbour atoms), indices into the A atoms of a pose;
 * sign of the signed volume (n1-c).((n2-c)x(n3-c)) vs ref_sign[nc] (+1 / -1); accept[b] = 1 iff every centre matches.
 * sign_out [B][nc] (optional) returns the signs themselves (used once, on the reference coordinates, to make ref_sign).   */
int pd_chirality(const float* x, const int* centres, const int* ref_sign, int* accept, int* sign_out, int B, int A,
                 int n_centres, void* stream);
/* ligand rows of a pose batch, for the relaxation branch (model.py:252-257):
 * pd_ligand_gather : lig[b,l,:] = x[b, lig_idx[l], :]                      (`x_denoised[:, is_ligand_atom]`)
 * pd_ligand_scatter: dst = src with dst[b,a,:] = lig[b, atom_slot[a], :] where atom_slot[a] >= 0
 *                    (`x_ref = deepcopy(x_denoised); x_ref[:, is_ligand_atom] = relaxed`); atom_slot[A] = ligand slot or -1 */
int pd_ligand_gather(const float* x, const int* lig_idx, float* lig, int B, int A, int L, void* stream);
int pd_ligand_scatter(float* dst, const float* src, const float* lig, const int* atom_slot, int B, int A, int L, void* stream);

/* ---- MMFF94 ligand relaxation on the device (mmff.hip) ----------------------------------------
 * replaces the host loop `get_next_step_pos` (models/model.py:26-52): per sample
 * AllChem.MMFFOptimizeMolecule(ref_mol, mmffVariant="MMFF94", maxIters=mmff_iters, ignoreInterfragInteractions=True)
 * (RDKit: ForceField::minimize -> BFGSOpt::minimize over the MMFF94 contribs).  The molecule arrives as plain term
 * tables (device pointers; int32 atom indices into the ligand, float64 parameters):
 *   bond   [n][2] i,j      | kb, r0
 *   angle  [n][3] i,j,k    | ka, theta0 (deg), linear flag (0/1)            j = central atom
 *   strbnd [n][3] i,j,k    | kbaIJK, kbaKJI, r0_ij, r0_kj, theta0
 *   oop    [n][4] i,j,k,l  | koop                                            j central, l the out-of-plane atom
 *   tors   [n][4] i,j,k,l  | V1, V2, V3
 *   vdw_R / vdw_eps / ele_qq [L][L] symmetric dense pair tables: R*_ij, eps_ij (0: pair excluded), q_i q_j / D with the
 *                            0.75 1-4 factor folded in (0: excluded)
 *   inc_ptr [L+1], inc: atom -> incident bonded terms (CSR); entry = kind << 28 | slot << 24 | term index,
 *                       kind 0..4 in the order above, slot = position of the atom in the term's index row          */
typedef struct pd_mmff_terms {
    int n_atoms, n_bond, n_angle, n_strbnd, n_oop, n_tors;
    const int* bond_idx;   const double* bond_par;
    const int* angle_idx;  const double* angle_par;
    const int* strbnd_idx; const double* strbnd_par;
    const int* oop_idx;    const double* oop_par;
    const int* tors_idx;   const double* tors_par;
    const double* vdw_R;   const double* vdw_eps;  const double* ele_qq;
    const int* inc_ptr;    const int* inc;
} pd_mmff_terms;
/* energy[b] (kcal/mol) and grad[b][L][3] of B conformations pos[b][L][3] (float64; either output may be NULL) */
int pd_mmff_energy_grad(const pd_mmff_terms* terms, const double* pos, double* energy, double* grad, int B, void* stream);
/* x_ref = x [B][A][3] with the rows lig_idx[0..L) replaced by their relaxed coordinates (`x_ref = deepcopy(x_denoised);
 * x_ref[:, is_ligand_atom] = get_next_step_pos(...)`, model.py:253-255); max_iters = mmff_iters.
 * ws: B * (9 L^2 + 24 L) float64 of scratch (inverse Hessian + BFGS vectors per sample)                            */
int pd_mmff_relax(const pd_mmff_terms* terms, const float* x, const int* lig_idx, float* x_ref, double* ws,
                  long long ws_doubles, int B, int A, int max_iters, void* stream);

/* ---- hipGraph helpers (api.hip): capture the host-deterministic step loop once, replay it */
int pd_graph_begin(void* stream);
int pd_graph_end(void* stream, void** exec_out);
int pd_graph_launch(void* exec, void* stream);
int pd_graph_destroy(void* exec);

/* ---- library management ------------------------------------------------------------- */
int pd_abi_version(void);
int pd_gemm_args_size(void);  /* sizeof(pd_gemm_args) / sizeof(pd_attn_args) as compiled: a binding checks its own struct  */
int pd_attn_args_size(void);  /* mirror against these before the first call (a short struct would be read past its end)   */
int pd_init(void);            /* sets per-kernel LDS limits; call once before graph capture */
int pd_attention_occupancy(void);   /* diagnostic: resident attention blocks per CU (runtime's figure) */

#ifdef __cplusplus
}
#endif
#endif
