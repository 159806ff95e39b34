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

#define PD_ABI_VERSION 1

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
} pd_gemm_args;
int pd_gemm(const pd_gemm_args* args, void* stream);

/* ---- pd_rowstats: per-row (mean, rstd) for the GEMM prologue --------------------------
 * mode 0: RMS  -> (0, rsqrt(mean(x^2)+eps));  mode 1: LayerNorm -> (mean, rsqrt(var+eps)).
 * x is [M,C] (ldx) or, if kmajor, [C,M] (ldx = M stride).                               */
int pd_rowstats(const float* x, float* stats, int M, int C, int ldx, int kmajor, int mode, float eps, void* stream);

/* ---- pd_rownorm: y = [res +] act(norm(x) * w + b) (standalone normalisation)            */
int pd_rownorm(const float* x, float* y, const float* res, const float* w, const float* b,
               int M, int C, int mode, float eps, int act, void* stream);

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
} pd_attn_args;
int pd_attention(const pd_attn_args* args, void* stream);

/* ---- library management ------------------------------------------------------------- */
int pd_abi_version(void);
int pd_init(void);            /* sets per-kernel LDS limits; call once before graph capture */

#ifdef __cplusplus
}
#endif
#endif
