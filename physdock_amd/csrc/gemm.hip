// fp32 MFMA GEMM with fused normalisation prologue and gating/residual epilogues.
//
//   Y = epilogue( prologue(A)[M,K] . W[N,K]^T )          (torch nn.Linear convention)
//
// One kernel family serves every dense contraction of the hot path:
//   * all Linear layers (reference primitives/linear.py:146-161) with the preceding
//     RMSNorm / LayerNorm / AdaLN-Zero folded into the A-tile staging
//     (rms_norm.py:14-19, adaptive_layer_norm_zero.py:18-21),
//   * SwiGLU (feed_forward.py:30-31) and the sigmoid-gated triangle projections
//     (attentions.py:161-162) as paired-tile "GLU" epilogues,
//   * the triangle-multiplication einsum (attentions.py:164) as a 32-channel batched GEMM,
//   * the outer-product-mean einsum (outer_product_mean.py:28),
//   * pair-bias projections written straight into the attention kernel's fragment layout.
//
// Arithmetic is exact fp32 on v_mfma_f32_32x32x2_f32 (157 TF peak = the roofline of this
// path; parity to 1e-3 A forbids reduced precision).  Block = 256 threads = 4 waves; LDS
// tiles are [rows][32+4] floats so that one ds_read_b128 per lane feeds four MFMA k-steps
// (lane half h takes k = 8*g + 4*h + e, the same permutation on both operands).
#include "common.h"
#include "physdock_hip.h"

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;   // padded row (floats): 144 B, keeps ds_read_b128 conflict-free
constexpr int NT = 256;

struct RowStat { float mean, rstd; };

template <int BM, int BN, int WM, int WN, bool AKM, bool WKM>
__global__ __launch_bounds__(NT) void gemm_kernel(const pd_gemm_args p) {
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int PADM = 4;
    constexpr int A_TILE = AKM ? BK * (BM + PADM) : BM * LDK;
    constexpr int W_TILE = WKM ? BK * (BN + PADM) : BN * LDK;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                 // 2 stages
    float* sW = smem + 2 * A_TILE;    // 2 stages

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hh = lane >> 5;
    const int bm0 = blockIdx.x * BM, bn0 = blockIdx.y * BN;
    const int bz = blockIdx.z;
    const float* __restrict__ A = p.A + (long long)bz * p.sA;
    const float* __restrict__ W = p.W + (long long)bz * p.sW;
    float* __restrict__ Y = p.Y + (long long)bz * p.sY;

    // ---- per-thread load slots -------------------------------------------------------
    constexpr int A_SLOTS = BM / 32, W_SLOTS = BN / 32;   // float4 per thread per k-tile
    // non-k-major: slot i -> row (tid>>3)+32*i, k-chunk (tid&7)
    // k-major    : chunks per k-row = R/4; slot i -> k-row tid/(R/4) + (NT/(R/4))*i, m-chunk tid%(R/4)
    f32x4 ra[A_SLOTS], rw[W_SLOTS];

    // prologue state for A (fixed rows per thread)
    const bool pro = p.stats != nullptr;
    RowStat st[AKM ? 4 : A_SLOTS];
    int grp_off[AKM ? 4 : A_SLOTS];
    if (pro) {
        const int nst = AKM ? 4 : A_SLOTS;
#pragma unroll
        for (int i = 0; i < nst; ++i) {
            int m = AKM ? bm0 + (tid % (BM / 4)) * 4 + i : bm0 + (tid >> 3) + 32 * i;
            st[i].mean = 0.f; st[i].rstd = 0.f; grp_off[i] = 0;
            if (m < p.M) {
                st[i].mean = p.stats[2 * ((long long)bz * p.M + m)];
                st[i].rstd = p.stats[2 * ((long long)bz * p.M + m) + 1];
                if (p.pro_rows_per_group > 0) grp_off[i] = (m / p.pro_rows_per_group) * p.pro_gstride;
            }
        }
    }

    auto load_A = [&](int k0) {
        if constexpr (!AKM) {
            const int kc = k0 + (tid & 7) * 4;
#pragma unroll
            for (int i = 0; i < A_SLOTS; ++i) {
                const int m = bm0 + (tid >> 3) + 32 * i;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (m < p.M) {
                    const float* src = A + (long long)m * p.lda + kc;
                    if (p.vecA && kc + 3 < p.K) v = *reinterpret_cast<const f32x4*>(src);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (kc + e < p.K) v[e] = src[e];
                    }
                    if (pro) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (kc + e < p.K) {
                                float pw = p.pro_w ? p.pro_w[grp_off[i] + kc + e] : 1.f;
                                float pb = p.pro_b ? p.pro_b[grp_off[i] + kc + e] : 0.f;
                                v[e] = (v[e] - st[i].mean) * st[i].rstd * pw + pb;
                            }
                        }
                    }
                    if (p.pro_act) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = pd_act(v[e], p.pro_act);
                    }
                }
                ra[i] = v;
            }
        } else {
            constexpr int CPR = BM / 4, KSTEP = NT / CPR;
            const int mc = bm0 + (tid % CPR) * 4;
#pragma unroll
            for (int i = 0; i < A_SLOTS; ++i) {
                const int k = k0 + tid / CPR + KSTEP * i;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (k < p.K) {
                    const float* src = A + (long long)k * p.lda + mc;
                    if (p.vecA && mc + 3 < p.M) v = *reinterpret_cast<const f32x4*>(src);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (mc + e < p.M) v[e] = src[e];
                    }
                    if (pro) {
                        float pw = p.pro_w ? p.pro_w[k] : 1.f;
                        float pb = p.pro_b ? p.pro_b[k] : 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (v[e] - st[e].mean) * st[e].rstd * pw + pb;
                    }
                }
                ra[i] = v;
            }
        }
    };
    auto load_W = [&](int k0) {
        if constexpr (!WKM) {
            const int kc = k0 + (tid & 7) * 4;
#pragma unroll
            for (int i = 0; i < W_SLOTS; ++i) {
                const int n = bn0 + (tid >> 3) + 32 * i;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (n < p.N) {
                    const float* src = W + (long long)n * p.ldw + kc;
                    if (p.vecW && kc + 3 < p.K) v = *reinterpret_cast<const f32x4*>(src);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (kc + e < p.K) v[e] = src[e];
                    }
                }
                rw[i] = v;
            }
        } else {
            constexpr int CPR = BN / 4, KSTEP = NT / CPR;
            const int nc = bn0 + (tid % CPR) * 4;
#pragma unroll
            for (int i = 0; i < W_SLOTS; ++i) {
                const int k = k0 + tid / CPR + KSTEP * i;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (k < p.K) {
                    const float* src = W + (long long)k * p.ldw + nc;
                    if (p.vecW && nc + 3 < p.N) v = *reinterpret_cast<const f32x4*>(src);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (nc + e < p.N) v[e] = src[e];
                    }
                }
                rw[i] = v;
            }
        }
    };
    auto store_tiles = [&](int stage) {
        float* a = sA + stage * A_TILE;
        float* w = sW + stage * W_TILE;
        if constexpr (!AKM) {
#pragma unroll
            for (int i = 0; i < A_SLOTS; ++i)
                *reinterpret_cast<f32x4*>(a + ((tid >> 3) + 32 * i) * LDK + (tid & 7) * 4) = ra[i];
        } else {
            constexpr int CPR = BM / 4, KSTEP = NT / CPR;
#pragma unroll
            for (int i = 0; i < A_SLOTS; ++i)
                *reinterpret_cast<f32x4*>(a + (tid / CPR + KSTEP * i) * (BM + PADM) + (tid % CPR) * 4) = ra[i];
        }
        if constexpr (!WKM) {
#pragma unroll
            for (int i = 0; i < W_SLOTS; ++i)
                *reinterpret_cast<f32x4*>(w + ((tid >> 3) + 32 * i) * LDK + (tid & 7) * 4) = rw[i];
        } else {
            constexpr int CPR = BN / 4, KSTEP = NT / CPR;
#pragma unroll
            for (int i = 0; i < W_SLOTS; ++i)
                *reinterpret_cast<f32x4*>(w + (tid / CPR + KSTEP * i) * (BN + PADM) + (tid % CPR) * 4) = rw[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    load_A(0); load_W(0);
    store_tiles(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) { load_A((kt + 1) * BK); load_W((kt + 1) * BK); }
        const float* a = sA + cur * A_TILE;
        const float* w = sW + cur * W_TILE;
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            f32x4 fa[TM], fw[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * (TM * 32) + i * 32 + l31;
                if constexpr (!AKM) fa[i] = *reinterpret_cast<const f32x4*>(a + row * LDK + g * 8 + 4 * hh);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) fa[i][e] = a[(g * 8 + 4 * hh + e) * (BM + PADM) + row];
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * (TN * 32) + j * 32 + l31;
                if constexpr (!WKM) fw[j] = *reinterpret_cast<const f32x4*>(w + row * LDK + g * 8 + 4 * hh);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) fw[j][e] = w[(g * 8 + 4 * hh + e) * (BN + PADM) + row];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fw[j][e], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue ------------------------------------------------------------------
    const int glu = p.glu;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (glu && (j & 1)) continue;
            const int ncol_packed = bn0 + wn * (TN * 32) + j * 32;     // packed column base of this tile
            const int n_in = ncol_packed + l31;                        // packed column (bias index)
            int n_out = n_in;
            if (glu) n_out = (bn0 + wn * (TN * 32)) / 2 + (j >> 1) * 32 + l31;
            const int N_out = glu ? p.N / 2 : p.N;
            const bool ncol_ok = n_out < N_out;
            float bias_a = 0.f, bias_b = 0.f;
            if (p.bias) {
                const float* bp = p.bias + (long long)bz * p.sBias;
                if (n_in < p.N) bias_a = bp[n_in];
                if (glu && n_in + 32 < p.N) bias_b = bp[n_in + 32];
            }
            const bool headnorm = p.hn_w != nullptr && ncol_packed < p.hn_cols;
            float hn_w = 0.f;
            if (headnorm) hn_w = p.hn_w[(ncol_packed / p.hn_split) * 32 + l31];
            float vals[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = bm0 + wm * (TM * 32) + i * 32 + pd_frag_row(r, hh);
                float v = acc[i][j][r];
                if (p.rowscale_acc) v *= (m < p.M) ? p.rowscale_acc[(long long)bz * p.M + m] : 0.f;
                v += bias_a;
                if (headnorm) {
                    float ss = v * v;
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
                    v = v * rsqrtf(ss * (1.0f / 32.0f) + p.hn_eps) * hn_w;
                }
                if (glu) {
                    float b2 = 0.f;
                    if constexpr (TN >= 2) b2 = acc[i][j | 1][r] + bias_b;
                    v = (glu == 1) ? pd_silu(v) * b2 : v * pd_sigmoid(b2);
                } else {
                    v = pd_act(v, p.act);
                }
                vals[r] = v;
            }
            // row-dependent post-ops + store
            if (p.out_mode == PD_OUT_ROWMAJOR || p.out_mode == PD_OUT_OPM || p.out_mode == PD_OUT_BIASFRAG) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = bm0 + wm * (TM * 32) + i * 32 + pd_frag_row(r, hh);
                    if (m >= p.M || !ncol_ok) continue;
                    float v = vals[r];
                    if (p.rowscale) v *= p.rowscale[m];
                    if (p.maskadd && p.maskadd[m] == 0.f) v += p.maskval;
                    if (p.mul) {
                        long long mi = p.mul_rows_per_group > 0
                            ? (long long)(m / p.mul_rows_per_group) * p.mul_gstride
                            : (long long)m * p.ldmul;
                        v *= p.mul[mi + n_out];
                    }
                    v *= p.out_scale;
                    if (p.res) {
                        const int mr = p.res_row_mod > 0 ? m % p.res_row_mod : m;
                        v += p.res[(long long)bz * p.sRes + (long long)mr * p.ldres + n_out];
                    }
                    if (p.out_mode == PD_OUT_ROWMAJOR) {
                        Y[(long long)m * p.ldy + n_out] = v;
                    } else if (p.out_mode == PD_OUT_OPM) {
                        // rows (i,c), cols (j,d) -> [i][j][c][d]   (outer_product_mean.py:28)
                        const int ii = m >> 5, c = m & 31, jj = n_out >> 5, d = n_out & 31;
                        Y[(((long long)ii * p.T2 + jj) * 32 + c) * 32 + d] = v;
                    } else {
                        // attention-bias fragment layout (see attention.hip); n_out = head
                        int qi = m / p.T2, kj = m % p.T2, nq = p.T1, nkk = p.T2;
                        if (p.frag_transpose) { int t = qi; qi = kj; kj = t; nq = p.T2; nkk = p.T1; }
                        const int nqt = (nq + 31) >> 5, nkt = (nkk + 31) >> 5;
                        const long long base = (((long long)n_out * nqt + (qi >> 5)) * nkt + (kj >> 5)) * 1024;
                        const int k5 = kj & 31;
                        Y[base + (k5 >> 3) * 256 + ((qi & 31) + 32 * ((k5 >> 2) & 1)) * 4 + (k5 & 3)] = v;
                    }
                }
            } else {   // PD_OUT_TRANSPOSED: Y[n][m], 4 consecutive m per register group
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int m0 = bm0 + wm * (TM * 32) + i * 32 + 8 * g4 + 4 * hh;
                    if (!ncol_ok) continue;
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = vals[4 * g4 + e];
                        const int m = m0 + e;
                        if (m < p.M) {
                            if (p.rowscale) v *= p.rowscale[m];
                            v *= p.out_scale;
                        }
                        o[e] = v;
                    }
                    float* dst = Y + (long long)n_out * p.ldy + m0;
                    if (m0 + 3 < p.M && (p.ldy & 3) == 0 && ((uintptr_t)Y & 15) == 0) *reinterpret_cast<f32x4*>(dst) = o;
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (m0 + e < p.M) dst[e] = o[e];
                    }
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, bool AKM, bool WKM>
int launch(const pd_gemm_args& p, hipStream_t s) {
    constexpr int PADM = 4;
    constexpr int A_TILE = AKM ? BK * (BM + PADM) : BM * LDK;
    constexpr int W_TILE = WKM ? BK * (BN + PADM) : BN * LDK;
    const size_t lds = 2 * (A_TILE + W_TILE) * sizeof(float);
    dim3 grid((p.M + BM - 1) / BM, (p.N + BN - 1) / BN, p.batch > 0 ? p.batch : 1);
    auto k = gemm_kernel<BM, BN, WM, WN, AKM, WKM>;
    hipLaunchKernelGGL(k, grid, dim3(NT), lds, s, p);
    return pd_check_launch();
}

template <int BM, int BN, int WM, int WN, bool AKM, bool WKM>
int set_lds_limit() {
    constexpr int PADM = 4;
    constexpr int A_TILE = AKM ? BK * (BM + PADM) : BM * LDK;
    constexpr int W_TILE = WKM ? BK * (BN + PADM) : BN * LDK;
    const int lds = 2 * (A_TILE + W_TILE) * (int)sizeof(float);
    auto k = gemm_kernel<BM, BN, WM, WN, AKM, WKM>;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess
               ? PD_OK : PD_ERR_LAUNCH;
}

}  // namespace

PD_EXPORT int pd_init(void) {
    int rc = PD_OK;
    rc |= set_lds_limit<128, 128, 2, 2, false, false>();
    rc |= set_lds_limit<128, 128, 2, 2, true, false>();
    rc |= set_lds_limit<128, 128, 2, 2, true, true>();
    rc |= set_lds_limit<128, 64, 2, 2, false, false>();
    rc |= set_lds_limit<128, 32, 4, 1, false, false>();
    rc |= set_lds_limit<64, 64, 2, 2, false, false>();
    return rc ? PD_ERR_LAUNCH : PD_OK;
}

PD_EXPORT int pd_gemm(const pd_gemm_args* args, void* stream) {
    if (!args || !args->A || !args->W || !args->Y) return PD_ERR_ARG;
    pd_gemm_args p = *args;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return PD_ERR_ARG;
    if (p.batch <= 0) p.batch = 1;
    if (p.out_scale == 0.f) p.out_scale = 1.f;
    if (p.glu && (p.N % 64 != 0)) return PD_ERR_ARG;
    if (p.hn_w && (p.hn_split % 32 != 0 || p.hn_cols % 32 != 0)) return PD_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const bool akm = p.a_kmajor != 0, wkm = p.w_kmajor != 0;
    // vector-load eligibility (16-byte alignment of every row start)
    p.vecA = (((uintptr_t)p.A & 15) == 0) && (p.lda % 4 == 0) && (p.sA % 4 == 0) && ((akm ? 0 : p.K % 4) == 0);
    p.vecW = (((uintptr_t)p.W & 15) == 0) && (p.ldw % 4 == 0) && (p.sW % 4 == 0) && ((wkm ? 0 : p.K % 4) == 0);
    if (akm != wkm) {
        if (akm && !wkm) return launch<128, 128, 2, 2, true, false>(p, s);
        return PD_ERR_UNSUPPORTED;
    }
    if (akm) return launch<128, 128, 2, 2, true, true>(p, s);
    const long long blocks128 = (long long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.batch;
    if (p.glu) return launch<128, 128, 2, 2, false, false>(p, s);
    if (p.N <= 32) return launch<128, 32, 4, 1, false, false>(p, s);
    if (blocks128 < 192) return launch<64, 64, 2, 2, false, false>(p, s);
    if (p.N <= 64) return launch<128, 64, 2, 2, false, false>(p, s);
    return launch<128, 128, 2, 2, false, false>(p, s);
}
