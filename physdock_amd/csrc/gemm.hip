// fp32 MFMA GEMM with fused normalisation prologue and gating/residual epilogues.
//
//   Y = epilogue( prologue(A)[M,K] . W[N,K]^T )          (torch nn.Linear convention)
//
// One kernel family serves every dense contraction of the hot path:
//   * all Linear layers (reference primitives/linear.py:146-161) with the preceding
//     RMSNorm / LayerNorm / AdaLN-Zero folded into the A-tile staging
//     (rms_norm.py:14-19, adaptive_layer_norm_zero.py:18-21),
//   * SwiGLU (feed_forward.py:30-31) and the sigmoid-gated triangle projections
//     (attentions.py:161-162) as paired-column "GLU" epilogues,
//   * the triangle-multiplication einsum (attentions.py:164) as a 32-channel batched GEMM,
//   * the outer-product-mean einsum (outer_product_mean.py:28),
//   * pair-bias projections written straight into the attention kernel's fragment layout.
//
// Arithmetic is exact fp32 on v_mfma_f32_32x32x2_f32 (157 TF peak = the roofline of this
// path; parity to 1e-3 A forbids reduced precision).  Block = 256 threads = 4 waves.
//   main loop : global -> registers (branch-free, clamped addresses) is issued BEFORE the MFMA
//               block of the current k-tile and written to the other LDS stage AFTER it (the
//               norm prologue is applied in that write), one barrier per k-tile.  LDS tiles are
//               [rows][32+4] floats so one ds_read_b128 per lane feeds four MFMA k-steps (lane
//               half h takes k = 8g+4h+e, the same permutation on both operands).
//   epilogue  : accumulators are parked in LDS (re-using the stage buffers) and re-read
//               row-major, so bias / gate / residual / output traffic is 16-byte coalesced and
//               the epilogue is one compact rolled loop whatever the fusion flags.
#include <stdlib.h>
#include "common.h"
#include "physdock_hip.h"

namespace {

#ifndef PD_BK
#define PD_BK 32
#endif
constexpr int BK = PD_BK;     // k-slice depth (32 or 16)
constexpr int LDK = BK + 4;   // padded row (floats): 144 B / 80 B, keeps ds_read_b128 conflict-free
constexpr int CH = BK / 4;    // 16-byte chunks per staged row
constexpr int RPP = 256 / CH; // rows staged per pass of the 256 threads
constexpr int NT = 256;
constexpr int PADM = 4;

#ifdef PD_LAB      // lab build only (tools/gemm_trace.py): in-kernel phase trace buffer + host-side tuning overrides
__device__ unsigned long long* g_gemm_trace = nullptr;
bool g_gemm_trace_on = false;
#define PD_TRACE_PTR g_gemm_trace
#else
#define PD_TRACE_PTR ((unsigned long long*)nullptr)
#endif

template <int BM, int BN, bool AKM, bool WKM>
struct Cfg {
    static constexpr int A_TILE = AKM ? BK * (BM + PADM) : BM * LDK;
    static constexpr int W_TILE = WKM ? BK * (BN + PADM) : BN * LDK;
    static constexpr int LDC = BN + 4;
    static constexpr int STAGE_FLOATS = 2 * (A_TILE + W_TILE);
    static constexpr int CPASS = (BM * LDC > STAGE_FLOATS && BM >= 128) ? 2 : 1;   // park the accumulators in row slabs
    static constexpr int C_FLOATS = (BM / CPASS) * LDC;
    static constexpr int LDS_BYTES = 4 * (STAGE_FLOATS > C_FLOATS ? STAGE_FLOATS : C_FLOATS);
};

// One operand tile loader: R rows (m or n) x BK, either [R][K] (k contiguous) or k-major [K][R].
template <int R, bool KM, bool VEC>
struct TileLoader {
    static constexpr int CPR = R / 4;                  // k-major: 16-byte chunks per k-row
    static constexpr int KSTEP = NT / CPR;
    static constexpr int SLOTS = KM ? BK / KSTEP : R / RPP;   // float4 per thread per k-slice
    f32x4 reg[SLOTS];

    __device__ __forceinline__ void load(const float* __restrict__ base, int ld, int r0, int rows, int k0, int K, int tid) {
        if constexpr (!KM) {
            const int kc = k0 + (tid % CH) * 4;
            const int kcl = kc < K ? kc : 0;
#pragma unroll
            for (int i = 0; i < SLOTS; ++i) {
                int r = r0 + tid / CH + RPP * i;
                r = r < rows ? r : rows - 1;
                const float* src = base + (long long)r * ld + kcl;
                if constexpr (VEC) reg[i] = *reinterpret_cast<const f32x4*>(src);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) reg[i][e] = src[kcl + e < K ? e : 0];
                }
            }
        } else {
            const int rc = r0 + (tid % CPR) * 4;
            const int rcl = rc < rows ? rc : 0;
#pragma unroll
            for (int i = 0; i < SLOTS; ++i) {
                int k = k0 + tid / CPR + KSTEP * i;
                k = k < K ? k : K - 1;
                reg[i] = *reinterpret_cast<const f32x4*>(base + (long long)k * ld + rcl);
            }
        }
    }
    // zero the out-of-range elements (rows >= rows_total, k >= K); interior tiles skip it (block-uniform test)
    __device__ __forceinline__ void mask(int r0, int rows, int k0, int K, int tid) {
        if (r0 + R <= rows && k0 + BK <= K) return;
        if constexpr (!KM) {
            const int kc = k0 + (tid % CH) * 4;
#pragma unroll
            for (int i = 0; i < SLOTS; ++i) {
                const bool rok = r0 + tid / CH + RPP * i < rows;
#pragma unroll
                for (int e = 0; e < 4; ++e) reg[i][e] = (rok && kc + e < K) ? reg[i][e] : 0.f;
            }
        } else {
            const int rc = r0 + (tid % CPR) * 4;
#pragma unroll
            for (int i = 0; i < SLOTS; ++i) {
                const bool kok = k0 + tid / CPR + KSTEP * i < K;
#pragma unroll
                for (int e = 0; e < 4; ++e) reg[i][e] = (kok && rc + e < rows) ? reg[i][e] : 0.f;
            }
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ s, int tid) const {
        if constexpr (!KM) {
#pragma unroll
            for (int i = 0; i < SLOTS; ++i)
                *reinterpret_cast<f32x4*>(s + (tid / CH + RPP * i) * LDK + (tid % CH) * 4) = reg[i];
        } else {
#pragma unroll
            for (int i = 0; i < SLOTS; ++i)
                *reinterpret_cast<f32x4*>(s + (tid / CPR + KSTEP * i) * (R + PADM) + (tid % CPR) * 4) = reg[i];
        }
    }
};

template <int BM, int BN, int WM, int WN, bool AKM, bool WKM, bool VEC, int PRO>
__global__ __launch_bounds__(NT) void gemm_kernel(const pd_gemm_args p) {
    using C_ = Cfg<BM, BN, AKM, WKM>;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int A_TILE = C_::A_TILE, W_TILE = C_::W_TILE, LDC = C_::LDC;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                 // 2 stages
    float* sW = smem + 2 * A_TILE;    // 2 stages

    const unsigned long long t_entry = PD_TRACE_PTR ? __builtin_amdgcn_s_memtime() : 0ull;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hh = lane >> 5;
    const int bm0 = blockIdx.x * BM, bn0 = blockIdx.y * BN;
    const int bz = blockIdx.z;
    const float* __restrict__ A = p.A + (long long)bz * p.sA;
    const float* __restrict__ W = p.W + (long long)bz * p.sW;
    float* __restrict__ Y = p.Y + (long long)bz * p.sY;

    TileLoader<BM, AKM, VEC> la;
    TileLoader<BN, WKM, VEC> lw;

    // ---- prologue state: this thread always stages the same rows of A ------------------
    // PRO 0: none   1: pro_w/pro_b shared by all rows   2: per row group (AdaLN with per-sample t)
    constexpr int ASLOTS = TileLoader<BM, AKM, VEC>::SLOTS;
    constexpr int NST = AKM ? 4 : ASLOTS;
    float st_mean[NST], st_rstd[NST];
    int grp_off[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) { st_mean[i] = 0.f; st_rstd[i] = 1.f; grp_off[i] = 0; }
    if constexpr (PRO != 0) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            int m = AKM ? bm0 + (tid % (BM / 4)) * 4 + i : bm0 + tid / CH + RPP * i;
            m = m < p.M ? m : p.M - 1;
            st_mean[i] = p.stats[2 * ((long long)bz * p.M + m)];
            st_rstd[i] = p.stats[2 * ((long long)bz * p.M + m) + 1];
            if constexpr (PRO == 2) grp_off[i] = (m / p.pro_rows_per_group) * p.pro_gstride;
        }
    }
    auto transform_A = [&](int k0) {
        if constexpr (PRO != 0) {
            if constexpr (!AKM) {
                int kc = k0 + (tid % CH) * 4;
                kc = kc < p.K ? kc : 0;
                f32x4 pw, pb;
                if constexpr (PRO == 1) {
                    pw = *reinterpret_cast<const f32x4*>(p.pro_w + kc);
                    pb = *reinterpret_cast<const f32x4*>(p.pro_b + kc);
                }
#pragma unroll
                for (int i = 0; i < ASLOTS; ++i) {
                    if constexpr (PRO == 2) {
                        pw = *reinterpret_cast<const f32x4*>(p.pro_w + grp_off[i] + kc);
                        pb = *reinterpret_cast<const f32x4*>(p.pro_b + grp_off[i] + kc);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        la.reg[i][e] = (la.reg[i][e] - st_mean[i]) * st_rstd[i] * pw[e] + pb[e];
                }
            } else {
                constexpr int CPR = BM / 4, KSTEP = NT / CPR;
#pragma unroll
                for (int i = 0; i < ASLOTS; ++i) {
                    int k = k0 + tid / CPR + KSTEP * i;
                    k = k < p.K ? k : 0;
                    const float pw = p.pro_w[k], pb = p.pro_b[k];
#pragma unroll
                    for (int e = 0; e < 4; ++e) la.reg[i][e] = (la.reg[i][e] - st_mean[e]) * st_rstd[e] * pw + pb;
                }
            }
        }
        if (p.pro_act == PD_ACT_RELU) {
#pragma unroll
            for (int i = 0; i < ASLOTS; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) la.reg[i][e] = fmaxf(la.reg[i][e], 0.f);
        } else if (p.pro_act == PD_ACT_SILU) {
#pragma unroll
            for (int i = 0; i < ASLOTS; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) la.reg[i][e] = pd_silu(la.reg[i][e]);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Row-major epilogue: a thread always handles the same 4 output columns (NT is a multiple of the chunks per
    // row), so bias / head-norm weights are fetched ONCE, here, and their latency disappears under the main loop.
    const int e_glu = p.glu;
    const int e_cpr = (e_glu ? BN / 2 : BN) / 4;
    const int e_c = tid % e_cpr;
    const int e_pc = e_glu ? ((e_c * 4) >> 5) * 64 + ((e_c * 4) & 31) : e_c * 4;      // packed column in Cs / bias
    f32x4 e_bias = {0.f, 0.f, 0.f, 0.f}, e_bias2 = {0.f, 0.f, 0.f, 0.f}, e_hw = {1.f, 1.f, 1.f, 1.f};
    const bool e_hn = p.hn_w != nullptr && bn0 + e_pc < p.hn_cols;
    if (p.out_mode == PD_OUT_ROWMAJOR) {
        if (p.bias) {
            const float* bp = p.bias + (long long)bz * p.sBias;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (bn0 + e_pc + e < p.N) e_bias[e] = bp[bn0 + e_pc + e];
                if (e_glu && bn0 + e_pc + 32 + e < p.N) e_bias2[e] = bp[bn0 + e_pc + 32 + e];
            }
        }
        if (e_hn) {
            const float* hw = p.hn_w + ((bn0 + e_pc) / p.hn_split) * 32 + ((bn0 + e_pc) & 31);
#pragma unroll
            for (int e = 0; e < 4; ++e) e_hw[e] = hw[e];
        }
    }

    const int nk = (p.K + BK - 1) / BK;
    la.load(A, p.lda, bm0, p.M, 0, p.K, tid);
    lw.load(W, p.ldw, bn0, p.N, 0, p.K, tid);
    transform_A(0);
    la.mask(bm0, p.M, 0, p.K, tid);
    lw.mask(bn0, p.N, 0, p.K, tid);
    la.store(sA, tid);
    lw.store(sW, tid);
    __syncthreads();

    // optional in-kernel phase trace (debug): lane 0 of every wave of blocks < 64 stamps the shader clock
    unsigned long long* dbg = nullptr;
    if (PD_TRACE_PTR && lane == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 64)
        dbg = PD_TRACE_PTR + ((long long)blockIdx.x * 4 + wave) * (5 * 64);
#define PD_STAMP(slot) if (dbg && kt < 60) dbg[kt * 5 + slot] = __builtin_amdgcn_s_memtime()
    if (dbg) { dbg[60 * 5 + 0] = t_entry; dbg[60 * 5 + 1] = __builtin_amdgcn_s_memtime(); }
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        PD_STAMP(0);
        if (more) {
            la.load(A, p.lda, bm0, p.M, (kt + 1) * BK, p.K, tid);
            lw.load(W, p.ldw, bn0, p.N, (kt + 1) * BK, p.K, tid);
        }
        PD_STAMP(1);
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
        PD_STAMP(2);
        if (more) {
            transform_A((kt + 1) * BK);
            la.mask(bm0, p.M, (kt + 1) * BK, p.K, tid);
            lw.mask(bn0, p.N, (kt + 1) * BK, p.K, tid);
            la.store(sA + (cur ^ 1) * A_TILE, tid);
            lw.store(sW + (cur ^ 1) * W_TILE, tid);
        }
        PD_STAMP(3);
        __syncthreads();
        PD_STAMP(4);
    }
#undef PD_STAMP
    if (dbg) dbg[60 * 5 + 2] = __builtin_amdgcn_s_memtime();

    // ---- park the accumulators in LDS one row slab at a time (stage buffers are free after the last barrier)
    float* Cs = smem;
    constexpr int CPASS = C_::CPASS, SLAB = BM / CPASS;

    // ---- epilogue ------------------------------------------------------------------------
    const int glu = p.glu;
    const int N_out = glu ? p.N / 2 : p.N;
    const int BN_out = glu ? BN / 2 : BN;
    const int bn_out0 = glu ? bn0 / 2 : bn0;
    const float* biasp = p.bias ? p.bias + (long long)bz * p.sBias : nullptr;
    const float* resp = p.res ? p.res + (long long)bz * p.sRes : nullptr;

  for (int pass = 0; pass < CPASS; ++pass) {
    const int slab0 = pass * SLAB;
    if (pass) __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rbase = wm * (TM * 32) + i * 32;
        if (rbase >= slab0 && rbase < slab0 + SLAB) {      // wave-uniform
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Cs[(rbase - slab0 + pd_frag_row(r, hh)) * LDC + wn * (TN * 32) + j * 32 + l31] = acc[i][j][r];
        }
    }
    if (dbg && pass == 0) dbg[61 * 5 + 0] = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (dbg && pass == 0) dbg[61 * 5 + 1] = __builtin_amdgcn_s_memtime();
    if (p.out_mode == PD_OUT_ROWMAJOR) {
        // U row-chunks per trip: all gate / residual loads of a trip are issued before the first store, so the
        // epilogue keeps U independent 16-byte loads in flight per lane instead of one dependent round trip per row.
        constexpr int U = 4;
        const int cpr = BN_out / 4;
        const int total = SLAB * cpr;          // rows of this slab x chunks
#pragma unroll 1
        for (int base = tid; base < total; base += NT * U) {
            if (dbg && base == tid + NT * U) dbg[61 * 5 + 2] = __builtin_amdgcn_s_memtime();
            f32x4 v[U], mulv[U], resv[U];
            int mrow[U], ncol[U];
            bool ok[U], full[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * NT;
                const int row = idx / cpr, c = idx - row * cpr;
                const int m = bm0 + slab0 + row, n = bn_out0 + c * 4;
                mrow[u] = m; ncol[u] = n;
                ok[u] = idx < total && m < p.M && n < N_out;
                full[u] = p.vecY && n + 3 < N_out;
                mulv[u] = f32x4{1.f, 1.f, 1.f, 1.f};
                resv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (ok[u]) {
                    if (p.mul) {
                        const float* mp = p.mul + (p.mul_rows_per_group > 0
                                                       ? (long long)(m / p.mul_rows_per_group) * p.mul_gstride
                                                       : (long long)m * p.ldmul) + n;
                        if (full[u]) mulv[u] = *reinterpret_cast<const f32x4*>(mp);
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (n + e < N_out) mulv[u][e] = mp[e];
                        }
                    }
                    if (resp) {
                        const int mr = p.res_row_mod > 0 ? m % p.res_row_mod : m;
                        const float* rp = resp + (long long)mr * p.ldres + n;
                        if (full[u]) resv[u] = *reinterpret_cast<const f32x4*>(rp);
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (n + e < N_out) resv[u][e] = rp[e];
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * NT;
                const int row = idx / cpr;
                const int m = mrow[u];
                f32x4 x = *reinterpret_cast<const f32x4*>(Cs + row * LDC + e_pc);
                if (p.rowscale_acc) x *= (m < p.M ? p.rowscale_acc[(long long)bz * p.M + m] : 0.f);
                x += e_bias;
                if (e_hn) {   // per-head RMSNorm: 8 consecutive lanes own one 32-wide head
                    float ss = x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
                    ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4);
                    const float rs = rsqrtf(ss * (1.0f / 32.0f) + p.hn_eps);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = x[e] * rs * e_hw[e];
                }
                if (glu) {
                    f32x4 b2 = *reinterpret_cast<const f32x4*>(Cs + row * LDC + e_pc + 32);
                    b2 += e_bias2;
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = (glu == 1) ? pd_silu(x[e]) * b2[e] : x[e] * pd_sigmoid(b2[e]);
                } else {
                    pd_act4(x, p.act);
                }
                v[u] = x;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!ok[u]) continue;
                const int m = mrow[u], n = ncol[u];
                f32x4 x = v[u];
                if (p.rowscale) x *= p.rowscale[m];
                if (p.maskadd && p.maskadd[m] == 0.f) x += p.maskval;
                x *= mulv[u];
                x *= p.out_scale;
                x += resv[u];
                float* yp = Y + (long long)m * p.ldy + n;
                if (full[u]) *reinterpret_cast<f32x4*>(yp) = x;
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (n + e < N_out) yp[e] = x[e];
                }
            }
        }
    } else if (p.out_mode == PD_OUT_TRANSPOSED) {
        // Y[n][m]: 4 consecutive m per thread.  Supports bias, glu/act, rowscale, out_scale.
        const int mch = SLAB / 4;
#pragma unroll 1
        for (int idx = tid; idx < BN_out * mch; idx += NT) {
            const int nl = idx / mch, mc = idx - nl * mch;
            const int n = bn_out0 + nl, m0 = bm0 + slab0 + mc * 4;
            if (n >= N_out || m0 >= p.M) continue;
            const int pc = glu ? (nl >> 5) * 64 + (nl & 31) : nl;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = Cs[(mc * 4 + e) * LDC + pc];
                if (biasp && bn0 + pc < p.N) a += biasp[bn0 + pc];
                if (glu) {
                    float b2 = Cs[(mc * 4 + e) * LDC + pc + 32];
                    if (biasp && bn0 + pc + 32 < p.N) b2 += biasp[bn0 + pc + 32];
                    a = (glu == 1) ? pd_silu(a) * b2 : a * pd_sigmoid(b2);
                } else a = pd_act(a, p.act);
                if (p.rowscale && m0 + e < p.M) a *= p.rowscale[m0 + e];
                v[e] = a * p.out_scale;
            }
            float* yp = Y + (long long)n * p.ldy + m0;
            if (p.vecY && m0 + 3 < p.M) *reinterpret_cast<f32x4*>(yp) = v;
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (m0 + e < p.M) yp[e] = v[e];
            }
        }
    } else {
        // PD_OUT_OPM / PD_OUT_BIASFRAG: scalar scatter (bias, maskadd, out_scale)
#pragma unroll 1
        for (int idx = tid; idx < SLAB * BN; idx += NT) {
            const int row = idx / BN, col = idx - row * BN;
            const int m = bm0 + slab0 + row, n = bn0 + col;
            if (m >= p.M || n >= p.N) continue;
            float v = Cs[row * LDC + col];
            if (biasp) v += biasp[n];
            if (p.maskadd && p.maskadd[m] == 0.f) v += p.maskval;
            v *= p.out_scale;
            if (p.out_mode == PD_OUT_OPM) {
                // rows (i,c), cols (j,d) -> [i][j][c][d]   (outer_product_mean.py:28)
                const int ii = m >> 5, c = m & 31, jj = n >> 5, d = n & 31;
                Y[(((long long)ii * p.T2 + jj) * 32 + c) * 32 + d] = v;
            } else {
                // attention-bias fragment layout (attention.hip); n = head, m = (i, j)
                int qi = m / p.T2, kj = m - qi * p.T2, nq = p.T1, nkk = p.T2;
                if (p.frag_transpose) { const int t = qi; qi = kj; kj = t; nq = p.T2; nkk = p.T1; }
                const int nqt = (nq + 31) >> 5, nkt = (nkk + 31) >> 5;
                const long long base = (((long long)n * nqt + (qi >> 5)) * nkt + (kj >> 5)) * 1024;
                const int k5 = kj & 31;
                Y[base + (k5 >> 3) * 256 + ((qi & 31) + 32 * ((k5 >> 2) & 1)) * 4 + (k5 & 3)] = v;
            }
        }
    }
  }   // slab pass
    if (dbg) dbg[60 * 5 + 3] = __builtin_amdgcn_s_memtime();
}

// op 0: launch, op 1: raise the dynamic-LDS limit of the instantiation (pd_init)
template <int BM, int BN, int WM, int WN, bool AKM, bool WKM, bool VEC, int PRO>
int run(int op, const pd_gemm_args* p, hipStream_t s) {
    constexpr int lds = Cfg<BM, BN, AKM, WKM>::LDS_BYTES;
    auto k = gemm_kernel<BM, BN, WM, WN, AKM, WKM, VEC, PRO>;
    if (op == 1)
        return hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess
                   ? PD_OK : PD_ERR_LAUNCH;
    dim3 grid((p->M + BM - 1) / BM, (p->N + BN - 1) / BN, p->batch);
    hipLaunchKernelGGL(k, grid, dim3(NT), lds, s, *p);
    return pd_check_launch();
}

// tile / layout / prologue dispatch shared by pd_gemm (op 0) and pd_init (op 1, every variant)
int dispatch(int op, int cfg, bool akm, bool wkm, bool vec, int pro, const pd_gemm_args* p, hipStream_t s) {
#define PD_CASE(C, BM, BN, WM, WN, A, W, V, P) \
    if (cfg == C && akm == A && wkm == W && vec == V && pro == P) return run<BM, BN, WM, WN, A, W, V, P>(op, p, s);
    PD_CASE(0, 128, 128, 2, 2, false, false, true, 0)
    PD_CASE(0, 128, 128, 2, 2, false, false, true, 1)
    PD_CASE(0, 128, 128, 2, 2, false, false, true, 2)
    PD_CASE(0, 128, 128, 2, 2, true, false, true, 0)
    PD_CASE(0, 128, 128, 2, 2, true, false, true, 1)
    PD_CASE(0, 128, 128, 2, 2, true, true, true, 0)
    PD_CASE(1, 128, 64, 2, 2, false, false, true, 0)
    PD_CASE(1, 128, 64, 2, 2, false, false, true, 1)
    PD_CASE(2, 128, 32, 4, 1, false, false, true, 0)
    PD_CASE(2, 128, 32, 4, 1, false, false, true, 1)
    PD_CASE(3, 64, 64, 2, 2, false, false, true, 0)
    PD_CASE(3, 64, 64, 2, 2, false, false, true, 1)
    PD_CASE(3, 64, 64, 2, 2, false, false, true, 2)
    PD_CASE(3, 64, 64, 2, 2, false, false, false, 0)
    PD_CASE(3, 64, 64, 2, 2, true, true, true, 0)        // k-major x k-major on small tiles: the column-variant triangle einsum
#undef PD_CASE
    return PD_ERR_UNSUPPORTED;
}

inline bool aligned16(const void* ptr) { return ((uintptr_t)ptr & 15) == 0; }

}  // namespace

// gemm_stream.hip: persistent variant with a direct-from-fragment epilogue for large full-tile row-major problems
extern "C" int pd_gemm_stream_try(const pd_gemm_args* args, int pro, int tile, void* stream, int init_only);
// gemm_split.hip: the same persistent structure on the bf16 matrix pipe (3 x bf16 split operands, needs args->W3)
extern "C" int pd_gemm_split_try(const pd_gemm_args* args, int pro, int tile, void* stream, int init_only);

// gemm_f16.hip: the same structure on the fp16 matrix pipe (2 x fp16 split operands with power-of-two scales; needs args->W2,
// w_inv and the magnitude bound a_amax)
extern "C" int pd_gemm_f16_try(const pd_gemm_args* args, int pro, int tile, void* stream, int init_only);

// the persistent kernel family a launch goes to: two-part fp16 operands when the caller supplied scaled weights AND a bound of
// |A|, three-part bf16 operands when it supplied pre-split weights, else fp32.  *split: 0 fp32, 1 bf16 x 6, 2 f16 x 3
static int persistent_try(const pd_gemm_args* args, int pro, int tile, void* stream, int init_only, int* split = nullptr) {
    if (args && args->W2 && args->w_inv && args->a_amax && !(args->A3 && !args->A2)) {
        const int r = pd_gemm_f16_try(args, pro, tile, stream, init_only);
        if (r != PD_ERR_UNSUPPORTED) { if (split) *split = 2; return r; }
    }
    if (args && args->A2 && !args->A3) return PD_ERR_UNSUPPORTED;       // a pre-split fp16 A is readable by the fp16 kernel only
    if (args && args->W3) {
        const int r = pd_gemm_split_try(args, pro, tile, stream, init_only);
        if (r != PD_ERR_UNSUPPORTED) { if (split) *split = 1; return r; }
    }
    if (split) *split = 0;
    return pd_gemm_stream_try(args, pro, tile, stream, init_only);
}

extern "C" int pd_attention_init(void);

PD_EXPORT int pd_init(void) {
    int rc = pd_gemm_stream_try(nullptr, 0, 0, nullptr, 1);
    { const int r = pd_attention_init(); if (r != PD_OK) rc = r; }
    { const int r = pd_gemm_split_try(nullptr, 0, 0, nullptr, 1); if (r != PD_OK) rc = r; }
    { const int r = pd_gemm_f16_try(nullptr, 0, 0, nullptr, 1); if (r != PD_OK) rc = r; }
    { const int r = pd_transition_f16(nullptr, nullptr); if (r != PD_OK) rc = r; }
    { const int r = pd_tri_tail(nullptr, nullptr); if (r != PD_OK) rc = r; }
    for (int cfg = 0; cfg < 4; ++cfg)
        for (int lay = 0; lay < 3; ++lay)
            for (int vec = 0; vec < 2; ++vec)
                for (int pro = 0; pro < 3; ++pro) {
                    const int r = dispatch(1, cfg, lay >= 1, lay == 2, vec != 0, pro, nullptr, nullptr);
                    if (r != PD_OK && r != PD_ERR_UNSUPPORTED) rc = r;
                }
    return rc;
}

// shared argument normalisation + variant selection; returns <0 on error, else variant id
// id = cfg*100 + layout*10 + pro  (cfg 0:128x128 1:128x64 2:128x32 3:64x64; layout 0:NN 1:A k-major 2:both; +1000 scalar loads)
static int select_variant(pd_gemm_args& p, int& cfg, bool& akm, bool& wkm, bool& vec, int& pro) {
    if (!p.A || !p.W || !p.Y) return PD_ERR_ARG;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return PD_ERR_ARG;
    if (p.batch <= 0) p.batch = 1;
    if (p.out_scale == 0.f) p.out_scale = 1.f;
    if (p.glu && (p.N % 64 != 0)) return PD_ERR_ARG;
    if (p.hn_w && (p.hn_split % 32 != 0 || p.hn_cols % 32 != 0 || p.glu)) return PD_ERR_ARG;
    akm = p.a_kmajor != 0; wkm = p.w_kmajor != 0;
    // 16-byte vector loads need aligned row starts; tails are masked per element, so a row may
    // end anywhere inside its (ld-padded) last chunk.
    p.vecA = aligned16(p.A) && (p.lda % 4 == 0) && (p.sA % 4 == 0);
    p.vecW = aligned16(p.W) && (p.ldw % 4 == 0) && (p.sW % 4 == 0);
    p.vecY = aligned16(p.Y) && (p.ldy % 4 == 0) && (p.sY % 4 == 0) &&
             (!p.res || (aligned16(p.res) && p.ldres % 4 == 0 && p.sRes % 4 == 0)) &&
             (!p.mul || (aligned16(p.mul) && (p.mul_rows_per_group > 0 ? p.mul_gstride % 4 == 0 : p.ldmul % 4 == 0)));
    vec = p.vecA && p.vecW;
    pro = 0;
    if (p.stats || p.stats_inline) {
        if (!p.pro_w || !p.pro_b) return PD_ERR_ARG;      // pass ones / zeros explicitly
        pro = p.pro_rows_per_group > 0 ? 2 : 1;
        if (!akm && (!aligned16(p.pro_w) || !aligned16(p.pro_b) || p.pro_gstride % 4 != 0)) return PD_ERR_UNSUPPORTED;
    }
    if ((akm || wkm) && !vec) return PD_ERR_UNSUPPORTED;
    if (wkm && !akm) return PD_ERR_UNSUPPORTED;
    if (!vec && p.glu) return PD_ERR_UNSUPPORTED;
    const long long blocks128 = (long long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.batch;
    if (akm && wkm && !p.glu && pro == 0 && blocks128 < 192) cfg = 3;     // 32 batches of 256^3: 128 blocks of 128x128 leave half the chip idle
    else if (akm || wkm || p.glu) cfg = 0;
    else if (!vec) cfg = 3;
    else if (p.N <= 32) cfg = 2;
    else if (blocks128 < 192) cfg = 3;
    else if (p.N <= 64) cfg = 1;
    else cfg = 0;
    if (pro == 2 && cfg != 0 && cfg != 3) cfg = 0;
#ifdef PD_LAB
    if (const char* f = getenv("PD_GEMM_CFG")) {      // tuning override: force a tile configuration where legal
        const int c = atoi(f);
        if (c >= 0 && c <= 3 && !akm && !wkm && !p.glu && vec && pro != 2) cfg = c;
    }
#endif
    return cfg * 100 + (akm ? (wkm ? 2 : 1) : 0) * 10 + pro + (vec ? 0 : 1000);
}

// Ragged M for the streaming kernel: head = the whole row blocks of the tile size, tail = the remaining rows with every row-indexed
// operand advanced.  Only when row groups (AdaLN prologue / gate tables) do not subdivide the rows.
static bool split_rows(const pd_gemm_args& p, int pro, int tile, pd_gemm_args& head, pd_gemm_args& tail) {
    const int tm = tile == 64 ? 64 : 128;                   // row blocks of the tile code
    const int M0 = p.M / tm * tm;
    if (M0 == p.M || M0 == 0 || p.batch != 1 || p.a_kmajor || p.out_mode != PD_OUT_ROWMAJOR) return false;
    if (pro == 2 && p.pro_rows_per_group < p.M) return false;
    if (p.mul && p.mul_rows_per_group > 0 && p.mul_rows_per_group < p.M) return false;
    if (p.res_row_mod > 0 || p.rowscale_acc || p.rowscale || p.maskadd) return false;
    head = p; head.M = M0;
    if (head.mul && head.mul_rows_per_group > 0) head.mul_rows_per_group = M0;     // one group: keep it a multiple of 64
    tail = p; tail.M = p.M - M0;
    tail.A += (long long)M0 * p.lda;
    tail.Y += (long long)M0 * p.ldy;
    if (tail.res) tail.res += (long long)M0 * p.ldres;
    if (tail.mul && tail.mul_rows_per_group <= 0) tail.mul += (long long)M0 * p.ldmul;
    if (tail.stats) tail.stats += 2ll * M0;
    return true;
}

// block tile of gemm_stream.hip that corresponds to a tile configuration of this file (0 = none)
// (128 x 128 for cfg 0, 64 x 64 for cfg 3; GLU problems are always cfg 0 here and get the 128 x 64 tile, code 12864, when
// they are too small to fill the chip with 128 x 128 tiles)
static int stream_tile(int cfg, const pd_gemm_args& p) {
    if (cfg == 3) return 64;
    if (cfg != 0) return 0;
    const long long blocks128 = (long long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    return (p.glu && blocks128 < 384) ? 12864 : 128;
}

static bool use_stream() {
#ifdef PD_LAB
    static const int on = [] { const char* e = getenv("PD_GEMM_STREAM"); return e ? atoi(e) : 1; }();
    return on != 0 && !g_gemm_trace_on;          // the phase trace lives in the general kernel
#else
    return true;
#endif
}

#ifdef PD_LAB
extern "C" __attribute__((visibility("default"))) int pd_lab_set_gemm_trace(void* buf) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(buf);
    g_gemm_trace_on = buf != nullptr;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_trace), &q, sizeof(q)) == hipSuccess ? PD_OK : PD_ERR_LAUNCH;
}
#endif

// variant id as documented above; + 5000 + 10000 * EPI + 100000 * tile (0: 128x128, 1: 64x64 [64x128 for the fp16 kernel], 2: 128x64, 3 / 4: the fp16 ROWS kernel on 128- / 64-row tiles) when the launch
// goes to gemm_stream_kernel<pro, EPI, Tile<...>>, + 1000000 more when it goes to gemm_split_kernel<pro, EPI, STile<...>>
PD_EXPORT int pd_gemm_variant(const pd_gemm_args* args) {
    if (!args) return PD_ERR_ARG;
    pd_gemm_args p = *args;
    int cfg, pro; bool akm, wkm, vec;
    const int v = select_variant(p, cfg, akm, wkm, vec, pro);
    if (v >= 0 && !p.stats && p.stats_inline) {       // inline row statistics: the fp16-format rows kernel or the fp32 streaming kernel on the whole problem, or nothing
        if (p.W2 && p.w_inv && p.a_amax && !akm && !wkm && vec) {
            const int q16 = pd_gemm_f16_try(&p, pro, 128, nullptr, 2);
            if (q16 >= 0) return v + 5000 + 10000 * (q16 & 0xff) + 100000 * (q16 >> 8) + 2000000;
        }
        const int tile = use_stream() ? stream_tile(cfg, p) : 0;
        const int q = (tile && !p.A3 && !p.A2 && !p.Y2) ? pd_gemm_stream_try(&p, pro, tile, nullptr, 2) : PD_ERR_UNSUPPORTED;
        return q >= 0 ? v + 5000 + 10000 * (q & 0xff) + 100000 * (tile == 128 ? 0 : tile == 64 ? 1 : 2) : PD_ERR_UNSUPPORTED;
    }
    if (v >= 0 && (p.A3 || p.A2 || p.Y2)) {  // pre-split A / Y2: exactly pd_gemm's rule - a split-operand kernel on the whole problem, or nothing
        int split = 0;
        const int q = persistent_try(&p, pro, 128, nullptr, 2, &split);
        if (p.Y2 && split != 2) return v;            // only the fp16-format kernel writes the pre-split k | v
        return q >= 0 && split ? v + 5000 + 10000 * (q & 0xff) + 100000 * (q >> 8) + 1000000 * split : v;
    }
    if (v >= 0 && use_stream() && stream_tile(cfg, p)) {
        pd_gemm_args head, tail;
        const int tile = stream_tile(cfg, p);
        int split = 0;
        const int q = persistent_try(split_rows(p, pro, tile, head, tail) ? &head : &p, pro, tile, nullptr, 2, &split);
        // (the fp16 kernel picks its own tile: bit 8 of its answer = 64 x 128)
        if (q >= 0) return v + 5000 + 10000 * (q & 0xff) + 100000 * (split == 2 ? q >> 8 : tile == 128 ? 0 : tile == 64 ? 1 : 2) + 1000000 * split;
    }
    return v;
}

PD_EXPORT int pd_gemm(const pd_gemm_args* args, void* stream) {
    if (!args) return PD_ERR_ARG;
    pd_gemm_args p = *args;
    int cfg, pro; bool akm, wkm, vec;
    const int v = select_variant(p, cfg, akm, wkm, vec, pro);
    if (v < 0) return v;
    if (!p.stats && p.stats_inline) {                 // inline row statistics: the fp16-format rows kernel (K = 128) or the fp32 streaming kernel (full tiles)
        if (p.W2 && p.w_inv && p.a_amax && !akm && !wkm && vec) {
            const int r16 = pd_gemm_f16_try(&p, pro, 128, stream, 0);
            if (r16 != PD_ERR_UNSUPPORTED) return r16;
        }
        const int tile = use_stream() ? stream_tile(cfg, p) : 0;
        if (!tile || p.A3 || p.A2 || p.Y2) return PD_ERR_UNSUPPORTED;
        return pd_gemm_stream_try(&p, pro, tile, stream, 0);
    }
    if (p.A3 || p.A2 || p.Y2) {  // pre-split A: only a split-operand kernel can read it; anything else is an error, never the raw A
        int split = 0;
        const int q = persistent_try(&p, pro, 128, nullptr, 2, &split);
        if (q < 0 || !split || (p.Y2 && split != 2)) return PD_ERR_UNSUPPORTED;
        return split == 2 ? pd_gemm_f16_try(&p, pro, 128, stream, 0) : pd_gemm_split_try(&p, pro, 128, stream, 0);
    }
    if (use_stream() && stream_tile(cfg, p)) {
        pd_gemm_args head, tail;
        const int tile = stream_tile(cfg, p);
        if (!split_rows(p, pro, tile, head, tail)) {
            const int r = persistent_try(&p, pro, tile, stream, 0);
            if (r != PD_ERR_UNSUPPORTED) return r;
        } else if (persistent_try(&head, pro, tile, nullptr, 2) >= 0) {
            // whole row blocks on the streaming kernel, the ragged remainder (< one tile of rows) on the general one
            const int r = persistent_try(&head, pro, tile, stream, 0);
            if (r != PD_OK) return r;
            int tcfg, tpro; bool takm, twkm, tvec;
            const int tv = select_variant(tail, tcfg, takm, twkm, tvec, tpro);
            if (tv < 0) return tv;
            return dispatch(0, tcfg, takm, twkm, tvec, tpro, &tail, (hipStream_t)stream);
        }
    }
    return dispatch(0, cfg, akm, wkm, vec, pro, &p, (hipStream_t)stream);
}
