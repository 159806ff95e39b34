// linear_downscale + SiLU + token pooling of the denoiser in ONE launch (reference layers/transformers.py:205-212):
//     bs[b, t, :] = ( sum_{atoms l of token t} silu(W ba[b, l, :] + bias) ) / (n_t + 1e-3) + s[t, :]
// Round 5.  The two-launch form writes u = silu(W ba + b) - [B A, 512] fp32, 268 MB at the benchmark shape - and
// reads it back for the segment mean: 135 + 46 us per step, store-bound.  Here a block owns the atoms of `tpb` consecutive tokens of
// one sample (at most 64 rows: the host picks tpb = 64 / max atoms per token) and keeps them in LDS in the two-part fp16 format of
// gemm_f16.hip.  The residual stream has no static magnitude bound - but the block stages its WHOLE operand tile, so it measures the
// tile's own maximum while loading and derives the power-of-two operand scale from that: the tightest scale there is, exact, data-
// dependent but order-free (a maximum), hence bit-reproducible.  Every wave then walks 32-column blocks of W (two fp16 parts with
// per-row scales, packing.split2_f16): 48 MFMAs of projection, bias + SiLU on the accumulator fragments, and then the POOLING ITSELF AS A
// MATRIX PRODUCT on the same pipe: pooled[t, j] = sum_rows P[t, row] u[row, j] with P the 0 / 1 membership matrix of the block's
// tokens (exact in bf16) and u split into three bf16 parts (exact), 12 MFMAs, accumulated in fp32 in the matrix pipe's fixed order -
// bit-reproducible, no atomics, no cross-lane shuffles.  The k index of that product is a free choice; it is chosen so that a lane's
// eight k-slots ARE the accumulator registers it already holds (rows (s & 3) + 8 (s >> 2) + 4 hh of a 16-row group), so u never
// moves between lanes.  Only bs (33 MB) is written.
#include "common.h"
#include "physdock_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int CIN = 128, ROWS = 64, PITCH = 136;          // bf16 elements per LDS row (272 bytes = 17 x 16: conflict-free ds_read_b128)
constexpr int PART = ROWS * PITCH;
constexpr int NKS = CIN / 16;

#ifndef PD_POOL_BPC
#define PD_POOL_BPC 3          // blocks per CU the register budget is cut for: 3 = 168 registers, no spill (round 6, late: 4 = 128 registers, 38 spill
                              // instructions, 122 us against 109 us per launch at 64 samples, bit-identical; profiles/r06_ab_pool_register_budget.txt)
#endif
__global__ __launch_bounds__(256, PD_POOL_BPC) void downscale_pool_kernel(const float* __restrict__ ba, const _Float16* __restrict__ W2,
                                                                const float* __restrict__ w_inv, const float* __restrict__ bias,
                                                                const int* __restrict__ tok_start,
                                                                const float* __restrict__ add, float* __restrict__ out,
                                                                int A, int T, int N, int tpb) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[2 * PART];
    __shared__ float wmax[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int b = blockIdx.y, t0 = blockIdx.x * tpb;
    const int ntok = T - t0 < tpb ? T - t0 : tpb;
    const int a0 = tok_start[t0];
    int n = tok_start[t0 + ntok] - a0;
    // The launcher's contract is n <= 64 (tpb = 64 / max atoms per token).  The token table lives in device memory, so the C entry point
    // cannot check it: a block that finds more atoms writes NaN for its tokens - the caller's finite check fails loudly instead of a
    // silently wrong mean.
    const bool over = n > ROWS;
    n = over ? ROWS : n;

    // ---- stage the block's atom rows: the tile's own maximum -> power-of-two scale -> two fp16 parts, once
    float a_s;
    {
        const int row = tid >> 2, c0 = (tid & 3) * 32;
        const float* src = ba + ((long long)b * A + a0 + row) * CIN + c0;
        f32x4 v[8];
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < n) v[i] = *reinterpret_cast<const f32x4*>(src + 4 * i);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v[i][0]), fabsf(v[i][1]))), fmaxf(fabsf(v[i][2]), fabsf(v[i][3])));
        }
        m = wave_max(m);
        if (lane == 0) wmax[wave] = m;
        __syncthreads();
        a_s = pd_pow2_scale(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const pd_parts2 p0 = pd_split2h(v[i][0] * a_s, v[i][1] * a_s), p1 = pd_split2h(v[i][2] * a_s, v[i][3] * a_s);
            unsigned short* d = lds + row * PITCH + c0 + 4 * i;
            *reinterpret_cast<u32x2*>(d) = u32x2{p0.h, p1.h};
            *reinterpret_cast<u32x2*>(d + PART) = u32x2{p0.l, p1.l};
        }
    }
    const float inv_a_s = 1.0f / a_s;
    // ---- membership fragments of the pooling product (A operand: lane = token slot l31, k-slot kappa = 8 hh + s of k-step q <-> atom
    //      row 16 q + 8 (s >> 2) + 4 hh + (s & 3)) and the tokens' 1 / (n_t + 1e-3) in accumulator-register order
    int ra = 0, rb = 0;
    if (l31 < ntok) { ra = tok_start[t0 + l31] - a0; rb = tok_start[t0 + l31 + 1] - a0; }
    bf16x8 pm[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        u32x4 f;
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            unsigned w = 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int s = 2 * s2 + e;
                const int row = 16 * q + 8 * (s >> 2) + 4 * hh + (s & 3);
                if (row >= ra && row < rb) w |= 0x3F80u << (16 * e);          // bf16 1.0
            }
            f[s2] = w;
        }
        pm[q] = __builtin_bit_cast(bf16x8, f);
    }
    float inv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int tt = pd_frag_row(r, hh);
        inv[r] = tt < ntok ? 1.0f / ((float)(tok_start[t0 + tt + 1] - tok_start[t0 + tt]) + 1e-3f) : 0.f;
    }
    __syncthreads();

    const int ncb = N >> 5;
    const unsigned short* abase = lds + l31 * PITCH + 8 * hh;
    const long long wpart = (long long)ncb * NKS * 512;                   // fp16 elements per part of the fragment-major weights
    for (int cb = wave; cb < ncb; cb += 4) {
        const _Float16* wb = W2 + ((long long)cb * NKS * 64 + lane) * 8;
        f16x8 wf[2][2];
        auto wload = [&](int buf, int ks) {
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) wf[buf][pt] = *reinterpret_cast<const f16x8*>(wb + pt * wpart + (long long)ks * 512);
        };
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        wload(0, 0);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks + 1 < NKS) wload((ks + 1) & 1, ks + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f16x8 a0 = *reinterpret_cast<const f16x8*>(abase + 32 * i * PITCH + 16 * ks);
                const f16x8 a1 = *reinterpret_cast<const f16x8*>(abase + PART + 32 * i * PITCH + 16 * ks);
                f32x16 c = acc[i];
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, wf[ks & 1][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, wf[ks & 1][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, wf[ks & 1][0], c, 0, 0, 0);
                acc[i] = c;
            }
        }
        const int col = cb * 32 + l31;
        const float bv = bias ? bias[col] : 0.f;
        const float cs = w_inv[col] * inv_a_s;                                // undo the weight row's and the tile's operand scales (exact)
        // u = silu(acc + bias) (rows beyond n are silu(bias): no token's membership row selects them), then pooled = P . u on the pipe
        f32x16 pz;
#pragma unroll
        for (int r = 0; r < 16; ++r) pz[r] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = q >> 1, e8 = 8 * (q & 1);
            u32x4 fh, fm, fl;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const pd_parts t = pd_split2(pd_silu_r(__builtin_fmaf(acc[i][e8 + 2 * s2], cs, bv)), pd_silu_r(__builtin_fmaf(acc[i][e8 + 2 * s2 + 1], cs, bv)));
                fh[s2] = t.h; fm[s2] = t.m; fl[s2] = t.l;
            }
            pz = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pm[q], __builtin_bit_cast(bf16x8, fl), pz, 0, 0, 0);
            pz = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pm[q], __builtin_bit_cast(bf16x8, fm), pz, 0, 0, 0);
            pz = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pm[q], __builtin_bit_cast(bf16x8, fh), pz, 0, 0, 0);
        }
        float* op = out + ((long long)b * T + t0) * N + col;
        const float* ad = add ? add + (long long)t0 * N + col : nullptr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int tt = pd_frag_row(r, hh);
            if (tt < ntok) op[(long long)tt * N] = over ? __builtin_nanf("") : pz[r] * inv[r] + (ad ? ad[(long long)tt * N] : 0.f);
        }
    }
}

}  // namespace

// ba [B][A][128] fp32 (row pitch 128), W2 / w_inv = packing.split2_f16 of the [N][128] weight (two fp16 parts, fragment-major, and the
// inverse row scales), bias [N] or NULL, tok_start [T + 1] (atoms of a token contiguous, tokens ascending), add [T][N] or NULL, out
// [B][T][N].  tpb tokens per block, 1 .. 32, and the caller guarantees that tpb consecutive tokens never hold more than 64 atoms (tpb =
// 64 / max atoms per token); a block that finds more writes NaN into its tokens' rows.  PD_ERR_UNSUPPORTED for other shapes: run pd_gemm
// (act = SiLU) + pd_segment_pool.
PD_EXPORT int pd_downscale_pool(const float* ba, const void* W2, const float* w_inv, const float* bias, const int* tok_start, const float* add,
                                float* out, int B, int A, int T, int Cin, int N, int tpb, void* stream) {
    if (!ba || !W2 || !w_inv || !tok_start || !out || B <= 0 || A <= 0 || T <= 0) return PD_ERR_ARG;
    if (Cin != CIN || N <= 0 || N % 32 != 0 || tpb < 1 || tpb > 32) return PD_ERR_UNSUPPORTED;
    if ((((uintptr_t)ba | (uintptr_t)W2) & 15) != 0) return PD_ERR_UNSUPPORTED;
    dim3 grid((unsigned)((T + tpb - 1) / tpb), (unsigned)B);
    hipLaunchKernelGGL(downscale_pool_kernel, grid, dim3(256), 0, (hipStream_t)stream, ba, reinterpret_cast<const _Float16*>(W2), w_inv, bias,
                       tok_start, add, out, A, T, N, tpb);
    return pd_check_launch();
}
