// Pair-bias projection as ONE streaming pass over the pair tensor: row statistics + norm + skinny projection (C -> H <= 24
// heads) + mask + fragment-layout store.
//
// Replaces, for the attention biases of the trunk (reference primitives/attentions.py:38-41,82-85,200-203: bias =
// linear_z(norm_z(z)), H = 4 / 8 / 16 over z [T,T,128]; H = 4 over ap [A,A,16]) and the hoisted DiT atom bias
// (attentions.py:246,254 with LayerNorm, 6 blocks x 4 heads over ap), the pair  pd_rowstats + pd_gemm(N = H, PD_OUT_BIASFRAG):
// the general GEMM reaches ~1 TB/s on these N <= 24 problems (33 us for a 33.5 MB z, 683 us for a 268 MB ap) and the
// statistics kernel reads the tensor a second time.  Here every row is read once, normalised in registers (two-pass
// variance like pd_rowstats), contracted with the norm-folded weights W'[h][k] = w[k] W[h][k] held in registers, reduced over
// the lanes of the row with DPP adds, and written straight into the attention kernel's bias fragment layout through a
// per-wave LDS transpose (16-byte stores of four consecutive keys).  The (mean, rstd) pairs are stored as a by-product:
// the q|k|v|g projection that follows consumes the same statistics.  HBM-bound: bytes = M C 4 read + M H 4 written.
#include <string.h>
#include "common.h"
#include "physdock_hip.h"

#ifndef PD_PB_UB
#define PD_PB_UB 8
#endif
#ifndef PD_PB_TR
#define PD_PB_TR 8
#endif
#ifndef PD_PB_TR4
#define PD_PB_TR4 32
#endif

namespace {

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {       // sum over the LPR lanes that share a row, result in all of them
    if constexpr (LPR == 32) return pd_half_sum32(v);
    else {                                                    // LPR == 4: the quad
        v += pd_dpp<0xB1>(v);
        v += pd_dpp<0x4E>(v);
        return v;
    }
}

// C = 4 LPR channels, H heads; 4 waves per block, every wave walks TR-row tiles.  TR = 16 for C = 128: the per-row work is a
// chain of dependent DPP reductions, so the kernel wants many short waves per SIMD (65 536 rows = 4 096 wave tiles), not 1 024
// long ones (21 -> see NOTES); C = 16 rows are cheap and stay at 64.
// Z2 (round 6, C = 128 / RMS only): the normalised rows x / rms(x), times the power-of-two operand scale z2_scale, are ALSO written in
// the two-part fp16 format and in the FRAGMENT-MAJOR order pd_tri_attention reads them in (csrc/tri_attn.hip):
//   z2[batch b][32-row tile][k-step s = 16 channels][part][lane' = (row & 31) + 32 hh][8 halves],  channels 16 s + 8 hh .. + 8,
// batch / row = the two pair indices (swapped for the column variant).  The lane that holds channels 4 sub .. + 4 of a row owns 8 bytes
// of one 16-byte fragment slot per part.
template <int LPR, int H, int TR, bool Z2 = false>
__global__ __launch_bounds__(256) void pair_bias_kernel(const float* __restrict__ x, const float* __restrict__ Wf,
                                                       const float* __restrict__ c2, float* __restrict__ stats,
                                                       const float* __restrict__ maskadd, float maskval, float out_scale,
                                                       float* __restrict__ frag, long long M, int T1, int T2, int transpose,
                                                       int mode, float eps, unsigned short* __restrict__ z2 = nullptr, float z2_scale = 0.f) {
    constexpr int C = 4 * LPR, RPI = 64 / LPR, NI = TR / RPI;
    static_assert(TR % RPI == 0 && TR % 4 == 0 && TR <= 64, "tile rows");
    __shared__ float tile[4][TR][H + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPR, grp = lane / LPR;
    f32x4 w[H];
    float cb[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        w[h] = *reinterpret_cast<const f32x4*>(Wf + h * C + 4 * sub);
        cb[h] = c2 ? c2[h] : 0.f;
    }
    const long long ntile = (M + TR - 1) / TR;
    const int nq = transpose ? T2 : T1, nk = transpose ? T1 : T2;
    const int nqt = (nq + 31) >> 5, nkt = (nk + 31) >> 5;
    for (long long t = (long long)blockIdx.x * 4 + wave; t < ntile; t += (long long)gridDim.x * 4) {
        const long long m0 = t * TR;
        constexpr int UB = NI < PD_PB_UB ? NI : PD_PB_UB;          // rows in flight per lane: UB independent 16-byte loads before any use
        for (int q0 = 0; q0 < NI; q0 += UB) {
            f32x4 vv[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int r = LPR == 32 ? (q0 + u) + (TR / 2) * grp : (q0 + u) * RPI + grp;
                const long long row = m0 + r;
                vv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (row < M) vv[u] = *reinterpret_cast<const f32x4*>(x + row * C + 4 * sub);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                // C = 128: half 0 takes the first TR/2 rows of the tile, half 1 the rest (two contiguous 512-byte reads per
                // instruction); C = 16: 16 consecutive rows per instruction
                const int r = LPR == 32 ? (q0 + u) + (TR / 2) * grp : (q0 + u) * RPI + grp;
                const long long row = m0 + r;
                const bool ok = row < M;
                f32x4 v = vv[u];
                float mean = 0.f, rstd;
                if (mode == 0) {
                    const float s2 = group_sum<LPR>(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
                    rstd = rsqrtf(s2 / (float)C + eps);
                } else {
                    mean = group_sum<LPR>(v[0] + v[1] + v[2] + v[3]) / (float)C;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] -= mean;
                    const float s2 = group_sum<LPR>(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
                    rstd = rsqrtf(s2 / (float)C + eps);
                }
                const float madd = (ok && maskadd && maskadd[row] == 0.f) ? maskval : 0.f;
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const float d = group_sum<LPR>(v[0] * w[h][0] + v[1] * w[h][1] + v[2] * w[h][2] + v[3] * w[h][3]);
                    if (sub == h % LPR) tile[wave][r][h] = ((d * rstd + cb[h]) + madd) * out_scale;   // spread the LDS writes over lanes
                }
                if (ok && stats && sub == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
                if constexpr (Z2) {
                    if (ok) {
                        const int i0 = (int)(row / T2), i1 = (int)(row - (long long)i0 * T2);
                        const int bz = transpose ? i1 : i0, rz = transpose ? i0 : i1;
                        const float f = rstd * z2_scale;
                        const pd_parts2 p0 = pd_split2h(v[0] * f, v[1] * f), p1 = pd_split2h(v[2] * f, v[3] * f);
                        const int ntile = (T2 + 31) >> 5;
                        // ((((b NT + tile) 8 + s) 2 + part) 64 + lane') 8 + 4 e   halves;  s = sub >> 2, hh = (sub >> 1) & 1, e = sub & 1
                        const long long o = (((((long long)bz * ntile + (rz >> 5)) * 8 + (sub >> 2)) * 2) * 64 + (rz & 31) + 32 * ((sub >> 1) & 1)) * 8 + 4 * (sub & 1);
                        typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
                        *reinterpret_cast<u32x2_*>(z2 + o) = u32x2_{p0.h, p1.h};
                        *reinterpret_cast<u32x2_*>(z2 + o + 512) = u32x2_{p0.l, p1.l};
                    }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);          // this wave's LDS writes have landed (the tile is private to the wave)
        asm volatile("" ::: "memory");
        if (!transpose) {
            // rows m = (i, j): query i, key j; four consecutive j share one 16-byte slot of the fragment layout
            for (int idx = lane; idx < H * (TR / 4); idx += 64) {
                const int h = idx / (TR / 4), quad = idx % (TR / 4);
                const long long m = m0 + 4 * quad;
                if (m >= M) continue;
                const int qi = (int)(m / T2), kj = (int)(m - (long long)qi * T2);
                const f32x4 o = {tile[wave][4 * quad][h], tile[wave][4 * quad + 1][h], tile[wave][4 * quad + 2][h],
                                 tile[wave][4 * quad + 3][h]};
                const int k5 = kj & 31;
                const long long a = (((long long)h * nqt + (qi >> 5)) * nkt + (kj >> 5)) * 1024 + (k5 >> 3) * 256 +
                                    ((qi & 31) + 32 * ((k5 >> 2) & 1)) * 4;
                *reinterpret_cast<f32x4*>(frag + a) = o;
            }
        } else {
            for (int idx = lane; idx < H * TR; idx += 64) {
                const int h = idx / TR, r = idx % TR;
                const long long m = m0 + r;
                if (m >= M) continue;
                const int i = (int)(m / T2), j = (int)(m - (long long)i * T2);
                const int qi = j, kj = i, k5 = kj & 31;
                frag[(((long long)h * nqt + (qi >> 5)) * nkt + (kj >> 5)) * 1024 + (k5 >> 3) * 256 +
                     ((qi & 31) + 32 * ((k5 >> 2) & 1)) * 4 + (k5 & 3)] = tile[wave][r][h];
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xc07f);          // reads done before the next tile overwrites
    }
}

template <int LPR, int H, int TR>
int launch(const float* x, const float* Wf, const float* c2, float* stats, const float* maskadd, float maskval, float out_scale,
           float* frag, long long M, int T1, int T2, int transpose, int mode, float eps, hipStream_t s, unsigned short* z2 = nullptr,
           float z2_scale = 0.f) {
    const long long ntile = (M + TR - 1) / TR;
    long long blocks = (ntile + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if constexpr (LPR == 32 && H == 4) {
        if (z2) {
            hipLaunchKernelGGL((pair_bias_kernel<LPR, H, TR, true>), dim3((unsigned)blocks), dim3(256), 0, s, x, Wf, c2, stats, maskadd, maskval,
                               out_scale, frag, M, T1, T2, transpose, mode, eps, z2, z2_scale);
            return pd_check_launch();
        }
    }
    hipLaunchKernelGGL((pair_bias_kernel<LPR, H, TR>), dim3((unsigned)blocks), dim3(256), 0, s, x, Wf, c2, stats, maskadd, maskval,
                       out_scale, frag, M, T1, T2, transpose, mode, eps, nullptr, 0.f);
    return pd_check_launch();
}

}  // namespace

PD_EXPORT int pd_pair_bias(const float* x, const float* Wf, const float* c2, float* stats_out, const float* maskadd,
                           float maskval, float out_scale, float* frag, int T1, int T2, int C, int H, int frag_transpose,
                           int mode, float eps, void* stream) {
    if (!x || !Wf || !frag || T1 <= 0 || T2 <= 0) return PD_ERR_ARG;
    if (T2 % 4 != 0) return PD_ERR_UNSUPPORTED;              // a 16-byte fragment slot = four consecutive keys of one query
    if (((uintptr_t)x | (uintptr_t)Wf | (uintptr_t)frag) & 15) return PD_ERR_UNSUPPORTED;
    const long long M = (long long)T1 * T2;
    if (out_scale == 0.f) out_scale = 1.f;
    hipStream_t s = (hipStream_t)stream;
#define PD_PB(LPR, HH) if (C == 4 * LPR && H == HH) \
        return launch<LPR, HH, (LPR == 32 ? PD_PB_TR : PD_PB_TR4)>(x, Wf, c2, stats_out, maskadd, maskval, out_scale, frag, M, T1, T2, frag_transpose, mode, eps, s);
    PD_PB(32, 4) PD_PB(32, 8) PD_PB(32, 16) PD_PB(4, 4) PD_PB(4, 24)
#undef PD_PB
    return PD_ERR_UNSUPPORTED;
}

// pd_pair_bias for the TriangleAttention (C = 128, H = 4, RMS, T1 == T2) that ALSO writes the normalised rows x / rms(x), scaled and split
// into the two-part fp16 format, in pd_tri_attention's fragment-major order (see pair_bias_kernel): z2 [T][ceil(T/32)][8][2][64][8]
// halves; rows beyond T of the last tile are never written (the caller zeroes the buffer once).  zn_amax: the bound of |x / rms(x)|
// (sqrt(C)) the operand scale derives from - the same float pd_tri_attention is given.
PD_EXPORT int pd_pair_bias_split(const float* x, const float* Wf, const float* c2, float* stats_out, const float* maskadd,
                                 float maskval, float out_scale, float* frag, int T, int frag_transpose, float eps, void* z2,
                                 float zn_amax, void* stream) {
    if (!x || !Wf || !frag || !z2 || T <= 0 || !(zn_amax > 0.f)) return PD_ERR_ARG;
    if (T % 4 != 0) return PD_ERR_UNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)Wf | (uintptr_t)frag | (uintptr_t)z2) & 15) return PD_ERR_UNSUPPORTED;
    if (out_scale == 0.f) out_scale = 1.f;
    int e;                                                   // pd_pow2_scale on the host (csrc/common.h): 2^(14 - floor(log2 amax))
    {
        unsigned u;
        memcpy(&u, &zn_amax, 4);
        e = (int)((u >> 23) & 0xff);
        e = e < 87 ? 87 : (e > 200 ? 200 : e);
    }
    const unsigned sb = (unsigned)(268 - e) << 23;
    float z2_scale;
    memcpy(&z2_scale, &sb, 4);
    return launch<32, 4, PD_PB_TR>(x, Wf, c2, stats_out, maskadd, maskval, out_scale, frag, (long long)T * T, T, T, frag_transpose, 0, eps,
                                   (hipStream_t)stream, reinterpret_cast<unsigned short*>(z2), z2_scale);
}
