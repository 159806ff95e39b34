// Row statistics and standalone row normalisation (HBM-bound streaming kernels).
//
// pd_rowstats feeds the GEMM prologue: RMSNorm (reference primitives/rms_norm.py:14-19),
// nn.LayerNorm (layer_norm.py:5) and the affine-free LayerNorm inside AdaLN-Zero
// (adaptive_layer_norm_zero.py:16,20) all reduce to a per-row (mean, rstd) pair that the
// GEMM applies while staging its A tile, so the normalised activations never touch HBM.
// pd_rownorm is the standalone form for the two places where a norm FOLLOWS a projection
// (outer_product_mean.py:30) or feeds a non-GEMM consumer.
#include "common.h"
#include "physdock_hip.h"

namespace {

constexpr int MAXV = 4;   // float4 per lane: C <= LPR*16

// LPR lanes cooperate on one row; rows are packed 64/LPR per wave, 4 waves per block.
// WRITE: 0 statistics only, 1 normalised fp32 rows, 2 normalised + modulated rows as THREE bf16 parts (error-free split,
// out3 = [3][M][C]: the pre-split A operand of csrc/gemm_split.hip, so that its staging is a plain copy)
template <int LPR, int WRITE>
__global__ __launch_bounds__(256) void rownorm_kernel(const float* __restrict__ x, float* __restrict__ stats,
                                                     float* __restrict__ y, const float* __restrict__ res,
                                                     const float* __restrict__ w, const float* __restrict__ b,
                                                     int M, int C, int ldx, int mode, float eps, int act,
                                                     int rows_per_group = 0, int gstride = 0, const float* __restrict__ amax = nullptr) {
    constexpr int RPB = 256 / LPR;
    const int sub = threadIdx.x % LPR;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / LPR;
    const bool ok = row < M;
    const int nchunk = C >> 2;                 // C % 4 == 0 checked by the launcher
    // split-writing modes: the gain / shift rows of this row's group, requested BEFORE the x loads so that all of them are in flight
    // together (as scalar loads after the reductions hipcc emitted eight dependent load round trips per
    // chunk: 25 us for 16384 x 512 where the bytes need ~12)
    f32x4 wv[MAXV], bv[MAXV];
    if constexpr (WRITE == 2 || WRITE == 3) {
        const long long goff = rows_per_group > 0 ? (row / rows_per_group) * (long long)gstride : 0;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = sub + i * LPR;
            wv[i] = f32x4{1.f, 1.f, 1.f, 1.f};
            bv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok && c < nchunk) {
                if (w) wv[i] = *reinterpret_cast<const f32x4*>(w + goff + c * 4);
                if (b) bv[i] = *reinterpret_cast<const f32x4*>(b + goff + c * 4);
            }
        }
    }
    f32x4 v[MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = sub + i * LPR;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok && c < nchunk) v[i] = *reinterpret_cast<const f32x4*>(x + row * ldx + c * 4);
        s1 += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        s2 += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    float mean = 0.f, rstd;
    if (mode == 0) {
        rstd = rsqrtf(s2 / (float)C + eps);
    } else {
        mean = s1 / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = sub + i * LPR;
            if (c < nchunk) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { float d = v[i][e] - mean; q += d * d; }
            }
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
        rstd = rsqrtf(q / (float)C + eps);
    }
    if (!ok) return;
    if constexpr (WRITE == 0) {
        if (sub == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
    } else if constexpr (WRITE == 3) {
        // the same rows times the power of two that brings the bound *amax below 2^14, as TWO fp16 parts (hi, lo): the
        // pre-split A operand of csrc/gemm_f16.hip (out2 = [2][M][C])
        const float a_s = pd_pow2_scale(*amax);
        unsigned short* out = reinterpret_cast<unsigned short*>(y);
        const long long part = (long long)M * C;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = sub + i * LPR;
            if (c >= nchunk) continue;
            float t[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = ((v[i][e] - mean) * rstd * wv[i][e] + bv[i][e]) * a_s;
            const pd_parts2 p0 = pd_split2h(t[0], t[1]), p1 = pd_split2h(t[2], t[3]);
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            unsigned short* o = out + row * (long long)C + c * 4;
            *reinterpret_cast<u32x2*>(o) = u32x2{p0.h, p1.h};
            *reinterpret_cast<u32x2*>(o + part) = u32x2{p0.l, p1.l};
        }
    } else if constexpr (WRITE == 2) {
        // a' = (x - mean) rstd w[g][k] + b[g][k], g = row / rows_per_group: the GEMM prologue's expression, then the 3-way split
        // (8-byte stores per part; pairing chunks into 16-byte stores measured slower: 27 -> 43 us at [16384, 512])
        unsigned short* out = reinterpret_cast<unsigned short*>(y);          // bf16 [3][M][C]
        const long long part = (long long)M * C;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = sub + i * LPR;
            if (c >= nchunk) continue;
            float t[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = (v[i][e] - mean) * rstd * wv[i][e] + bv[i][e];
            const pd_parts p0 = pd_split2(t[0], t[1]), p1 = pd_split2(t[2], t[3]);
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            unsigned short* o = out + row * (long long)C + c * 4;
            *reinterpret_cast<u32x2*>(o) = u32x2{p0.h, p1.h};
            *reinterpret_cast<u32x2*>(o + part) = u32x2{p0.m, p1.m};
            *reinterpret_cast<u32x2*>(o + 2 * part) = u32x2{p0.l, p1.l};
        }
    } else {
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = sub + i * LPR;
            if (c >= nchunk) continue;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = (v[i][e] - mean) * rstd;
                if (w) t *= w[c * 4 + e];
                if (b) t += b[c * 4 + e];
                o[e] = pd_act(t, act);
            }
            if (res) {
                f32x4 r = *reinterpret_cast<const f32x4*>(res + row * (long long)C + c * 4);
                o += r;
            }
            *reinterpret_cast<f32x4*>(y + row * (long long)C + c * 4) = o;
        }
    }
}

// stats over the slow axis of a [C][M] array (triangle-update output, channel-major)
__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ x, float* __restrict__ stats,
                                                      int M, int C, long long ldx, int mode, float eps) {
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float s1 = 0.f, s2 = 0.f;
    for (int c = 0; c < C; ++c) { float v = x[c * ldx + m]; s1 += v; s2 += v * v; }
    float mean = 0.f, rstd;
    if (mode == 0) rstd = rsqrtf(s2 / (float)C + eps);
    else {
        mean = s1 / (float)C;
        float q = 0.f;
        for (int c = 0; c < C; ++c) { float d = x[c * ldx + m] - mean; q += d * d; }
        rstd = rsqrtf(q / (float)C + eps);
    }
    stats[2 * m] = mean; stats[2 * m + 1] = rstd;
}

template <int WRITE>
int dispatch(const float* x, float* stats, float* y, const float* res, const float* w, const float* b,
             int M, int C, int ldx, int mode, float eps, int act, hipStream_t s, int rows_per_group = 0, int gstride = 0,
             const float* amax = nullptr) {
    if (C % 4 != 0 || ldx % 4 != 0 || ((uintptr_t)x & 15)) return PD_ERR_UNSUPPORTED;
    const int nchunk = C / 4;
    // lanes per row = min(64, pow2 >= C/4): one float4 per lane while the row fits a wave
    int lpr = 4;
    while (lpr < 64 && (lpr * MAXV < nchunk || lpr * 2 <= nchunk)) lpr <<= 1;
    if (lpr * MAXV < nchunk) return PD_ERR_UNSUPPORTED;   // C > 1024
#define PD_LAUNCH(L)                                                                               \
    {                                                                                              \
        const int rpb = 256 / L;                                                                   \
        hipLaunchKernelGGL((rownorm_kernel<L, WRITE>), dim3((M + rpb - 1) / rpb), dim3(256), 0, s, \
                           x, stats, y, res, w, b, M, C, ldx, mode, eps, act, rows_per_group, gstride, amax); \
    }
    switch (lpr) {
        case 4: PD_LAUNCH(4) break;
        case 8: PD_LAUNCH(8) break;
        case 16: PD_LAUNCH(16) break;
        case 32: PD_LAUNCH(32) break;
        default: PD_LAUNCH(64) break;
    }
#undef PD_LAUNCH
    return pd_check_launch();
}

}  // namespace

PD_EXPORT int pd_rowstats(const float* x, float* stats, int M, int C, int ldx, int kmajor, int mode, float eps,
                          void* stream) {
    if (!x || !stats || M <= 0 || C <= 0) return PD_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (kmajor) {
        hipLaunchKernelGGL(colstats_kernel, dim3((M + 255) / 256), dim3(256), 0, s, x, stats, M, C, (long long)ldx,
                           mode, eps);
        return pd_check_launch();
    }
    return dispatch<0>(x, stats, nullptr, nullptr, nullptr, nullptr, M, C, ldx, mode, eps, 0, s);
}

PD_EXPORT int pd_rownorm(const float* x, float* y, const float* res, const float* w, const float* b, int M, int C,
                         int mode, float eps, int act, void* stream) {
    if (!x || !y || M <= 0 || C <= 0) return PD_ERR_ARG;
    return dispatch<1>(x, nullptr, y, res, w, b, M, C, C, mode, eps, act, (hipStream_t)stream);
}

PD_EXPORT int pd_norm_split(const float* x, int ldx, int M, int C, int mode, float eps, const float* w, const float* b,
                            int rows_per_group, int gstride, void* out3, void* stream) {
    if (!x || !out3 || M <= 0 || C <= 0) return PD_ERR_ARG;
    if (C % 32 != 0 || ((uintptr_t)out3 & 15)) return PD_ERR_UNSUPPORTED;      // rows of whole 32-k slices, 16-byte aligned
    if (((uintptr_t)w & 15) || ((uintptr_t)b & 15) || gstride % 4 != 0) return PD_ERR_UNSUPPORTED;      // gain / shift rows as 16-byte vectors
    return dispatch<2>(x, nullptr, reinterpret_cast<float*>(out3), nullptr, w, b, M, C, ldx, mode, eps, 0, (hipStream_t)stream,
                       rows_per_group, gstride);
}

PD_EXPORT int pd_norm_split2(const float* x, int ldx, int M, int C, int mode, float eps, const float* w, const float* b,
                             int rows_per_group, int gstride, const float* a_amax, void* out2, void* stream) {
    if (!x || !out2 || !a_amax || M <= 0 || C <= 0) return PD_ERR_ARG;
    if (C % 32 != 0 || ((uintptr_t)out2 & 15)) return PD_ERR_UNSUPPORTED;
    if (((uintptr_t)w & 15) || ((uintptr_t)b & 15) || gstride % 4 != 0) return PD_ERR_UNSUPPORTED;      // gain / shift rows as 16-byte vectors
    return dispatch<3>(x, nullptr, reinterpret_cast<float*>(out2), nullptr, w, b, M, C, ldx, mode, eps, 0, (hipStream_t)stream,
                       rows_per_group, gstride, a_amax);
}
