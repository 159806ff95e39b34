// The triangle-multiplication einsum of the trunk's TriangleUpdate (reference primitives/attentions.py:164):
//   outgoing (row) form      o[c,i,I] = sum_j q[c,i,j] k[c,I,j]
//   incoming (column) form   o[c,a,b] = sum_j k[c,j,a] q[c,j,b]      (the reference transposes z, runs the row form, transposes back)
// for the 32 channel-major planes q, k [c][T][T] the gated projection writes: 32 independent T x T x T contractions of two
// ACTIVATION operands.  They ran on the generic fp32-MFMA kernel (k-major operands for the column form) at 40 - 58 TF; here they
// take the two-part fp16 operand format (three products per block, bounds of |q|, |k| from the projection weights and the norm
// gain): one wave = one channel x one 32 x 32 output tile; per 32-j slice a lane fetches 16 values of each operand with
// coalesced 16-byte loads (row form: 8 lanes = one 128-byte row piece; column form: the [j][a] tile as it lies in memory),
// scales, splits and writes them to a wave-private LDS tile - TRANSPOSED for the column form, so that both forms read their MFMA
// fragments (8 consecutive j of one output row / column) with the same ds_read_b128.  No block barriers: the tile is private to
// the wave and LDS operations of one wave execute in order.
#include "common.h"
#include "physdock_hip.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int TP = 40;                  // LDS row pitch in fp16 (80 bytes: conflict-free ds_read_b128)
constexpr int PART = 32 * TP;           // one part of one operand tile
constexpr int WAVE_LDS = 4 * PART;      // (A, B) x (hi, lo), fp16 elements
constexpr int NWAVES = 4;

template <bool TR>
__global__ __launch_bounds__(64 * NWAVES) void tri_mul_kernel(const pd_tri_mul_args p) {
    __shared__ __attribute__((aligned(16))) unsigned short lds_all[NWAVES * WAVE_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    unsigned short* lds = lds_all + wave * WAVE_LDS;
    const int nt = (p.T + 31) >> 5;
    const int ti = blockIdx.x / nt, tI = blockIdx.x % nt;
    const int c = blockIdx.y * NWAVES + wave;
    if (c >= p.nch) return;
    const int i0 = ti * 32, I0 = tI * 32;
    const float sa = pd_pow2_scale(TR ? *p.k_amax : *p.q_amax), sb = pd_pow2_scale(TR ? *p.q_amax : *p.k_amax);
    const float inv = 1.0f / (sa * sb);
    // operand A supplies the output rows, operand B the output columns
    const float* Ap = (TR ? p.k : p.q) + (long long)c * p.ch_stride;
    const float* Bp = (TR ? p.q : p.k) + (long long)c * p.ch_stride;
    const int lr = lane >> 3, lq = lane & 7;     // fetch pattern: 8 lanes x 4 floats = 32 consecutive elements of one memory row
    f32x4 ra[4], rb[4];
    auto gload = [&](int j0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
            if constexpr (!TR) {                 // memory row = output row / column, 4 consecutive j
                const int r = lr + 8 * t, j = j0 + 4 * lq;
                if (i0 + r < p.T && j < p.Treal) va = *reinterpret_cast<const f32x4*>(Ap + (long long)(i0 + r) * p.T + j);
                if (I0 + r < p.T && j < p.Treal) vb = *reinterpret_cast<const f32x4*>(Bp + (long long)(I0 + r) * p.T + j);
                if (j + 3 >= p.Treal) {          // last partial quad: padded j never enter the reduction (as K = Treal did in the fp32 GEMM)
#pragma unroll
                    for (int e = 1; e < 4; ++e) if (j + e >= p.Treal) { va[e] = 0.f; vb[e] = 0.f; }
                }
            } else {                             // memory row = j, 4 consecutive output rows / columns
                const int j = j0 + lr + 8 * t;
                if (j < p.Treal && i0 + 4 * lq < p.T) va = *reinterpret_cast<const f32x4*>(Ap + (long long)j * p.T + i0 + 4 * lq);
                if (j < p.Treal && I0 + 4 * lq < p.T) vb = *reinterpret_cast<const f32x4*>(Bp + (long long)j * p.T + I0 + 4 * lq);
            }
            ra[t] = va; rb[t] = vb;
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int op = 0; op < 2; ++op) {
                const f32x4 v = op ? rb[t] : ra[t];
                const float s = op ? sb : sa;
                const pd_parts2 p0 = pd_split2h(v[0] * s, v[1] * s), p1 = pd_split2h(v[2] * s, v[3] * s);
                unsigned short* base = lds + op * 2 * PART;
                if constexpr (!TR) {
                    const int o = (lr + 8 * t) * TP + 4 * lq;
                    *reinterpret_cast<u32x2*>(base + o) = u32x2{p0.h, p1.h};
                    *reinterpret_cast<u32x2*>(base + PART + o) = u32x2{p0.l, p1.l};
                } else {                         // transposed scatter: tile row = 4 lq + e, column = j - j0
                    const int col = lr + 8 * t;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned h = e < 2 ? p0.h : p1.h, l = e < 2 ? p0.l : p1.l;
                        const int sh = 16 * (e & 1);
                        base[(4 * lq + e) * TP + col] = (unsigned short)(h >> sh);
                        base[PART + (4 * lq + e) * TP + col] = (unsigned short)(l >> sh);
                    }
                }
            }
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nsl = (p.Treal + 31) >> 5;
    gload(0);
    for (int s = 0; s < nsl; ++s) {
        sstore();
        if (s + 1 < nsl) gload((s + 1) * 32);
        __builtin_amdgcn_s_waitcnt(0xc07f);      // this wave's LDS writes have landed (the tile is private to the wave)
        const unsigned short* fa = lds + l31 * TP + 8 * hh;
        const unsigned short* fb = lds + 2 * PART + l31 * TP + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(fa + 16 * ks), al = *reinterpret_cast<const f16x8*>(fa + PART + 16 * ks);
            const f16x8 bh = *reinterpret_cast<const f16x8*>(fb + 16 * ks), bl = *reinterpret_cast<const f16x8*>(fb + PART + 16 * ks);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);      // fragment reads done before the next slice overwrites the tile
    }
    // lane = output column, register = output row: 128-byte row segments
    float* Op = p.o + (long long)c * p.ch_stride;
    if (I0 + l31 < p.T) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i0 + pd_frag_row(r, hh);
            if (row < p.T) Op[(long long)row * p.T + I0 + l31] = acc[r] * inv;
        }
    }
}

}  // namespace

PD_EXPORT int pd_tri_mul(const pd_tri_mul_args* a, void* stream) {
    if (!a || !a->q || !a->k || !a->o || !a->q_amax || !a->k_amax) return PD_ERR_ARG;
    if (a->T <= 0 || a->Treal <= 0 || a->Treal > a->T || a->nch <= 0 || a->T % 4 != 0 || a->ch_stride % 4 != 0) return PD_ERR_UNSUPPORTED;
    if (((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->o) & 15) return PD_ERR_UNSUPPORTED;
    const int nt = (a->T + 31) / 32;
    dim3 grid(nt * nt, (a->nch + NWAVES - 1) / NWAVES);
    if (a->transpose) hipLaunchKernelGGL(tri_mul_kernel<true>, grid, dim3(64 * NWAVES), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL(tri_mul_kernel<false>, grid, dim3(64 * NWAVES), 0, (hipStream_t)stream, *a);
    return pd_check_launch();
}
