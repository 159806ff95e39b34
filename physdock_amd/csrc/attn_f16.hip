// Biased attention, head width 32, on the fp16 matrix pipe with TWO-part split operands ("f16 x 3") - the second operand
// format of the split-operand attention (the first, three bf16 parts / six products, is attn_split.hip, whose structure,
// LDS layouts, bias fragment layout and accumulator layouts this kernel shares):
//   * every fp32 operand value times a power of two is split into (hi, lo) fp16 parts: 22 significand bits;
//   * three partial products per block (hi.hi, hi.lo, lo.hi): 12 v_mfma_f32_32x32x16_f16 per 32-key sub-tile instead of 24
//     bf16 MFMAs, 6 instead of 11 VALU operations per split pair, two thirds of the LDS bytes;
//   * the scales come from UPPER BOUNDS of |q|, |k|, |v| (pd_attn_args.f16_amax / f16_*_amax: by value or read from device
//     memory) so that no scaled value overflows fp16; p = exp2(s - m) in [0, 1] is carried times 2^14 inside the exponent.
// Measured against float64 the contraction is at least as accurate as v_mfma_f32_32x32x2_f32 for K >= 32
// (tools/micro/f16x2_probe.hip, profiles/r03_f16x2_probe.txt; tests/test_attention_f16_gpu.py for this kernel).
// The kernel template is written for NP = 2 or 3 parts; only NP = 2 is instantiated here.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "physdock_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int KT = 64;        // keys per LDS tile
constexpr int KP = 40;        // 16-bit elements per K row (80 bytes)
constexpr int VP = 72;        // 16-bit elements per V^T row (144 bytes)
constexpr int K_PART = KT * KP, V_PART = 32 * VP;

template <int NP> struct Parts;
template <> struct Parts<3> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ void split(float a, float b, unsigned (&o)[3]) {
        const pd_parts t = pd_split2(a, b);
        o[0] = t.h; o[1] = t.m; o[2] = t.l;
    }
};
template <> struct Parts<2> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ void split(float a, float b, unsigned (&o)[2]) {
        const pd_parts2 t = pd_split2h(a, b);
        o[0] = t.h; o[1] = t.l;
    }
};

// c += a . b through the partial products that matter, smallest first (fp32 accumulation in the matrix pipe)
template <int NP>
__device__ __forceinline__ f32x16 contract(const typename Parts<NP>::frag (&a)[NP], const typename Parts<NP>::frag (&b)[NP], f32x16 c) {
    typedef Parts<NP> P;
    if constexpr (NP == 3) {
        c = P::mfma(a[0], b[2], c);
        c = P::mfma(a[2], b[0], c);
        c = P::mfma(a[1], b[1], c);
    }
    c = P::mfma(a[0], b[1], c);
    c = P::mfma(a[1], b[0], c);
    c = P::mfma(a[0], b[0], c);
    return c;
}

// position of key k (0..31 inside a sub-tile) in a V^T row: lane half hh and k-step s of the P operand hold, in register
// order, the keys (r&3) + 8(r>>2) + 4hh, r = 8s .. 8s+7
__device__ __forceinline__ int vpos(int k) {
    const int s = k >> 4, j = k & 15;
    return 16 * s + 8 * ((j >> 2) & 1) + 4 * (j >> 3) + (j & 3);
}

// PRE: K and V arrive already scaled and split (pd_attn_args.K2 / V2, written by the q|k|v projection's epilogue): staging copies
// SPLIT: the key range is cut into p.nsplit chunks (blockIdx.y = query block * nsplit + chunk), partial results to p.ws in the
// format of attention.hip's attn_combine_kernel - the fp16-parts form of its key-split launch for a handful of samples
template <int NW, int NP, bool PRE = false, bool SPLIT = false>
__global__ __launch_bounds__(64 * NW, 2) __attribute__((amdgpu_waves_per_eu(NW == 8 ? 4 : 2, 4))) void attn_parts_kernel(const pd_attn_args p) {
    typedef Parts<NP> PT;
    typedef typename PT::frag frag;
    constexpr int STAGE = NP * (K_PART + V_PART);      // 16-bit elements per stage
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int b = blockIdx.x, h = blockIdx.z;
    const int qb = SPLIT ? blockIdx.y / p.nsplit : blockIdx.y, chunk = SPLIT ? blockIdx.y % p.nsplit : 0;
    const int q0 = qb * (32 * NW) + wave * 32;
    const int query = q0 + l31;
    const bool wave_active = q0 < p.nq;

    const float* Kb = p.K + (long long)b * p.k_bs + h * 32;
    const float* Vb = p.V + (long long)b * p.v_bs + h * 32;

    // fp16 format: power-of-two scales that bring the operands' upper bounds to 2^14 (the launcher guarantees the bounds)
    float sk = 1.f, sv = 1.f, c_s = 1.f, inv_sv = 1.f, qs = p.scale * PD_LOG2E;
    if constexpr (NP == 2) {
        const float sq = pd_pow2_scale(p.f16_amax ? p.f16_amax[0] * qs : p.f16_q_amax * qs);
        sk = pd_pow2_scale(p.f16_amax ? p.f16_amax[1] : p.f16_k_amax);
        sv = pd_pow2_scale(p.f16_amax ? p.f16_amax[2] : p.f16_v_amax);
        qs *= sq;
        c_s = 1.0f / (sq * sk);                        // exact: powers of two
        inv_sv = 1.0f / sv;
    }

    // Q fragments: k-step s covers dims 16 s + 8 hh .. + 8 of the lane's query
    frag qf[2][NP];
    {
        const float* qp = p.Q + (long long)b * p.q_bs + (long long)query * p.q_ss + h * 32 + 8 * hh;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
            if (query < p.nq) {
                v0 = *reinterpret_cast<const f32x4*>(qp + 16 * s);
                v1 = *reinterpret_cast<const f32x4*>(qp + 16 * s + 4);
            }
            u32x4 f[NP];
            unsigned t[NP];
            PT::split(v0[0] * qs, v0[1] * qs, t);
#pragma unroll
            for (int k = 0; k < NP; ++k) f[k][0] = t[k];
            PT::split(v0[2] * qs, v0[3] * qs, t);
#pragma unroll
            for (int k = 0; k < NP; ++k) f[k][1] = t[k];
            PT::split(v1[0] * qs, v1[1] * qs, t);
#pragma unroll
            for (int k = 0; k < NP; ++k) f[k][2] = t[k];
            PT::split(v1[2] * qs, v1[3] * qs, t);
#pragma unroll
            for (int k = 0; k < NP; ++k) f[k][3] = t[k];
#pragma unroll
            for (int k = 0; k < NP; ++k) qf[s][k] = __builtin_bit_cast(frag, f[k]);
        }
    }

    const int nkt32 = ((p.bias_nk > 0 ? p.bias_nk : p.nk) + 31) >> 5;
    const int nqt32 = (p.nq + 31) >> 5;
    const float* bias_wave = nullptr;
    if (p.bias && wave_active)
        bias_wave = p.bias + (((long long)h * nqt32 + (q0 >> 5)) * nkt32) * 1024 + lane * 4;

    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // staging: thread -> key row (tid >> 3) + RPP i, 4 dims at 4 (tid & 7)
    constexpr int RPP = 8 * NW, NST = KT / RPP;
    const int srow = tid >> 3, sc = tid & 7;
    f32x4 rk[NST], rv[NST];
    // PRE: one 16-byte chunk per tensor holds (hi(e0,e1), hi(e2,e3), lo(e0,e1), lo(e2,e3)) of the thread's four dims
    // (pd_gemm_args.Y2 row layout): the same registers, the same address arithmetic as the fp32 rows, no scale / split VALU
    const float* K2b = reinterpret_cast<const float*>(p.K2) + (((long long)b * p.kv2_bs + h * 64) >> 1) + 4 * sc;
    const float* V2b = reinterpret_cast<const float*>(p.V2) + (((long long)b * p.kv2_bs + h * 64) >> 1) + 4 * sc;
    auto gload = [&](int key0) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int key = key0 + srow + RPP * i;
            rk[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            rv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (key < p.nk) {
                if constexpr (PRE) {
                    rk[i] = *reinterpret_cast<const f32x4*>(K2b + (((long long)key * p.kv2_ss) >> 1));
                    rv[i] = *reinterpret_cast<const f32x4*>(V2b + (((long long)key * p.kv2_ss) >> 1));
                } else {
                    rk[i] = *reinterpret_cast<const f32x4*>(Kb + (long long)key * p.k_ss + 4 * sc);
                    rv[i] = *reinterpret_cast<const f32x4*>(Vb + (long long)key * p.v_ss + 4 * sc);
                }
            }
        }
    };
    auto sstore = [&](int st) {
        unsigned short* sK = lds + st * STAGE;
        unsigned short* sV = sK + NP * K_PART;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int kr = srow + RPP * i;                         // key row inside the tile
            unsigned k0[NP], k1[NP], v0[NP], v1[NP];               // packed pairs: (e0, e1), (e2, e3)
            if constexpr (PRE) {
                static_assert(!PRE || NP == 2, "pre-split K / V: two fp16 parts");
                const u32x4 kb = __builtin_bit_cast(u32x4, rk[i]), vb = __builtin_bit_cast(u32x4, rv[i]);
                k0[0] = kb[0]; k1[0] = kb[1]; k0[NP - 1] = kb[2]; k1[NP - 1] = kb[3];
                v0[0] = vb[0]; v1[0] = vb[1]; v0[NP - 1] = vb[2]; v1[NP - 1] = vb[3];
            } else if constexpr (NP == 2) {
#if defined(PD_ATTN_ABL) && PD_ATTN_ABL == 1      // lab ablation (wrong results): K / V staged without scale + split VALU work
                k0[0] = __float_as_uint(rk[i][0]); k0[1] = __float_as_uint(rk[i][1]); k1[0] = __float_as_uint(rk[i][2]); k1[1] = __float_as_uint(rk[i][3]);
                v0[0] = __float_as_uint(rv[i][0]); v0[1] = __float_as_uint(rv[i][1]); v1[0] = __float_as_uint(rv[i][2]); v1[1] = __float_as_uint(rv[i][3]);
#else
                PT::split(rk[i][0] * sk, rk[i][1] * sk, k0); PT::split(rk[i][2] * sk, rk[i][3] * sk, k1);
                PT::split(rv[i][0] * sv, rv[i][1] * sv, v0); PT::split(rv[i][2] * sv, rv[i][3] * sv, v1);
#endif
            } else {
                PT::split(rk[i][0], rk[i][1], k0); PT::split(rk[i][2], rk[i][3], k1);
                PT::split(rv[i][0], rv[i][1], v0); PT::split(rv[i][2], rv[i][3], v1);
            }
            const int ko = kr * KP + 4 * sc;
#pragma unroll
            for (int k = 0; k < NP; ++k) *reinterpret_cast<u32x2*>(sK + k * K_PART + ko) = u32x2{k0[k], k1[k]};
            const int vo = (kr & 32) + vpos(kr & 31);              // column of this key in the transposed tile
#pragma unroll
            for (int e = 0; e < 4; ++e) {                          // transposed scatter: dim 4 sc + e, column vo
                const int ro = (4 * sc + e) * VP + vo;
                const int sh = 16 * (e & 1);
#pragma unroll
                for (int k = 0; k < NP; ++k) sV[k * V_PART + ro] = (unsigned short)((e < 2 ? v0[k] : v1[k]) >> sh);
            }
        }
    };

    const int nit_all = (p.nk + KT - 1) / KT;
    const int it_lo = SPLIT ? (int)((long long)nit_all * chunk / p.nsplit) : 0;
    const int nit = SPLIT ? (int)((long long)nit_all * (chunk + 1) / p.nsplit) : nit_all;
    gload(it_lo * KT);
    sstore(0);
    __syncthreads();

    auto subtile = [&](auto ragged_tag, int cur, int sub, int kt32) {
        constexpr bool RAGGED = decltype(ragged_tag)::value;
        const unsigned short* sK = lds + cur * STAGE;
        const unsigned short* sV = sK + NP * K_PART;
        f32x4 bf[4];
        if (bias_wave) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bf[g] = *reinterpret_cast<const f32x4*>(bias_wave + (long long)kt32 * 1024 + g * 256);
        }
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const unsigned short* kbase = sK + (sub * 32 + l31) * KP + 8 * hh;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            frag kf[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) kf[k] = *reinterpret_cast<const frag*>(kbase + k * K_PART + 16 * st);
            s = contract<NP>(kf, qf[st], s);
        }
        if constexpr (NP == 2) {                           // undo the operand scales (exact), then the bias
            if (bias_wave) {
                if constexpr (SPLIT) {
                    // the key-split form also serves the tail round of a pipelined launch (pd_attention_tail), whose bias tiles were
                    // produced times the product of the q and k operand scales: times c_s that is the plain bias again, exactly
                    if (p.bias_prescale > 0.f) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) bf[g] *= c_s;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = __builtin_fmaf(s[r], c_s, bf[r >> 2][r & 3]);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] *= c_s;
            }
        } else if (bias_wave) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] += bf[r >> 2][r & 3];
        }
        if constexpr (RAGGED) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kt32 * 32 + pd_frag_row(r, hh) >= p.nk) s[r] = -INFINITY;
        }
        float mloc = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s[r]);
        mloc = pd_xhalf_max(mloc);
#if defined(PD_ATTN_LAZY)      // lab experiment: the reference maximum moves only when some row's maximum exceeds it by more than
        // PD_ATTN_LAZY (log2 units); p is carried times 2^12 so that 2^PD_ATTN_LAZY of head room exists below the fp16 maximum
        float alpha = 1.0f;
        if (__builtin_amdgcn_ballot_w64(mloc > m_run + (float)PD_ATTN_LAZY) != 0ull) {
            const float m_new = fmaxf(m_run, mloc);
            alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] *= alpha;
        }
        const float m_new = m_run;
        const float m_exp = NP == 2 ? m_new - 12.0f : m_new;
#else
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
#pragma unroll
#if !(defined(PD_ATTN_ABL) && PD_ATTN_ABL == 2)   // lab ablation 2 (wrong results): no rescale of the accumulator
        for (int r = 0; r < 16; ++r) o[r] *= alpha;      // (a wave-uniform "no maximum moved" skip measured -20 %: it splits the schedule)
#endif
        // fp16 parts: p is carried times 2^14 (inside the exponent), so that its low part stays a normal fp16 number down
        // to p = 2^-16; the sum l carries the same factor and it cancels in o / l
        const float m_exp = NP == 2 ? m_new - 14.0f : m_new;
#endif
        float psum = 0.f;
        const unsigned short* vbase = sV + l31 * VP + sub * 32 + 8 * hh;
        // one k-step (8 of the lane's 16 keys) at a time: exp, split, the MFMAs - the probabilities of the second half are
        // computed while the matrix pipe works on the first, and only one set of P fragments is live
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            u32x4 f[NP];
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                const float p0 = __builtin_amdgcn_exp2f(s[8 * st + 2 * e2] - m_exp);
                const float p1 = __builtin_amdgcn_exp2f(s[8 * st + 2 * e2 + 1] - m_exp);
                psum += p0 + p1;
                unsigned t[NP];
                PT::split(p0, p1, t);
#pragma unroll
                for (int k = 0; k < NP; ++k) f[k][e2] = t[k];
            }
            frag pf[NP], vf[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                pf[k] = __builtin_bit_cast(frag, f[k]);
                vf[k] = *reinterpret_cast<const frag*>(vbase + k * V_PART + 16 * st);
            }
            o = contract<NP>(vf, pf, o);
        }
        l_run = l_run * alpha + psum;
    };
    const int nfull32 = p.nk >> 5;

    for (int it = it_lo; it < nit; ++it) {
        const int cur = (it - it_lo) & 1;
        if (it + 1 < nit) gload((it + 1) * KT);
        if (wave_active) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int kt32 = it * 2 + sub;
                if (kt32 < nfull32) subtile(std::false_type{}, cur, sub, kt32);
                else if (kt32 * 32 < p.nk) subtile(std::true_type{}, cur, sub, kt32);
            }
        }
        if (it + 1 < nit) sstore(cur ^ 1);
        __syncthreads();
    }

    if constexpr (SPLIT) {
        if (query < p.nq) {      // partial result of this key chunk: [chunk][b][query][h*32 + dim], (m, l) per (chunk, b, h, query);
            // o and l both carry the 2^14 of the p format (it cancels in the combine), o additionally the V scale: undone here
            const float l = pd_xhalf_sum(l_run);
            const int C = p.nheads * 32;
            float* wo = p.ws + (((long long)chunk * p.nbatch + b) * p.nq + query) * C + h * 32 + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {o[4 * g] * inv_sv, o[4 * g + 1] * inv_sv, o[4 * g + 2] * inv_sv, o[4 * g + 3] * inv_sv};
                *reinterpret_cast<f32x4*>(wo + 8 * g) = v;
            }
            if (hh == 0) {
                float* ml = p.ws + (long long)p.nsplit * p.nbatch * p.nq * C
                            + ((((long long)chunk * p.nbatch + b) * p.nheads + h) * p.nq + query) * 2;
                ml[0] = m_run; ml[1] = l;
            }
        }
        return;
    }
    if (query < p.nq) {
        const float l = pd_xhalf_sum(l_run);
        if (NP == 2 && p.O2) {
            // output already split for the projection that follows (pd_gemm_args.A2): o times the V scale (|o| <= max|v|, the
            // same bound and hence the same power of two the GEMM derives from f16_amax[2]) is exactly O' / l'
            const float inv = 1.0f / l;
            const long long rows = p.o2_rows > 0 ? p.o2_rows : (long long)p.nbatch * p.nq, C = (long long)p.nheads * 32;
            unsigned short* op = reinterpret_cast<unsigned short*>(p.O2) + ((long long)b * p.nq + query) * C + h * 32 + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const pd_parts2 p0 = pd_split2h(o[4 * g] * inv, o[4 * g + 1] * inv), p1 = pd_split2h(o[4 * g + 2] * inv, o[4 * g + 3] * inv);
                *reinterpret_cast<u32x2*>(op + 8 * g) = u32x2{p0.h, p1.h};
                *reinterpret_cast<u32x2*>(op + rows * C + 8 * g) = u32x2{p0.l, p1.l};
            }
        } else {
            const float inv = NP == 2 ? inv_sv / l : 1.0f / l;
            float* op = p.O + (long long)b * p.o_bs + (long long)query * p.o_ss + h * 32 + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv};
                *reinterpret_cast<f32x4*>(op + 8 * g) = v;
            }
        }
    }
}

template <int NP> constexpr int lds_bytes() { return 2 * NP * (K_PART + V_PART) * 2; }

template <int NW, int NP, bool PRE>
bool raise_lds() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(attn_parts_kernel<NW, NP, PRE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               lds_bytes<NP>()) == hipSuccess;
}

template <int NP, bool PRE>
void launch(const pd_attn_args* a, hipStream_t stream) {
    if (a->nq > 128) {
        dim3 grid(a->nbatch, (a->nq + 255) / 256, a->nheads);
        hipLaunchKernelGGL((attn_parts_kernel<8, NP, PRE>), grid, dim3(512), lds_bytes<NP>(), stream, *a);
    } else {
        dim3 grid(a->nbatch, 1, a->nheads);
        hipLaunchKernelGGL((attn_parts_kernel<4, NP, PRE>), grid, dim3(256), lds_bytes<NP>(), stream, *a);
    }
}

}  // namespace

// key-split launch of the fp16-parts kernel (pd_attention, a->nsplit chunks, 4-wave blocks as the fp32 form): partial results
// to a->ws; the caller runs attn_combine_kernel afterwards.  init_only: 1 raise the dynamic-LDS limit.
extern "C" int pd_attention_f16_split(const pd_attn_args* a, void* stream, int init_only) {
    auto k = attn_parts_kernel<4, 2, false, true>;
    auto kpre = attn_parts_kernel<4, 2, true, true>;          // K / V pre-split by the projection (the tail round of a DiT launch)
    if (init_only == 1)
        return hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes<2>()) == hipSuccess &&
                       hipFuncSetAttribute(reinterpret_cast<const void*>(kpre), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes<2>()) == hipSuccess
                   ? PD_OK : PD_ERR_LAUNCH;
    if (!a->f16_amax && !(a->f16_q_amax > 0.f && a->f16_k_amax > 0.f && a->f16_v_amax > 0.f)) return PD_ERR_ARG;
    if (a->nsplit < 2 || !a->ws) return PD_ERR_UNSUPPORTED;       // (O / O2 are the combine kernel's business: the chunks go to ws)
    if (a->K2 && (!a->V2 || (((uintptr_t)a->K2 | (uintptr_t)a->V2) & 15) || a->kv2_ss % 8 || a->kv2_bs % 8)) return PD_ERR_ARG;
    dim3 grid(a->nbatch, ((a->nq + 127) / 128) * a->nsplit, a->nheads);
    if (a->K2) hipLaunchKernelGGL(kpre, grid, dim3(256), lds_bytes<2>(), (hipStream_t)stream, *a);
    else hipLaunchKernelGGL(k, grid, dim3(256), lds_bytes<2>(), (hipStream_t)stream, *a);
    return pd_check_launch();
}

// init_only: 1 raise the dynamic-LDS limits; 0 launch.  Called by pd_attention_split_try (attn_split.hip) for f16x3 launches.
extern "C" int pd_attention_f16_try(const pd_attn_args* a, void* stream, int init_only) {
    if (init_only == 1)
        return raise_lds<8, 2, false>() && raise_lds<4, 2, false>() && raise_lds<8, 2, true>() && raise_lds<4, 2, true>() &&
                       pd_attention_f16_split(nullptr, nullptr, 1) == PD_OK ? PD_OK : PD_ERR_LAUNCH;
    // the fp16 format needs finite positive magnitude bounds for q, k, v: by value or in device memory (f16_amax[3])
    if (!a->f16_amax && !(a->f16_q_amax > 0.f && a->f16_k_amax > 0.f && a->f16_v_amax > 0.f)) return PD_ERR_ARG;
    if (a->O2 && (((uintptr_t)a->O2 & 15) || a->o_ss != (long long)a->nheads * 32 || a->o_bs != (long long)a->nq * a->o_ss))
        return PD_ERR_ARG;                       // the split output is a dense [rows][C] operand
    if (a->K2) {                                 // pre-split K / V: 16-byte chunks (four dims, both parts)
        if (!a->V2 || (((uintptr_t)a->K2 | (uintptr_t)a->V2) & 15) || a->kv2_ss % 8 || a->kv2_bs % 8) return PD_ERR_ARG;
        launch<2, true>(a, (hipStream_t)stream);
    } else {
        launch<2, false>(a, (hipStream_t)stream);
    }
    return pd_check_launch();
}
