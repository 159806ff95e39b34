// Biased attention, head width 32, on the bf16 matrix pipe at fp32 accuracy (3-way error-free operand split, six
// partial products per block, fp32 accumulation - see gemm_split.hip for the numerics and tools/micro/bf16x3_probe.hip
// for the measurement: at least the accuracy of v_mfma_f32_32x32x2_f32).
//
// Same contract, same flash structure, same bias fragment layout and the same accumulator layouts as attention.hip
// (S^T = K.Q^T and O^T = V^T.P^T, lane = one query, softmax in registers in fp32); what changes is how the two
// contractions are fed:
//   * Q is scaled by scale*log2(e), split into (hi, mid, lo) bf16 fragments once per wave and kept in registers;
//   * K is split while it is staged: LDS holds K as three bf16 [64 keys][32 dims] tiles (80-byte rows: the ds_read_b128
//     of a fragment is bank-conflict free) - the A operand of v_mfma_f32_32x32x16_bf16 is 8 consecutive dims of one key;
//   * V is split AND transposed while it is staged: three bf16 [32 dims][64 keys] tiles (144-byte rows) with the keys of
//     each 32-key sub-tile stored in the order the softmax leaves them in the accumulator registers, so that the lane's
//     16 probabilities are, as they stand, the B operand of the two k-steps of O^T += V^T.P^T and the matching A
//     operand (8 keys of one dim) is one contiguous 16-byte read;
//   * P is split in registers (p in [0,1]: hi + mid + lo is exact).
// Per 32-key sub-tile a wave issues 24 bf16 MFMAs of 32 cycles instead of 32 fp32 MFMAs of 64 cycles.
// Blocks are 8 waves = 256 queries (two per CU, four waves per SIMD) or 4 waves for short query ranges.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "physdock_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int KT = 64;        // keys per LDS tile
constexpr int KP = 40;        // bf16 per K row (80 bytes)
constexpr int VP = 72;        // bf16 per V^T row (144 bytes)
constexpr int K_PART = KT * KP, V_PART = 32 * VP;
constexpr int STAGE = 3 * (K_PART + V_PART);       // bf16 per stage (29 184 bytes)

__device__ __forceinline__ void split1(float v, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)v;
    const float r1 = v - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

// position of key k (0..31 inside a sub-tile) in a V^T row: lane half hh and k-step s of the P operand hold, in register
// order, the keys (r&3) + 8(r>>2) + 4hh, r = 8s .. 8s+7
__device__ __forceinline__ int vpos(int k) {
    const int s = k >> 4, j = k & 15;
    return 16 * s + 8 * ((j >> 2) & 1) + 4 * (j >> 3) + (j & 3);
}

template <int NW>
__global__ __launch_bounds__(64 * NW, 2) __attribute__((amdgpu_waves_per_eu(NW == 8 ? 4 : 2, 4))) void attn_split_kernel(const pd_attn_args p) {
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int b = blockIdx.x, h = blockIdx.z, qb = blockIdx.y;
    const int q0 = qb * (32 * NW) + wave * 32;
    const int query = q0 + l31;
    const bool wave_active = q0 < p.nq;

    const float* Kb = p.K + (long long)b * p.k_bs + h * 32;
    const float* Vb = p.V + (long long)b * p.v_bs + h * 32;

    // Q fragments: k-step s covers dims 16 s + 8 hh .. + 8 of the lane's query
    bf16x8 qh[2], qm[2], ql[2];
    {
        const float qs = p.scale * PD_LOG2E;
        const float* qp = p.Q + (long long)b * p.q_bs + (long long)query * p.q_ss + h * 32 + 8 * hh;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
            if (query < p.nq) {
                v0 = *reinterpret_cast<const f32x4*>(qp + 16 * s);
                v1 = *reinterpret_cast<const f32x4*>(qp + 16 * s + 4);
            }
            u32x4 fh, fm, fl;
            { const pd_parts t_ = pd_split2(v0[0] * qs, v0[1] * qs); fh[0] = t_.h; fm[0] = t_.m; fl[0] = t_.l; }
            { const pd_parts t_ = pd_split2(v0[2] * qs, v0[3] * qs); fh[1] = t_.h; fm[1] = t_.m; fl[1] = t_.l; }
            { const pd_parts t_ = pd_split2(v1[0] * qs, v1[1] * qs); fh[2] = t_.h; fm[2] = t_.m; fl[2] = t_.l; }
            { const pd_parts t_ = pd_split2(v1[2] * qs, v1[3] * qs); fh[3] = t_.h; fm[3] = t_.m; fl[3] = t_.l; }
            qh[s] = __builtin_bit_cast(bf16x8, fh); qm[s] = __builtin_bit_cast(bf16x8, fm); ql[s] = __builtin_bit_cast(bf16x8, fl);
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
    auto gload = [&](int key0) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int key = key0 + srow + RPP * i;
            rk[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            rv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (key < p.nk) {
                rk[i] = *reinterpret_cast<const f32x4*>(Kb + (long long)key * p.k_ss + 4 * sc);
                rv[i] = *reinterpret_cast<const f32x4*>(Vb + (long long)key * p.v_ss + 4 * sc);
            }
        }
    };
    auto sstore = [&](int st) {
        __bf16* sK = lds + st * STAGE;
        __bf16* sV = sK + 3 * K_PART;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int kr = srow + RPP * i;                         // key row inside the tile
            u32x2 kh, km, kl, vh, vm, vl;                          // packed pairs: (e0, e1), (e2, e3)
            { const pd_parts t_ = pd_split2(rk[i][0], rk[i][1]); kh[0] = t_.h; km[0] = t_.m; kl[0] = t_.l; }
            { const pd_parts t_ = pd_split2(rk[i][2], rk[i][3]); kh[1] = t_.h; km[1] = t_.m; kl[1] = t_.l; }
            { const pd_parts t_ = pd_split2(rv[i][0], rv[i][1]); vh[0] = t_.h; vm[0] = t_.m; vl[0] = t_.l; }
            { const pd_parts t_ = pd_split2(rv[i][2], rv[i][3]); vh[1] = t_.h; vm[1] = t_.m; vl[1] = t_.l; }
            const int ko = kr * KP + 4 * sc;
            *reinterpret_cast<u32x2*>(sK + ko) = kh;
            *reinterpret_cast<u32x2*>(sK + K_PART + ko) = km;
            *reinterpret_cast<u32x2*>(sK + 2 * K_PART + ko) = kl;
            const int vo = (kr & 32) + vpos(kr & 31);              // column of this key in the transposed tile
            unsigned short* sV16 = reinterpret_cast<unsigned short*>(sV);
#pragma unroll
            for (int e = 0; e < 4; ++e) {                          // transposed scatter: dim 4 sc + e, column vo
                const int ro = (4 * sc + e) * VP + vo;
                const int sh = 16 * (e & 1);
                sV16[ro] = (unsigned short)(vh[e >> 1] >> sh);
                sV16[V_PART + ro] = (unsigned short)(vm[e >> 1] >> sh);
                sV16[2 * V_PART + ro] = (unsigned short)(vl[e >> 1] >> sh);
            }
        }
    };

    const int nit = (p.nk + KT - 1) / KT;
    gload(0);
    sstore(0);
    __syncthreads();

    auto subtile = [&](auto ragged_tag, int cur, int sub, int kt32) {
        constexpr bool RAGGED = decltype(ragged_tag)::value;
        const __bf16* sK = lds + cur * STAGE;
        const __bf16* sV = sK + 3 * K_PART;
        f32x4 bf[4];
        if (bias_wave) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bf[g] = *reinterpret_cast<const f32x4*>(bias_wave + (long long)kt32 * 1024 + g * 256);
        }
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const __bf16* kbase = sK + (sub * 32 + l31) * KP + 8 * hh;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const bf16x8 kh = *reinterpret_cast<const bf16x8*>(kbase + 16 * st);
            const bf16x8 km = *reinterpret_cast<const bf16x8*>(kbase + K_PART + 16 * st);
            const bf16x8 kl = *reinterpret_cast<const bf16x8*>(kbase + 2 * K_PART + 16 * st);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[st], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[st], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(km, qm[st], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qm[st], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(km, qh[st], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[st], s, 0, 0, 0);
        }
        if (bias_wave) {
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
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= alpha;      // (a wave-uniform "no maximum moved" skip measured -20 %: it splits the schedule)
        float psum = 0.f;
        const __bf16* vbase = sV + l31 * VP + sub * 32 + 8 * hh;
        // one k-step (8 of the lane's 16 keys) at a time: exp, split, six MFMAs - the probabilities of the second half are
        // computed while the matrix pipe works on the first, and only one set of P fragments is live
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            u32x4 fh, fm, fl;
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                const float p0 = __builtin_amdgcn_exp2f(s[8 * st + 2 * e2] - m_new);
                const float p1 = __builtin_amdgcn_exp2f(s[8 * st + 2 * e2 + 1] - m_new);
                psum += p0 + p1;
                { const pd_parts t_ = pd_split2(p0, p1); fh[e2] = t_.h; fm[e2] = t_.m; fl[e2] = t_.l; }
            }
            const bf16x8 ph = __builtin_bit_cast(bf16x8, fh), pm = __builtin_bit_cast(bf16x8, fm), pl = __builtin_bit_cast(bf16x8, fl);
            const bf16x8 vh = *reinterpret_cast<const bf16x8*>(vbase + 16 * st);
            const bf16x8 vm = *reinterpret_cast<const bf16x8*>(vbase + V_PART + 16 * st);
            const bf16x8 vl = *reinterpret_cast<const bf16x8*>(vbase + 2 * V_PART + 16 * st);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vm, pm, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pm, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vm, ph, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, o, 0, 0, 0);
        }
        l_run = l_run * alpha + psum;
    };
    const int nfull32 = p.nk >> 5;

    for (int it = 0; it < nit; ++it) {
        const int cur = it & 1;
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

    if (query < p.nq) {
        const float l = pd_xhalf_sum(l_run);
        const float inv = 1.0f / l;
        float* op = p.O + (long long)b * p.o_bs + (long long)query * p.o_ss + h * 32 + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v = {o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv};
            *reinterpret_cast<f32x4*>(op + 8 * g) = v;
        }
    }
}

constexpr int LDS_BYTES = 2 * STAGE * 2;

}  // namespace

// attn_f16.hip: the same structure with two-part fp16 operands (three products per block) for launches that carry bounds
extern "C" int pd_attention_f16_try(const pd_attn_args* a, void* stream, int init_only);

// init_only: 1 raise the dynamic-LDS limits; 0 launch (returns PD_ERR_UNSUPPORTED when the launch should stay on attention.hip)
extern "C" int pd_attention_split_try(const pd_attn_args* a, void* stream, int init_only) {
    if (init_only == 1) {
        if (pd_attention_f16_try(nullptr, nullptr, 1) != PD_OK) return PD_ERR_LAUNCH;
        const bool ok =
            hipFuncSetAttribute(reinterpret_cast<const void*>(attn_split_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                LDS_BYTES) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(attn_split_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                LDS_BYTES) == hipSuccess;
        return ok ? PD_OK : PD_ERR_LAUNCH;
    }
    if (a->fp32_mfma) return PD_ERR_UNSUPPORTED;
    if (a->f16x3) return pd_attention_f16_try(a, stream, 0);
    if (a->nq > 128) {
        dim3 grid(a->nbatch, (a->nq + 255) / 256, a->nheads);
        hipLaunchKernelGGL((attn_split_kernel<8>), grid, dim3(512), LDS_BYTES, (hipStream_t)stream, *a);
    } else {
        dim3 grid(a->nbatch, 1, a->nheads);
        hipLaunchKernelGGL((attn_split_kernel<4>), grid, dim3(256), LDS_BYTES, (hipStream_t)stream, *a);
    }
    return pd_check_launch();
}
