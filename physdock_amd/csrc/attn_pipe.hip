// Biased attention, head width 32, two-part fp16 operands ("f16 x 3", the arithmetic of attn_f16.hip) as a SOFTWARE-PIPELINED
// wave program: round 4's form of the chip-filling launches (DiT atom / token attention, triangle / MSA / pair-biased attention).
//
// attn_f16.hip runs, per wave and 32-key sub-tile, the chain  S = K.Q^T (6 MFMAs) -> + bias -> softmax (146 VALU) -> O += V^T.P
// (6 MFMAs)  strictly in that order, and the per-tile block barrier puts the four waves of a SIMD into the same phase: issue
// counters showed VALU busy 0.75, matrix pipe 0.37, time = their sum (profiles/r03_attn_issue_counters.txt).  Here
//   * the score tile of sub-tile j+1 is computed WHILE the softmax of sub-tile j runs: two score accumulators per wave, every
//     MFMA of the instruction stream is followed by the ~9 VALU instructions that fit under it (order pinned with
//     sched_barrier), so a wave overlaps its own matrix and vector work whatever its neighbours do;
//   * the bias tile is the INITIAL VALUE of the score accumulator (C operand of the first MFMA): it is fetched - a whole phase
//     ahead, into the accumulator registers the previous softmax has just vacated - already multiplied by the product of the
//     q and k operand scales (pd_attn_args.bias_prescale, a power of two the producer folds into its out_scale), which removes
//     the 16 fma of the bias add; the subtraction of the running maximum and the undoing of the operand scales are ONE fma in
//     front of the exp2;
//   * the running maximum is taken with v_max3_f32 (8 instead of 15 instructions);
//   * three LDS stages, ONE block barrier per 64-key tile (the tile after the next one is requested right after the barrier).
// VALU instructions per (wave, sub-tile): 146 -> ~115; layouts (K [64 keys][40] and V^T [32 dims][72] fp16 part planes, bias
// fragments, accumulators) are those of attn_f16.hip / attn_split.hip.
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "physdock_hip.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef f16x8 frag;

constexpr int KT = 64;        // keys per LDS tile
constexpr int KP = 40;        // 16-bit elements per K row (80 bytes: conflict-free ds_read_b128)
constexpr int VP = 72;        // 16-bit elements per V^T row (144 bytes)
constexpr int K_PART = KT * KP, V_PART = 32 * VP;
constexpr int STAGE = 2 * (K_PART + V_PART);       // 16-bit elements per stage (19 456 bytes)
constexpr int NSTAGE = 3;

#define PD_SB() __builtin_amdgcn_sched_barrier(0)

#ifdef PD_LAB      // lab build only (tools/attn_pipe_trace.py): s_memtime stamps along a block's life; 16 slots per wave
__device__ unsigned long long* g_pipe_trace = nullptr;
#define PD_PSTAMP(slot) do { if (ptr_) ptr_[slot] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PD_PSTAMP(slot) do { } while (0)
#endif

// lab ablations (timing only, wrong results): 1 no block barrier in the main loop, 2 exp2 -> identity, 4 no MFMAs in the phases,
// 8 no softmax VALU in the phases, 16 static s_setprio 1 for the second half of the block's waves, 32 no bias fetch in the
// phases, 64 no K / V staging in the main loop
#ifdef PD_PIPE_ABL
constexpr int ABL = PD_PIPE_ABL;
#else
constexpr int ABL = 0;
#endif

// Lazy running maximum: the reference maximum of a row moves only when some row of the wave exceeds it by more than PD_PIPE_LAZY
// (log2 units) - a wave-uniform, rarely taken branch holds the exp2 of the correction and the 17 multiplications of the
// accumulator / row-sum rescale; probabilities may then reach 2^PD_PIPE_LAZY, so they are carried times 2^(14 - PD_PIPE_LAZY).
// 0: the plain update (maximum, correction factor and rescale in every sub-tile).
#ifndef PD_PIPE_LAZY
#define PD_PIPE_LAZY 3
#endif
constexpr int LAZY = PD_PIPE_LAZY;
#ifndef PD_PIPE_XCD
#define PD_PIPE_XCD 0
#endif
constexpr bool XCDMAP = PD_PIPE_XCD != 0;
constexpr float PSH = 14.0f - (float)LAZY;

__device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
// MFMA of a pipelined phase (ablation 4 removes it)
__device__ __forceinline__ f32x16 pmma(frag a, frag b, f32x16 c) {
    if constexpr (ABL & 4) {
        c[0] += __builtin_bit_cast(f32x4, a)[0] * 0.f + __builtin_bit_cast(f32x4, b)[0] * 0.f;      // keep the operands alive
        return c;
    }
    return mma(a, b, c);
}

// position of key k (0..31 inside a sub-tile) in a V^T row (attn_f16.hip)
__device__ __forceinline__ int vpos(int k) {
    const int s = k >> 4, j = k & 15;
    return 16 * s + 8 * ((j >> 2) & 1) + 4 * (j >> 3) + (j & 3);
}

__device__ __forceinline__ void lds_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0); outstanding global loads (bias / next tile) stay in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// PRE: K and V arrive already scaled and split (pd_attn_args.K2 / V2); HASBIAS: bias fragments multiplied by bias_prescale;
// RES (round 6): ALL key tiles of the block resident - launches with at most RES_TILES x 64 keys (token DiT attention, MSA row / pair-biased
// attention of the trunk: 256 keys) request every K / V tile in the prologue, stage them into four LDS tiles (78 KB: still two blocks per
// CU) and pass ONE block barrier; the main loop then has no staging, no requests but the bias tiles, and no barrier.  A four-tile block
// of the streaming form spends two thirds of its life outside the pipelined phases (profiles/r05_attn_pipe_block_life_token_shape.txt).
// Built, correct, and no faster (see PD_PIPE_RES below): kept as a lab form.
constexpr int RES_TILES = 4;
template <int NW, bool PRE, bool HASBIAS, bool RES = false>
__global__ __launch_bounds__(64 * NW, 2) __attribute__((amdgpu_waves_per_eu(NW == 8 ? 4 : 2, 4))) void attn_pipe_kernel(const pd_attn_args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    int b = blockIdx.x, h = blockIdx.z, qb = blockIdx.y;
    if constexpr (XCDMAP) {
        // Workgroups go to the eight XCDs round-robin in dispatch order (x fastest).  With the plain (sample, query block, head)
        // grid XCD k gets the samples b = k mod 8 of EVERY (head, query block): each of the eight L2s fetches the whole bias.
        // Here an XCD owns P / 8 (head, query block) pairs - for the atom shape four query blocks of one head - for all samples:
        // it fetches 1 / 8 of the bias per round of co-resident samples, and a (sample, head) K / V pair goes to two L2s.
        const int nqb = gridDim.y, P = nqb * gridDim.z;
        if ((P & 7) == 0) {
            const int L = blockIdx.x + gridDim.x * (blockIdx.y + nqb * blockIdx.z);
            const int ppx = P >> 3, slot = L >> 3;
            const int pair = (L & 7) * ppx + slot % ppx;
            b = slot / ppx; h = pair / nqb; qb = pair % nqb;
        }
    }
    const int q0 = qb * (32 * NW) + wave * 32;
    const int query = q0 + l31;
    const bool wave_active = q0 < p.nq;
    // lanes with a real query: the lanes of the padding rows of a ragged last wave hold whatever the bias buffer's padding holds and
    // must not take part in the wave-wide decision to move the running maximum (the result would depend on that padding)
    const unsigned long long qlanes = __builtin_amdgcn_ballot_w64(query < p.nq);
#ifdef PD_LAB
    unsigned long long* ptr_ = nullptr;
    {
        const int lid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (g_pipe_trace && lane == 0 && lid < 1024) ptr_ = g_pipe_trace + ((long long)lid * NW + wave) * 16;
    }
#endif
    PD_PSTAMP(0);                                          // kernel entry

    // power-of-two operand scales from the magnitude bounds (as attn_f16.hip); S' = S / c_s is what the matrix pipe accumulates
    float qs = p.scale * PD_LOG2E;
    const float sq = pd_pow2_scale(p.f16_amax ? p.f16_amax[0] * qs : p.f16_q_amax * qs);
    const float sk = pd_pow2_scale(p.f16_amax ? p.f16_amax[1] : p.f16_k_amax);
    const float sv = pd_pow2_scale(p.f16_amax ? p.f16_amax[2] : p.f16_v_amax);
    qs *= sq;
    const float c_s = 1.0f / (sq * sk);                    // exact: powers of two
    const float inv_sv = 1.0f / sv;

    // Q fragments (filled in the prologue): k-step s covers dims 16 s + 8 hh .. + 8 of the lane's query; [s][0] high, [s][1] low parts
    frag qf[2][2];
    const int nsub = (p.nk + 31) >> 5;                     // 32-key sub-tiles with at least one real key
    const int nit = (p.nk + KT - 1) / KT;
    const int nkt32 = ((p.bias_nk > 0 ? p.bias_nk : p.nk) + 31) >> 5;
    const int nqt32 = (p.nq + 31) >> 5;
    // Every request of the main loop is a buffer load: descriptor in SGPRs, ONE loop-invariant VGPR offset per stream, the tile /
    // sub-tile offset as a scalar - no 64-bit per-lane address arithmetic, and rows beyond the last key (or bias tiles beyond the
    // last sub-tile) read as zero through the descriptor's range check instead of through per-lane conditionals.
    const float* bias_base = p.bias;
    if constexpr (HASBIAS) bias_base += (((long long)h * nqt32 + ((wave_active ? q0 : 0) >> 5)) * nkt32) * 1024;
    const auto rs_bias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias_base), 0, HASBIAS ? nkt32 * 4096 : 0, 0x00020000);
    const int boff = lane * 16;

    // ---- staging: thread -> key row (tid >> 3) + RPP i, 4 dims at 4 (tid & 7)
    constexpr int RPP = 8 * NW, NST = KT / RPP;
    const int srow = tid >> 3, sc = tid & 7;
    f32x4 rk[NST], rv[NST];
    // PRE: rows of pd_gemm_args.Y2 (fp16 elements; 16 bytes = both parts of four dims); else fp32 rows
    const long long kss = PRE ? p.kv2_ss * 2 : p.k_ss * 4, vss = PRE ? p.kv2_ss * 2 : p.v_ss * 4;      // bytes per key
    const char* kbase = PRE ? reinterpret_cast<const char*>(p.K2) + ((long long)b * p.kv2_bs + h * 64) * 2
                            : reinterpret_cast<const char*>(p.K + (long long)b * p.k_bs + h * 32);
    const char* vbase = PRE ? reinterpret_cast<const char*>(p.V2) + ((long long)b * p.kv2_bs + h * 64) * 2
                            : reinterpret_cast<const char*>(p.V + (long long)b * p.v_bs + h * 32);
    const auto rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(kbase), 0, (int)((p.nk - 1) * kss + 128), 0x00020000);
    const auto rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vbase), 0, (int)((p.nk - 1) * vss + 128), 0x00020000);
    int koff_g[NST], voff_g[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        koff_g[i] = (int)((srow + RPP * i) * kss) + 16 * sc;
        voff_g[i] = (int)((srow + RPP * i) * vss) + 16 * sc;
    }
    const int ktile_b = (int)(KT * kss), vtile_b = (int)(KT * vss);      // bytes per 64-key tile
    auto gload_to = [&](f32x4 (&rk)[NST], f32x4 (&rv)[NST], int tile) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            rk[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_k, koff_g[i], tile * ktile_b, 0));
            rv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_v, voff_g[i], tile * vtile_b, 0));
        }
    };
    auto gload = [&](int tile) { gload_to(rk, rv, tile); };
    auto sstore_from = [&](const f32x4 (&rk)[NST], const f32x4 (&rv)[NST], int stage_off) {
        unsigned short* sK = lds + stage_off;
        unsigned short* sV = sK + 2 * K_PART;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int kr = srow + RPP * i;                         // key row inside the tile
            unsigned k0[2], k1[2], v0[2], v1[2];                   // packed pairs (e0, e1), (e2, e3): [0] high, [1] low parts
            if constexpr (PRE) {
                const u32x4 kb = __builtin_bit_cast(u32x4, rk[i]), vb = __builtin_bit_cast(u32x4, rv[i]);
                k0[0] = kb[0]; k1[0] = kb[1]; k0[1] = kb[2]; k1[1] = kb[3];
                v0[0] = vb[0]; v1[0] = vb[1]; v0[1] = vb[2]; v1[1] = vb[3];
            } else {
                pd_parts2 t;
                t = pd_split2h(rk[i][0] * sk, rk[i][1] * sk); k0[0] = t.h; k0[1] = t.l;
                t = pd_split2h(rk[i][2] * sk, rk[i][3] * sk); k1[0] = t.h; k1[1] = t.l;
                t = pd_split2h(rv[i][0] * sv, rv[i][1] * sv); v0[0] = t.h; v0[1] = t.l;
                t = pd_split2h(rv[i][2] * sv, rv[i][3] * sv); v1[0] = t.h; v1[1] = t.l;
            }
            const int ko = kr * KP + 4 * sc;
#pragma unroll
            for (int k = 0; k < 2; ++k) *reinterpret_cast<u32x2*>(sK + k * K_PART + ko) = u32x2{k0[k], k1[k]};
            const int vo = (kr & 32) + vpos(kr & 31);              // column of this key in the transposed tile
#pragma unroll
            for (int e = 0; e < 4; ++e) {                          // transposed scatter: dim 4 sc + e, column vo
                const int ro = (4 * sc + e) * VP + vo;
                const int sh = 16 * (e & 1);
#pragma unroll
                for (int k = 0; k < 2; ++k) sV[k * V_PART + ro] = (unsigned short)((e < 2 ? v0[k] : v1[k]) >> sh);
            }
        }
    };
    auto sstore = [&](int stage_off) { sstore_from(rk, rv, stage_off); };

    // ---- fragment addresses: lane offsets are loop-invariant, stage / sub-tile offsets are scalars
    const int koff = l31 * KP + 8 * hh;                    // K fragment (A operand: 8 dims of key l31), + part * K_PART + 16 st
    const int voff = 2 * K_PART + l31 * VP + 8 * hh;       // V^T fragment (A operand: 8 keys of dim l31), + part * V_PART + 16 st
    auto kfrag = [&](int base, int st, int pt) { return *reinterpret_cast<const frag*>(lds + base + koff + pt * K_PART + 16 * st); };
    auto vfrag = [&](int base, int st, int pt) { return *reinterpret_cast<const frag*>(lds + base + voff + pt * V_PART + 16 * st); };

    auto load_bias = [&](f32x16& s, int kt32) {            // score accumulator <- bias tile (x bias_prescale)
        if constexpr (HASBIAS) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_bias, boff, kt32 * 4096 + g * 1024, 0));
                s[4 * g] = v[0]; s[4 * g + 1] = v[1]; s[4 * g + 2] = v[2]; s[4 * g + 3] = v[3];
            }
        }
    };
    auto first = [&](const f32x16& s) {                    // C operand of a score tile's first MFMA
        if constexpr (HASBIAS) return s;
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        return z;
    };

    f32x16 o, sA, sB;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;                  // m_run in units of S' (= s / c_s)
    float mneg_run = 0.f;                                  // 14 - LAZY - m_run c_s: the additive constant in front of the exp2
    const float tau_s = (float)LAZY / c_s;                 // the lazy-maximum threshold in units of S'

    auto rowmax = [&](const f32x16& s) {
        float m = max3(s[0], s[1], s[2]);
        m = max3(m, s[3], s[4]); m = max3(m, s[5], s[6]); m = max3(m, s[7], s[8]); m = max3(m, s[9], s[10]);
        m = max3(m, s[11], s[12]); m = max3(m, s[13], s[14]);
        m = __builtin_fmaxf(m, s[15]);
        return pd_xhalf_max(m);
    };
    // p = 2^14 exp2(s - m) of four elements, in place; the factor 2^14 keeps the low fp16 part normal down to p = 2^-16
    auto fe4 = [&](f32x16& s, int r0, float mneg) {
#pragma unroll
        for (int r = r0; r < r0 + 4; ++r) s[r] = (ABL & 2) ? __builtin_fmaf(s[r], c_s, mneg) : __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c_s, mneg));
    };
    auto split8 = [&](const f32x16& s, int r0, frag (&pf)[2]) {       // eight probabilities -> (hi, lo) B fragments of one k-step
        u32x4 fh, fl;
        if constexpr (ABL & 8) {
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) { fh[e2] = __float_as_uint(s[r0 + 2 * e2]); fl[e2] = __float_as_uint(s[r0 + 2 * e2 + 1]); }
            pf[0] = __builtin_bit_cast(frag, fh); pf[1] = __builtin_bit_cast(frag, fl);
            return;
        }
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
            const pd_parts2 t = pd_split2h(s[r0 + 2 * e2], s[r0 + 2 * e2 + 1]);
            fh[e2] = t.h; fl[e2] = t.l;
        }
        pf[0] = __builtin_bit_cast(frag, fh);
        pf[1] = __builtin_bit_cast(frag, fl);
    };

    // ---- one pipelined phase: softmax + P.V of sub-tile j (scores in `cur`, their row maximum in `mloc`) interleaved with the
    // score MFMAs of sub-tile j+1 (`nxt`, preloaded with its bias tile).  kn: LDS base (stage + sub-tile row offset) of the K
    // rows of sub-tile j+1, vc: LDS base (stage + sub-tile column offset) of the V^T columns of sub-tile j, kn2: K rows of
    // sub-tile j+2 (k-step 0 fragments requested at the end, handed over in kf0), bt: bias tile to fetch into `cur`.
    auto phase = [&](f32x16& cur, f32x16& nxt, float& mloc, frag (&kf0)[2], int kn, int vc, int kn2, int bt) {
        frag kf1[2], vf0[2], vf1[2], pf0[2], pf1[2];
        // slot 1
        nxt = pmma(kf0[0], qf[0][1], first(nxt));                       // k_hi . q_lo (C = the bias tile, or zero)
        PD_SB();
        kf1[0] = kfrag(kn, 1, 0); kf1[1] = kfrag(kn, 1, 1);
        float alpha = 1.0f;
        if constexpr (LAZY > 0) {
            if ((__builtin_amdgcn_ballot_w64(mloc > m_run + tau_s) & qlanes) != 0ull) {      // rare after the first sub-tiles
                const float m_new = __builtin_fmaxf(m_run, mloc);
                alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c_s);
                m_run = m_new;
                mneg_run = __builtin_fmaf(-m_new, c_s, PSH);
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] *= alpha;
                l_run *= alpha;
            }
        } else {
            const float m_new = __builtin_fmaxf(m_run, mloc);
            alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c_s);
            m_run = m_new;
            mneg_run = __builtin_fmaf(-m_new, c_s, PSH);
        }
        const float mneg = mneg_run;
        if constexpr (!(ABL & 8)) fe4(cur, 0, mneg);
        PD_SB();
        // slot 2
        nxt = pmma(kf0[1], qf[0][0], nxt);                              // k_lo . q_hi
        PD_SB();
        if constexpr (!(ABL & 8)) fe4(cur, 4, mneg);
        float ps0 = (cur[0] + cur[1]) + (cur[2] + cur[3]);
        PD_SB();
        // slot 3
        nxt = pmma(kf0[0], qf[0][0], nxt);                              // k_hi . q_hi
        PD_SB();
        vf0[0] = vfrag(vc, 0, 0); vf0[1] = vfrag(vc, 0, 1);
        split8(cur, 0, pf0);
        PD_SB();
        // slot 4
        nxt = pmma(kf1[0], qf[1][1], nxt);
        PD_SB();
        if constexpr (!(ABL & 8)) { fe4(cur, 8, mneg); fe4(cur, 12, mneg); }
        PD_SB();
        // slot 5
        nxt = pmma(kf1[1], qf[1][0], nxt);
        PD_SB();
#pragma unroll
        for (int r = 0; r < 8; ++r) if constexpr (!(ABL & 8) && LAZY == 0) o[r] *= alpha;
        float ps1 = (cur[4] + cur[5]) + (cur[6] + cur[7]);
        ps0 += (cur[8] + cur[9]) + (cur[10] + cur[11]);
        PD_SB();
        // slot 6
        nxt = pmma(kf1[0], qf[1][0], nxt);
        PD_SB();
#pragma unroll
        for (int r = 8; r < 16; ++r) if constexpr (!(ABL & 8) && LAZY == 0) o[r] *= alpha;
        ps1 += (cur[12] + cur[13]) + (cur[14] + cur[15]);
        PD_SB();
        // slot 7: the last use of `cur` - the bias tile of sub-tile j+2 is fetched into it right behind
        o = pmma(vf0[0], pf0[1], o);                                    // v_hi . p_lo
        PD_SB();
        vf1[0] = vfrag(vc, 1, 0); vf1[1] = vfrag(vc, 1, 1);
        split8(cur, 8, pf1);
        PD_SB();
        if constexpr (!(ABL & 32)) load_bias(cur, bt);
        PD_SB();
        // slot 8
        o = pmma(vf0[1], pf0[0], o);                                    // v_lo . p_hi
        PD_SB();
        if constexpr (LAZY > 0) l_run += ps0 + ps1; else l_run = __builtin_fmaf(l_run, alpha, ps0 + ps1);
        PD_SB();
        // slot 9: K fragments of sub-tile j+2's first k-step
        o = pmma(vf0[0], pf0[0], o);                                    // v_hi . p_hi
        PD_SB();
        kf0[0] = kfrag(kn2, 0, 0); kf0[1] = kfrag(kn2, 0, 1);
        PD_SB();
        // slots 10 - 12: row maximum of the NEXT sub-tile's scores under the last three P.V MFMAs
        o = pmma(vf1[0], pf1[1], o);
        PD_SB();
        float m0 = max3(nxt[0], nxt[1], nxt[2]);
        m0 = max3(m0, nxt[3], nxt[4]); m0 = max3(m0, nxt[5], nxt[6]); m0 = max3(m0, nxt[7], nxt[8]);
        PD_SB();
        o = pmma(vf1[1], pf1[0], o);
        PD_SB();
        m0 = max3(m0, nxt[9], nxt[10]); m0 = max3(m0, nxt[11], nxt[12]); m0 = max3(m0, nxt[13], nxt[14]);
        m0 = __builtin_fmaxf(m0, nxt[15]);
        PD_SB();
        o = pmma(vf1[0], pf1[0], o);
        PD_SB();
        mloc = pd_xhalf_max(m0);
        PD_SB();
    };

    // score MFMAs of one sub-tile, not interleaved (prologue / last tile)
    auto scores = [&](f32x16& s, int kn) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const frag kh = kfrag(kn, st, 0), kl = kfrag(kn, st, 1);
            s = mma(kh, qf[st][1], st == 0 ? first(s) : s);
            s = mma(kl, qf[st][0], s);
            s = mma(kh, qf[st][0], s);
        }
    };
    // softmax + P.V of one sub-tile, not interleaved (last tile); ragged: keys >= nk are masked
    auto finish = [&](f32x16& s, int kt32, int vc) {
        if ((kt32 + 1) * 32 > p.nk) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kt32 * 32 + pd_frag_row(r, hh) >= p.nk) s[r] = -INFINITY;
        }
        const float mloc = rowmax(s);
        const float m_new = __builtin_fmaxf(m_run, mloc);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c_s);
        m_run = m_new;
        const float mneg = __builtin_fmaf(-m_new, c_s, PSH);
        mneg_run = mneg;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= alpha;
        float ps = 0.f;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            fe4(s, 8 * st, mneg);
            fe4(s, 8 * st + 4, mneg);
#pragma unroll
            for (int r = 8 * st; r < 8 * st + 8; ++r) ps += s[r];
            frag pf[2];
            split8(s, 8 * st, pf);
            const frag vh = vfrag(vc, st, 0), vl = vfrag(vc, st, 1);
            o = mma(vh, pf[1], o);
            o = mma(vl, pf[0], o);
            o = mma(vh, pf[0], o);
        }
        l_run = l_run * alpha + ps;
    };

    // ---- prologue: every independent request first (K / V tile 0, bias tiles 0 / 1 into the two score accumulators, the lane's
    // query row) - their latencies overlap instead of adding up, which is most of a short launch (256 keys: 4 tiles per block)
    f32x4 rkr[RES ? RES_TILES - 1 : 1][NST], rvr[RES ? RES_TILES - 1 : 1][NST];      // RES: tiles 1 .. 3 (tile 0 in rk / rv)
    gload(0);
    if constexpr (RES) {
#pragma unroll
        for (int t = 1; t < RES_TILES; ++t) gload_to(rkr[t - 1], rvr[t - 1], t);      // (tiles beyond the last key read as zero)
    }
    load_bias(sA, 0);
    load_bias(sB, 1);
    f32x4 qraw[2][2];
    {
        // rows beyond nq read as zero through the descriptor's range check
        const auto rs_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Q + (long long)b * p.q_bs + h * 32), 0,
                                                            (int)(((long long)p.nq - 1) * p.q_ss * 4 + 128), 0x00020000);
        const int qoff = (int)((long long)query * p.q_ss * 4) + 32 * hh;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            qraw[s][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_q, qoff, 64 * s, 0));
            qraw[s][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_q, qoff, 64 * s + 16, 0));
        }
    }
    PD_PSTAMP(1);                                          // every prologue request issued
    PD_SB();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const f32x4 v0 = qraw[s][0], v1 = qraw[s][1];
        u32x4 fh, fl;
        pd_parts2 t;
        t = pd_split2h(v0[0] * qs, v0[1] * qs); fh[0] = t.h; fl[0] = t.l;
        t = pd_split2h(v0[2] * qs, v0[3] * qs); fh[1] = t.h; fl[1] = t.l;
        t = pd_split2h(v1[0] * qs, v1[1] * qs); fh[2] = t.h; fl[2] = t.l;
        t = pd_split2h(v1[2] * qs, v1[3] * qs); fh[3] = t.h; fl[3] = t.l;
        qf[s][0] = __builtin_bit_cast(frag, fh);
        qf[s][1] = __builtin_bit_cast(frag, fl);
    }
    PD_PSTAMP(2);                                          // Q arrived and split
    sstore(0);
    PD_PSTAMP(3);                                          // K / V tile 0 arrived and staged
    if constexpr (RES) {
#pragma unroll
        for (int t = 1; t < RES_TILES; ++t) sstore_from(rkr[t - 1], rvr[t - 1], t * STAGE);
    } else {
        gload(1);
    }
    lds_barrier();
    PD_PSTAMP(4);                                          // first barrier passed
    int s_cur = 0, s_nxt = STAGE, s_nn = 2 * STAGE;
    float mloc = 0.f;
    frag kf0[2];
    if (wave_active) {
        scores(sA, s_cur);
        kf0[0] = kfrag(s_cur + 32 * KP, 0, 0); kf0[1] = kfrag(s_cur + 32 * KP, 0, 1);
        mloc = rowmax(sA);
    }

    PD_PSTAMP(5);                                          // first score tile + row maximum
    if constexpr (ABL & 16) { if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1); }
    if (wave_active) {
        for (int it = 0; it < nit - 1; ++it) {
            if constexpr (!RES) {
                if constexpr (!(ABL & 64)) sstore(s_nxt);      // tile it + 1 (requested one iteration ago)
                if constexpr (!(ABL & 1)) lds_barrier();
                if constexpr (!(ABL & 64)) gload(it + 2);      // tile it + 2 (rows beyond the last key read as zero)
            }
            PD_SB();
            // sub-tile 2 it (cur = sA): next scores = sub-tile 2 it + 1 (same tile, second half)
            phase(sA, sB, mloc, kf0, s_cur + 32 * KP, s_cur, s_nxt, 2 * it + 2);
            // sub-tile 2 it + 1 (cur = sB): next scores = first half of tile it + 1
            phase(sB, sA, mloc, kf0, s_nxt, s_cur + 32, s_nxt + 32 * KP, 2 * it + 3);
            if constexpr (RES) { s_cur = s_nxt; s_nxt += STAGE; }          // resident tiles sit at it x STAGE
            else { const int t = s_cur; s_cur = s_nxt; s_nxt = s_nn; s_nn = t; }
#ifdef PD_LAB
            if (it < 6) PD_PSTAMP(6 + it);                 // end of main-loop iteration it
#endif
        }
    } else if constexpr (!RES) {                           // a wave without queries (ragged last block) only stages
        for (int it = 0; it < nit - 1; ++it) {
            sstore(s_nxt);
            lds_barrier();
            gload(it + 2);
            const int t = s_cur; s_cur = s_nxt; s_nxt = s_nn; s_nn = t;
        }
    }

    // ---- last tile: scores of its first sub-tile are in sA (issued by the last phase / the prologue)
    if (wave_active) {
        const int kt0 = 2 * (nit - 1);
        const bool two = (kt0 + 1) * 32 < p.nk;
        if (two) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const frag kh = st == 0 ? kf0[0] : kfrag(s_cur + 32 * KP, 1, 0), kl = st == 0 ? kf0[1] : kfrag(s_cur + 32 * KP, 1, 1);
                sB = mma(kh, qf[st][1], st == 0 ? first(sB) : sB);
                sB = mma(kl, qf[st][0], sB);
                sB = mma(kh, qf[st][0], sB);
            }
        }
        finish(sA, kt0, s_cur);
        if (two) finish(sB, kt0 + 1, s_cur + 32);
    }

    PD_PSTAMP(12);                                         // last tile finished
    if (query < p.nq) {
        const float l = pd_xhalf_sum(l_run);
        if (p.O2) {
            // output already split for the projection that follows (pd_gemm_args.A2): o times the V scale (|o| <= max|v|)
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
            const float inv = inv_sv / l;
            float* op = p.O + (long long)b * p.o_bs + (long long)query * p.o_ss + h * 32 + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv};
                *reinterpret_cast<f32x4*>(op + 8 * g) = v;
            }
        }
    }
    PD_PSTAMP(13);                                         // output stores issued
}

constexpr int LDS_BYTES = NSTAGE * STAGE * 2;
constexpr int LDS_BYTES_RES = RES_TILES * STAGE * 2;

template <int NW, bool PRE, bool HASBIAS>
bool raise_lds() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(attn_pipe_kernel<NW, PRE, HASBIAS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               LDS_BYTES) == hipSuccess &&
           (NW != 8 || hipFuncSetAttribute(reinterpret_cast<const void*>(attn_pipe_kernel<8, PRE, HASBIAS, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_RES) == hipSuccess);
}

// lab: 1 = launches of at most 256 keys take the resident form.  Measured and OFF (profiles/r06_ab_pipe_resident.txt): token DiT attention
// 44.4 vs 44.5 us, triangle shape 42.4 vs 43.9 us, 64-sample call 402.4 / 401.9 vs 401.7 / 402.7 ms - the per-tile barrier and staging
// are not what a four-tile launch waits for
#ifndef PD_PIPE_RES
#define PD_PIPE_RES 0
#endif

// Short key ranges (<= PD_PIPE_NW4_MAXNK keys: token / triangle / MSA attention at 256 - 512 tokens) run on FOUR-wave blocks of 128
// queries: a block lives for only four to eight key tiles, most of it prologue and tail, and twice as many independent blocks per CU
// (four resident instead of two) fill each other's gaps; the K / V tiles are staged by both query blocks of a (sample, head).
#ifndef PD_PIPE_NW4_MAXNK
#define PD_PIPE_NW4_MAXNK 0
#endif
template <bool PRE, bool HASBIAS>
void launch(const pd_attn_args* a, hipStream_t stream) {
    if (a->nq > 128 && a->nk <= PD_PIPE_NW4_MAXNK) {
        dim3 grid(a->nbatch, (a->nq + 127) / 128, a->nheads);
        hipLaunchKernelGGL((attn_pipe_kernel<4, PRE, HASBIAS>), grid, dim3(256), LDS_BYTES, stream, *a);
    } else if (a->nq > 128 && PD_PIPE_RES && a->nk <= RES_TILES * KT && a->nk > KT) {
        dim3 grid(a->nbatch, (a->nq + 255) / 256, a->nheads);       // every key tile resident: one barrier, no staging in the loop
        hipLaunchKernelGGL((attn_pipe_kernel<8, PRE, HASBIAS, true>), grid, dim3(512), LDS_BYTES_RES, stream, *a);
    } else if (a->nq > 128) {
        dim3 grid(a->nbatch, (a->nq + 255) / 256, a->nheads);
        hipLaunchKernelGGL((attn_pipe_kernel<8, PRE, HASBIAS>), grid, dim3(512), LDS_BYTES, stream, *a);
    } else {
        dim3 grid(a->nbatch, 1, a->nheads);
        hipLaunchKernelGGL((attn_pipe_kernel<4, PRE, HASBIAS>), grid, dim3(256), LDS_BYTES, stream, *a);
    }
}

}  // namespace

// log2 of the power of two a bias producer folds into its out_scale (x log2 e) for launches of this kernel: the product of the
// q and k operand scales the kernel derives from the same bounds - host arithmetic identical to the device's (IEEE fp32 products).
PD_EXPORT int pd_attention_bias_prescale_log2(float q_amax, float k_amax, float scale) {
    auto exp2_of_scale = [](float amax) {                // exponent of pd_pow2_scale(amax)
        unsigned u;
        memcpy(&u, &amax, 4);
        int e = (int)((u >> 23) & 0xff);
        e = e < 87 ? 87 : (e > 200 ? 200 : e);
        return (268 - e) - 127;
    };
    volatile float qs = scale * PD_LOG2E;
    volatile float qa = q_amax * qs;
    return exp2_of_scale(qa) + exp2_of_scale(k_amax);
}

#ifdef PD_LAB
extern "C" __attribute__((visibility("default"))) int pd_lab_set_pipe_trace(void* buf) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_pipe_trace), &q, sizeof(q)) == hipSuccess ? PD_OK : PD_ERR_LAUNCH;
}
#endif

// can this launch take the pipelined kernel?  (fp16-format unsplit launches whose bias - if any - was produced pre-scaled)
extern "C" int pd_attention_pipe_ok(const pd_attn_args* a) {
    if (!a->f16x3 || a->fp32_mfma || a->bias_prescale < 0.f) return 0;      // < 0: the caller asks for attn_parts_kernel (A/B runs)
    if (a->bias && !(a->bias_prescale > 0.f)) return 0;
    // the masked entries of a pre-scaled tile are -1e9 log2(e) 2^k (~ -2^(30.4 + k)): beyond k = 90 they would overflow to -inf and a
    // fully masked query row would compute (-inf) - (-inf).  Tiny q / k bounds (near-zero projections) stay on attn_parts_kernel.
    if (a->bias && a->bias_prescale > 0x1p90f) return 0;
    // the kernel addresses q, k, v and the bias through buffer descriptors with 32-bit offsets relative to the (batch, head) base
    const long long lim = 0xffffff00ll;
    const long long kss = a->K2 ? a->kv2_ss * 2 : a->k_ss * 4, vss = a->K2 ? a->kv2_ss * 2 : a->v_ss * 4;
    const long long nkt32 = ((a->bias_nk > 0 ? a->bias_nk : a->nk) + 31) >> 5;
    if ((long long)a->nq * a->q_ss * 4 >= lim || (long long)(a->nk + 64) * kss >= lim || (long long)(a->nk + 64) * vss >= lim || nkt32 * 4096 >= lim)
        return 0;
    return 1;
}

// init_only: 1 raise the dynamic-LDS limits; 0 launch
extern "C" int pd_attention_pipe_try(const pd_attn_args* a, void* stream, int init_only) {
    if (init_only == 1)
        return raise_lds<8, false, false>() && raise_lds<8, false, true>() && raise_lds<8, true, false>() && raise_lds<8, true, true>() &&
                       raise_lds<4, false, false>() && raise_lds<4, false, true>() && raise_lds<4, true, false>() && raise_lds<4, true, true>()
                   ? PD_OK : PD_ERR_LAUNCH;
    if (!a->f16_amax && !(a->f16_q_amax > 0.f && a->f16_k_amax > 0.f && a->f16_v_amax > 0.f)) return PD_ERR_ARG;
    if (a->O2 && (((uintptr_t)a->O2 & 15) || a->o_ss != (long long)a->nheads * 32 || a->o_bs != (long long)a->nq * a->o_ss))
        return PD_ERR_ARG;
    if (a->K2 && (!a->V2 || (((uintptr_t)a->K2 | (uintptr_t)a->V2) & 15) || a->kv2_ss % 8 || a->kv2_bs % 8)) return PD_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (a->K2) { if (a->bias) launch<true, true>(a, s); else launch<true, false>(a, s); }
    else       { if (a->bias) launch<false, true>(a, s); else launch<false, false>(a, s); }
    return pd_check_launch();
}
