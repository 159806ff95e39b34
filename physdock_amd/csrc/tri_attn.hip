// Triangle attention with the q | k | v projection INSIDE the attention block (round 6; reference primitives/attentions.py:194-217).
//
// The chain  RMSNorm(z) -> q | k | v projection (pd_gemm: 44 us, 168 MB written and read back) -> attention (attn_pipe_kernel: 35 - 43 us,
// a four-tile launch that is mostly prologue: MfmaUtil 0.23)  becomes one launch.  A block owns one (pair row i, head h): 256 query rows
// = the 256 key rows of that row of z.  Every wave
//   1. reads ITS 32 rows of z[i] / rms(z[i]) as two-part fp16 FRAGMENTS: pd_pair_bias_split - the streaming pass over z that produces the
//      bias tiles and knows the row statistics - writes them scaled, split and in fragment order (lane = row, eight channels per
//      k-step), so a request is one coalesced kilobyte per wave and needs no conversion; the norm gain is folded into the weights;
//   2. contracts them with the head's 96 weight rows (q, k, v: 32 each; fragment-major two-part fp16, packing.split2_f16, straight from
//      L2): 72 MFMAs, three accumulator tiles;
//   3. turns the q tile into its own Q fragments (v_permlane32_swap: no LDS round trip), writes its 32 keys' K rows and V^T columns
//      into the block's LDS tiles (the layouts of attn_pipe.hip);
// and after ONE block barrier all 256 keys of the (row, head) are resident: the software-pipelined wave program of attn_pipe.hip (score
// MFMAs of sub-tile j + 1 under the softmax of sub-tile j, bias tile as the accumulator's initial value, lazy running maximum) runs
// over them without staging, without global K / V requests and without another barrier.
// q | k | v never exist in HBM.  T <= 256 (four 64-key tiles, 78 KB of LDS: two blocks per CU); C = 128.
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "physdock_hip.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef f16x8 frag;

constexpr int CZ = 128, NKS = CZ / 16;          // channels, k-steps of the projection
constexpr int KT = 64, KP = 40, VP = 72;        // attn_pipe.hip's tile layouts
constexpr int K_PART = KT * KP, V_PART = 32 * VP;
constexpr int STAGE = 2 * (K_PART + V_PART);
constexpr int NTILE = 4;                        // 256 keys
// PD_TRI_WLDS: where the head's 96 weight rows (48 KB of two-part fp16 fragments) come from during the projection.
//   0 (form 2): every wave requests its weight fragments - and the low parts of its rows of z once per output tile - from L2: 0.9 MB of
//     requests per block on the CU's vector-memory path, which is what the projection phase waits for (profiles/r06_tri_attn_forms.txt).
//   1 (form 4): the weights are staged ONCE per block in LDS - in the space of the K / V tiles, which are not written before every wave
//     has left the projection (the projected k / v tiles wait, packed, in 32 registers; one more block barrier) - so LDS stays at 78 KB =
//     two blocks per CU, and the projection runs k-step-major over THREE accumulators (q, k, v), which streams both parts of the wave's
//     rows of z through a three-deep ring exactly once: 0.43 MB of requests per block.  Every accumulator sees the same MFMA sequence as
//     in form 0: results bit-identical.  (Form 3 of round 6 - weights in LDS BEHIND the K / V tiles, 126 KB = one block per CU - was slower
//     than form 2: the attention phase at two waves per SIMD takes 45 us instead of 32.)
#ifndef PD_TRI_WLDS
#define PD_TRI_WLDS 1
#endif
constexpr bool WLDS = PD_TRI_WLDS != 0;
#ifndef PD_TRI_ZD
#define PD_TRI_ZD 2
#endif
constexpr int ZD = PD_TRI_ZD;                    // form 4: k-steps of the wave's rows of z in flight ahead of the MFMAs
constexpr int W_HALVES = 3 * 2 * NKS * 64 * 8;              // q, k, v tiles x 2 parts x 8 k-steps x 64 lanes x 8 halves = 48 KB
static_assert(W_HALVES <= NTILE * STAGE, "the staged weights live in the K / V tiles' space");
// PD_TRI_ROWS2 = 1 (lab): a block of SIXTEEN waves owns TWO pair rows of one head (waves 0 - 7 row 2 ip, waves 8 - 15 row 2 ip + 1): one
// weight stage per two rows (in the first row's K / V space), and the two rows' waves walk the same bias tiles at the same time.  156 KB
// of LDS: one block per CU, the same four waves per SIMD.
#ifndef PD_TRI_ROWS2
#define PD_TRI_ROWS2 0
#endif
constexpr int RPB = (PD_TRI_ROWS2 != 0 && WLDS) ? 2 : 1;    // pair rows per block
constexpr int NTHR = 512 * RPB;
constexpr int LDS_BYTES = RPB * NTILE * STAGE * 2;
constexpr int LAZY = 3;
constexpr float PSH = 14.0f - (float)LAZY;

#define PD_SB() __builtin_amdgcn_sched_barrier(0)

// lab ablations (timing only, wrong results; tools/abl_tri_attn.sh): 1 no attention phase, 2 weight fragments not loaded, 4 no
// projection MFMAs, 8 no bias fetches, 16 rows of z not loaded, 32 no output stores
#ifdef PD_TRI_ABL
constexpr int ABL = PD_TRI_ABL;
#else
constexpr int ABL = 0;
#endif

__device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
// MFMA of the projection (ablation 4 removes it)
__device__ __forceinline__ f32x16 jmma(frag a, frag b, f32x16 c) {
#ifdef PD_TRI_ABL
    if (PD_TRI_ABL & 4) {
        c[0] += __builtin_bit_cast(f32x4, a)[0] * 0.f + __builtin_bit_cast(f32x4, b)[0] * 0.f;
        return c;
    }
#endif
    return mma(a, b, c);
}
__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

__device__ __forceinline__ void lds_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0); outstanding global loads (bias tiles) stay in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(NTHR, RPB == 2 ? 1 : 4) void tri_attn_kernel(const pd_tri_attn_args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds_all[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_b = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave of the block
    const int wave = wave_b & 7, grp = wave_b >> 3;                      // wave of its pair row, pair row of the block
    unsigned short* const lds = lds_all + grp * (NTILE * STAGE);         // the row's K / V tiles (the weights: lds_all, i.e. row 0's)
    const int l31 = lane & 31, hh = lane >> 5;
    int i = blockIdx.x, h = blockIdx.y;
#ifndef PD_TRI_XCD
#define PD_TRI_XCD 1
#endif
    if (PD_TRI_XCD && (gridDim.x & 7) == 0) {
        // Workgroups go to the eight XCDs round-robin in dispatch order.  With the plain (row, head) grid the four heads of a pair row are
        // 256 dispatches apart: same XCD, but two of them a whole round later - the row's 128 KB of split z come from HBM twice.  Here the
        // four heads of a row are consecutive dispatches of ONE XCD: one fetch per row, the other three blocks hit that XCD's L2.
        const int L = blockIdx.x + gridDim.x * blockIdx.y;
        const int slot = L >> 3;
        i = 8 * (slot >> 2) + (L & 7);
        h = slot & 3;
    }
    i = RPB * i + grp;                                                  // (RPB = 2: the grid's x counts row PAIRS)
#ifdef PD_TRI_SKEW
    // lab: the two blocks resident on a CU are the linear ids L and L + 256 (round-robin over 8 XCDs x 32 CUs): delay every second group of
    // 256 by PD_TRI_SKEW x 8 128 cycles so that one block projects (latency-bound) while the other attends (issue-bound)
    if (((blockIdx.x + gridDim.x * blockIdx.y) >> 8) & 1) {
#pragma unroll
        for (int k = 0; k < PD_TRI_SKEW; ++k) __builtin_amdgcn_s_sleep(127);
    }
#endif
    const int T = p.T, nk = p.Treal;
    const long long bs = p.transpose ? CZ : (long long)T * CZ;          // floats between the batch rows i
    const long long ss = p.transpose ? (long long)T * CZ : CZ;          // floats between the sequence rows of one batch row
    const int q0 = wave * 32;
    const int row = q0 + l31;                                           // the lane's query = key row
    const bool wave_active = q0 < T;
    const unsigned long long qlanes = __builtin_amdgcn_ballot_w64(row < T);

    // ---- operand scales (powers of two): zn = z / rms is bounded by sqrt(C); q, k, v by the static bounds of the projection
    float qs = p.scale * PD_LOG2E;
    const float sq = pd_pow2_scale(p.qkv_amax[0] * qs);
    const float sk = pd_pow2_scale(p.qkv_amax[1]);
    const float sv = pd_pow2_scale(p.qkv_amax[2]);
    qs *= sq;
    const float c_s = 1.0f / (sq * sk);
    const float inv_sv = 1.0f / sv;
    const float a_s = pd_pow2_scale(p.zn_amax);
    const float inv_as = 1.0f / a_s;

    // ---- 1. the wave's 32 rows of z[i] / rms, already scaled and split, in fragment order (written by pd_pair_bias_split)
    const int ntile = (T + 31) >> 5;
    // (a wave without rows - T < 256 - reads the rows of wave 0: in bounds, projected, stored into tiles no query reads as keys < nk)
    const frag* zbase = reinterpret_cast<const frag*>(p.z2) + ((long long)i * ntile + (wave_active ? wave : 0)) * (NKS * 2 * 64) + lane;
    auto zfrag_g = [&](int s, int part) {
        if constexpr (ABL & 16) return __builtin_bit_cast(frag, part ? u32x4{0x1c001c00u, 0x1c009c00u, 0x18001c00u, 0x1c001400u}
                                                                      : u32x4{0x3c003c00u, 0x3c00bc00u, 0x38003c00u, 0x3c003400u});
        return zbase[(2 * s + part) * 64];
    };
    frag qf[2][2];                                                      // the wave's Q fragments (as attn_pipe.hip)
    const int tile = wave >> 1, half = wave & 1;                        // the wave's keys: LDS tile and its 32-key half
    unsigned short* sK = lds + tile * STAGE;
    unsigned short* sV = sK + 2 * K_PART;
    const float fq = qs * inv_as * p.w_inv[32 * h];                     // (one weight scale per 32-row tile: packing.qkv_folded_w2)
    const float fk = sk * inv_as * p.w_inv[CZ + 32 * h];
    const float fv = sv * inv_as * p.w_inv[2 * CZ + 32 * h];
    const int ko = (half * 32 + l31) * KP + 4 * hh;
    const int vo = l31 * VP + half * 32 + 8 * hh;
    // q (transposed accumulator: acc[r] = q[dim pd_frag_row(r, hh)][row l31]) -> the wave's Q fragments
    auto make_q = [&](const f32x16& acc) {
        unsigned ph[8], pl[8];                                          // packed pairs: group g = dims 8 g + 4 hh .. + 4 -> [2 g], [2 g + 1]
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const pd_parts2 t0 = pd_split2h(acc[4 * g] * fq, acc[4 * g + 1] * fq);
            const pd_parts2 t1 = pd_split2h(acc[4 * g + 2] * fq, acc[4 * g + 3] * fq);
            ph[2 * g] = t0.h; ph[2 * g + 1] = t1.h; pl[2 * g] = t0.l; pl[2 * g + 1] = t1.l;
        }
        // k-step st needs dims 16 st + 8 hh .. + 8 of the lane's row: groups 2 st (lower lane half owns dims + 0..3, upper + 4..7) and
        // 2 st + 1 (+ 8..11 / + 12..15).  v_permlane32_swap(a = group 2 st, b = group 2 st + 1): a' = (own a | partner's b ... ) - see below
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            u32x4 fh, fl;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                // a' : lanes 0-31 keep group 2 st (dims 16 st + 0..3), lanes 32-63 receive the lower half's group 2 st + 1 (dims 16 st + 8..11)
                // b' : lanes 0-31 receive the upper half's group 2 st (dims 16 st + 4..7), lanes 32-63 keep group 2 st + 1 (dims 16 st + 12..15)
                const auto sh = __builtin_amdgcn_permlane32_swap(ph[4 * st + e], ph[4 * st + 2 + e], false, false);
                const auto sl = __builtin_amdgcn_permlane32_swap(pl[4 * st + e], pl[4 * st + 2 + e], false, false);
                fh[e] = sh[0]; fh[2 + e] = sh[1];
                fl[e] = sl[0]; fl[2 + e] = sl[1];
            }
            qf[st][0] = __builtin_bit_cast(frag, fh);
            qf[st][1] = __builtin_bit_cast(frag, fl);
        }
    };
    // a projected 32 x 32 tile scaled and split: [g] = (high pair 0, high pair 1), (low pair 0, low pair 1) of register group g
    auto pack = [&](const f32x16& acc, float f, u32x2 (&hi)[4], u32x2 (&lo)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const pd_parts2 t0 = pd_split2h(acc[4 * g] * f, acc[4 * g + 1] * f);
            const pd_parts2 t1 = pd_split2h(acc[4 * g + 2] * f, acc[4 * g + 3] * f);
            hi[g] = u32x2{t0.h, t1.h}; lo[g] = u32x2{t0.l, t1.l};
        }
    };
    // k (transposed: acc[r] = k[dim pd_frag_row(r, hh)][key l31]) -> the key's row of the K tile
    auto store_k = [&](const u32x2 (&hi)[4], const u32x2 (&lo)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<u32x2*>(sK + ko + 8 * g) = hi[g];
            *reinterpret_cast<u32x2*>(sK + K_PART + ko + 8 * g) = lo[g];
        }
    };
    // v (straight: acc[r] = v[key pd_frag_row(r, hh)][dim l31]) -> row l31 of the V^T tile; four consecutive keys of a register group
    // sit in four consecutive columns of attn_pipe.hip's key permutation: 16 (g >> 1) + 8 hh + 4 (g & 1) + 0..3
    auto store_v = [&](const u32x2 (&hi)[4], const u32x2 (&lo)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = vo + 16 * (g >> 1) + 4 * (g & 1);
            *reinterpret_cast<u32x2*>(sV + c) = hi[g];
            *reinterpret_cast<u32x2*>(sV + V_PART + c) = lo[g];
        }
    };

    // ---- bias tiles of the attention phase (requested at the end of the projection, section 2)
    const int nsub = (nk + 31) >> 5;
    const int nkt32 = ((p.bias_nk > 0 ? p.bias_nk : nk) + 31) >> 5;
    const int nqt32 = (T + 31) >> 5;
    const float* bias_base = p.bias + (((long long)h * nqt32 + ((wave_active ? q0 : 0) >> 5)) * nkt32) * 1024;
    const auto rs_bias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias_base), 0, nkt32 * 4096, 0x00020000);
    const int boff = lane * 16;
    auto load_bias = [&](f32x16& s, int kt32) {
        if constexpr (ABL & 8) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            return;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_bias, boff, kt32 * 4096 + g * 1024, 0));
            s[4 * g] = v[0]; s[4 * g + 1] = v[1]; s[4 * g + 2] = v[2]; s[4 * g + 3] = v[3];
        }
    };
    // (the lane's output address now: row / stride arithmetic does not stay live through the attention phase)
    float* const op = row < T ? p.o + (long long)i * bs + (long long)row * ss + h * 32 + 4 * hh : nullptr;
    f32x16 o, sA, sB;
    // ---- 2. projection: weight fragments [2 parts][12 tiles][8 k-steps][64 lanes][8] (packing.split2_f16 of the [3 C][C] matrix)
    if constexpr (WLDS) {
        // 3 072 weight fragments of 16 bytes, six per thread: thread t takes fragments t, t + 512, ... of the head's slice, read in the
        // order they are stored in (coalesced kilobytes): LDS [3 tiles q, k, v][2 parts][8 k-steps][64 lanes][8].  The weights (L2 hits)
        // are requested FIRST and the first two k-steps of the wave's rows (HBM for the first head of a row) behind them: requests
        // return in order, so the LDS stores wait for the weights only and the rows travel while the block stages and meets
        frag zr[ZD + 1][2];
        {
            frag tmp[6 / RPB];
#pragma unroll
            for (int j = 0; j < 6 / RPB; ++j) {
                const int f = tid + NTHR * j;                           // (tile3 * 2 + part) * 512 + s * 64 + lane'
                const int tp = f >> 9, sl = f & 511;
                const int t3 = tp >> 1, part = tp & 1;
                tmp[j] = (ABL & 2) ? __builtin_bit_cast(frag, u32x4{0x2c002c00u + (unsigned)j, 0xac002c00u, 0x28002c00u, 0x2c00a400u})      // (not zeros: a matrix pipe fed zeros draws less power and clocks higher)
                                : reinterpret_cast<const frag*>(p.W2)[((part * 12 + 4 * t3 + h) * NKS) * 64 + sl];
            }
            PD_SB();
#pragma unroll
            for (int d = 0; d < ZD; ++d) { zr[d][0] = zfrag_g(d, 0); zr[d][1] = zfrag_g(d, 1); }
            PD_SB();
#pragma unroll
            for (int j = 0; j < 6 / RPB; ++j) *reinterpret_cast<frag*>(lds_all + (tid + NTHR * j) * 8) = tmp[j];
        }
        lds_barrier();                                                  // A: the weights are staged
        auto wl = [&](int t3, int s, int part) { return *reinterpret_cast<const frag*>(lds_all + ((((t3 * 2 + part) * NKS + s) * 64 + lane) * 8)); };
        f32x16 aq, ak, av;
#pragma unroll
        for (int r = 0; r < 16; ++r) { aq[r] = 0.f; ak[r] = 0.f; av[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            const int c = s % (ZD + 1), n = (s + ZD) % (ZD + 1);
            if (s + ZD < NKS) { zr[n][0] = zfrag_g(s + ZD, 0); zr[n][1] = zfrag_g(s + ZD, 1); }
            PD_SB();
            const frag wqh = wl(0, s, 0), wql = wl(0, s, 1), wkh = wl(1, s, 0), wkl = wl(1, s, 1), wvh = wl(2, s, 0), wvl = wl(2, s, 1);
            aq = jmma(wqh, zr[c][1], aq);
            aq = jmma(wql, zr[c][0], aq);
            aq = jmma(wqh, zr[c][0], aq);
            ak = jmma(wkh, zr[c][1], ak);
            ak = jmma(wkl, zr[c][0], ak);
            ak = jmma(wkh, zr[c][0], ak);
            av = jmma(zr[c][1], wvh, av);
            av = jmma(zr[c][0], wvl, av);
            av = jmma(zr[c][0], wvh, av);
            PD_SB();
        }
        make_q(aq);
        u32x2 kh[4], kl[4], vh[4], vl[4];
        pack(ak, fk, kh, kl);
        pack(av, fv, vh, vl);
        PD_SB();
        load_bias(sA, 0);                                               // the first two bias tiles travel while the block meets twice
        load_bias(sB, 1);
        lds_barrier();                                                  // B: every wave has left the projection: the weights are dead
        store_k(kh, kl);
        store_v(vh, vl);
    } else {
        frag zh[NKS];                                                   // high parts of all eight k-steps
#pragma unroll
        for (int s = 0; s < NKS; ++s) zh[s] = zfrag_g(s, 0);
        const frag* wbase = reinterpret_cast<const frag*>(p.W2) + lane;
        auto wfrag = [&](int tile, int s, int part) {
            if constexpr (ABL & 2) return __builtin_bit_cast(frag, u32x4{0x3c003c00u + tile, 0x3c003c00u + s, 0x3c003c00u + part, 0x3c003c00u});
            return wbase[((part * 12 + tile) * NKS + s) * 64];
        };
        // one 32-row output tile of the projection: 24 MFMAs, the weight fragments of k-step s + 1 requested in front of the MFMAs of
        // k-step s and no further ahead (sched_barrier: the row fragments already hold 64 registers).  transposed: rows of the
        // accumulator = output channels (A = weights, B = rows of z); else rows = rows of z
        auto project = [&](int wtile, bool transposed) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            // fragments of k-steps s + 1 and s + 2 are in flight while the MFMAs of k-step s issue (L2 round trips of ~1 us against 96 matrix
            // cycles per k-step; the sched_barriers keep hipcc from hoisting all sixteen requests - 64 registers - to the top)
            frag wh[3], wl[3], zl[3];
#pragma unroll
            for (int d = 0; d < 2; ++d) { wh[d] = wfrag(wtile, d, 0); wl[d] = wfrag(wtile, d, 1); zl[d] = zfrag_g(d, 1); }
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                const int c = s % 3, n = (s + 2) % 3;
                if (s + 2 < NKS) { wh[n] = wfrag(wtile, s + 2, 0); wl[n] = wfrag(wtile, s + 2, 1); zl[n] = zfrag_g(s + 2, 1); }
                PD_SB();
                if (transposed) {
                    acc = jmma(wh[c], zl[c], acc);
                    acc = jmma(wl[c], zh[s], acc);
                    acc = jmma(wh[c], zh[s], acc);
                } else {
                    acc = jmma(zl[c], wh[c], acc);
                    acc = jmma(zh[s], wl[c], acc);
                    acc = jmma(zh[s], wh[c], acc);
                }
                PD_SB();
            }
            return acc;
        };
        u32x2 ph[4], pl[4];
        make_q(project(h, true));
        pack(project(4 + h, true), fk, ph, pl);
        store_k(ph, pl);
        pack(project(8 + h, false), fv, ph, pl);
        store_v(ph, pl);
        PD_SB();                                                        // (not above the projection: its fragments hold the registers)
        load_bias(sA, 0);
        load_bias(sB, 1);
    }

    // ---- 3. attention over the resident tiles (the wave program of attn_pipe.hip)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;

    lds_barrier();                                                      // all keys of the (row, head) are resident

    const int koff = l31 * KP + 8 * hh;
    const int voff = 2 * K_PART + l31 * VP + 8 * hh;
    auto kfrag = [&](int base, int st, int pt) { return *reinterpret_cast<const frag*>(lds + base + koff + pt * K_PART + 16 * st); };
    auto vfrag = [&](int base, int st, int pt) { return *reinterpret_cast<const frag*>(lds + base + voff + pt * V_PART + 16 * st); };
    // LDS bases of sub-tile j: its K rows / its V^T columns (j beyond the last tile: clamped, the values are never used)
    auto kbase = [&](int j) { j = j < 2 * NTILE ? j : 2 * NTILE - 1; return (j >> 1) * STAGE + (j & 1) * 32 * KP; };
    auto vbase = [&](int j) { j = j < 2 * NTILE ? j : 2 * NTILE - 1; return (j >> 1) * STAGE + (j & 1) * 32; };

    float m_run = -INFINITY, l_run = 0.f, mneg_run = 0.f;
    const float tau_s = (float)LAZY / c_s;
    auto rowmax = [&](const f32x16& s) {
        float m = max3(s[0], s[1], s[2]);
        m = max3(m, s[3], s[4]); m = max3(m, s[5], s[6]); m = max3(m, s[7], s[8]); m = max3(m, s[9], s[10]);
        m = max3(m, s[11], s[12]); m = max3(m, s[13], s[14]);
        m = __builtin_fmaxf(m, s[15]);
        return pd_xhalf_max(m);
    };
    auto fe4 = [&](f32x16& s, int r0, float mneg) {
#pragma unroll
        for (int r = r0; r < r0 + 4; ++r) s[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c_s, mneg));
    };
    auto split8 = [&](const f32x16& s, int r0, frag (&pf)[2]) {
        u32x4 fh, fl;
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
            const pd_parts2 t = pd_split2h(s[r0 + 2 * e2], s[r0 + 2 * e2 + 1]);
            fh[e2] = t.h; fl[e2] = t.l;
        }
        pf[0] = __builtin_bit_cast(frag, fh);
        pf[1] = __builtin_bit_cast(frag, fl);
    };
    // one pipelined phase (attn_pipe.hip): softmax + P.V of sub-tile j (scores in `cur`) interleaved with the score MFMAs of sub-tile
    // j + 1 (`nxt`, preloaded with its bias tile); kn / vc / kn2: LDS bases of K(j + 1), V(j), K(j + 2); bt: bias tile fetched into `cur`
    auto phase = [&](f32x16& cur, f32x16& nxt, float& mloc, frag (&kf0)[2], int kn, int vc, int kn2, int bt) {
        frag kf1[2], vf0[2], vf1[2], pf0[2], pf1[2];
        nxt = mma(kf0[0], qf[0][1], nxt);
        PD_SB();
        kf1[0] = kfrag(kn, 1, 0); kf1[1] = kfrag(kn, 1, 1);
        if ((__builtin_amdgcn_ballot_w64(mloc > m_run + tau_s) & qlanes) != 0ull) {
            const float m_new = __builtin_fmaxf(m_run, mloc);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c_s);
            m_run = m_new;
            mneg_run = __builtin_fmaf(-m_new, c_s, PSH);
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] *= alpha;
            l_run *= alpha;
        }
        const float mneg = mneg_run;
        fe4(cur, 0, mneg);
        PD_SB();
        nxt = mma(kf0[1], qf[0][0], nxt);
        PD_SB();
        fe4(cur, 4, mneg);
        float ps0 = (cur[0] + cur[1]) + (cur[2] + cur[3]);
        PD_SB();
        nxt = mma(kf0[0], qf[0][0], nxt);
        PD_SB();
        vf0[0] = vfrag(vc, 0, 0); vf0[1] = vfrag(vc, 0, 1);
        split8(cur, 0, pf0);
        PD_SB();
        nxt = mma(kf1[0], qf[1][1], nxt);
        PD_SB();
        fe4(cur, 8, mneg); fe4(cur, 12, mneg);
        PD_SB();
        nxt = mma(kf1[1], qf[1][0], nxt);
        PD_SB();
        float ps1 = (cur[4] + cur[5]) + (cur[6] + cur[7]);
        ps0 += (cur[8] + cur[9]) + (cur[10] + cur[11]);
        PD_SB();
        nxt = mma(kf1[0], qf[1][0], nxt);
        PD_SB();
        ps1 += (cur[12] + cur[13]) + (cur[14] + cur[15]);
        PD_SB();
        o = mma(vf0[0], pf0[1], o);
        PD_SB();
        vf1[0] = vfrag(vc, 1, 0); vf1[1] = vfrag(vc, 1, 1);
        split8(cur, 8, pf1);
        PD_SB();
        load_bias(cur, bt);
        PD_SB();
        o = mma(vf0[1], pf0[0], o);
        PD_SB();
        l_run += ps0 + ps1;
        PD_SB();
        o = mma(vf0[0], pf0[0], o);
        PD_SB();
        kf0[0] = kfrag(kn2, 0, 0); kf0[1] = kfrag(kn2, 0, 1);
        PD_SB();
        o = mma(vf1[0], pf1[1], o);
        PD_SB();
        float m0 = max3(nxt[0], nxt[1], nxt[2]);
        m0 = max3(m0, nxt[3], nxt[4]); m0 = max3(m0, nxt[5], nxt[6]); m0 = max3(m0, nxt[7], nxt[8]);
        PD_SB();
        o = mma(vf1[1], pf1[0], o);
        PD_SB();
        m0 = max3(m0, nxt[9], nxt[10]); m0 = max3(m0, nxt[11], nxt[12]); m0 = max3(m0, nxt[13], nxt[14]);
        m0 = __builtin_fmaxf(m0, nxt[15]);
        PD_SB();
        o = mma(vf1[0], pf1[0], o);
        PD_SB();
        mloc = pd_xhalf_max(m0);
        PD_SB();
    };
    auto scores = [&](f32x16& s, int kn) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const frag kh = kfrag(kn, st, 0), kl = kfrag(kn, st, 1);
            s = mma(kh, qf[st][1], s);
            s = mma(kl, qf[st][0], s);
            s = mma(kh, qf[st][0], s);
        }
    };
    // softmax + P.V of the last sub-tile; keys >= nk are masked
    auto finish = [&](f32x16& s, int kt32, int vc) {
        if ((kt32 + 1) * 32 > nk) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kt32 * 32 + pd_frag_row(r, hh) >= nk) s[r] = -INFINITY;
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

    if (wave_active && !(ABL & 1)) {
        float mloc;
        frag kf0[2];
        scores(sA, kbase(0));
        kf0[0] = kfrag(kbase(1), 0, 0); kf0[1] = kfrag(kbase(1), 0, 1);
        mloc = rowmax(sA);
        int j = 0;
        for (; j + 2 <= nsub - 1; j += 2) {                             // afterwards sA holds the scores of sub-tile j
            phase(sA, sB, mloc, kf0, kbase(j + 1), vbase(j), kbase(j + 2), j + 2);
            phase(sB, sA, mloc, kf0, kbase(j + 2), vbase(j + 1), kbase(j + 3), j + 3);
        }
        if (j + 1 <= nsub - 1) {
            phase(sA, sB, mloc, kf0, kbase(j + 1), vbase(j), kbase(j + 2), j + 2);
            finish(sB, j + 1, vbase(j + 1));
        } else {
            finish(sA, j, vbase(j));
        }
    }

    // ---- 4. output, in accumulator order.  (The stores cost ~5 of 54 us - ablation 32 - and it is their bytes, not their shape: turning the
    // wave's tile row-major through LDS so that eight lanes write one row's 128 bytes - 4 x fewer, whole-line requests, one more block
    // barrier - measured 56.3 / 59.4 us against 57.0 / 58.6 us, profiles/r06_tri_attn_form4.txt.)
    if (op && (!(ABL & 32) || l_run == 12345.f)) {                      // (ablation 32: no output stores)
        const float inv = inv_sv / pd_xhalf_sum(l_run);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v = {o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv};
            *reinterpret_cast<f32x4*>(op + 8 * g) = v;
        }
    }
}

}  // namespace

PD_EXPORT int pd_tri_attn_args_size(void) { return (int)sizeof(pd_tri_attn_args); }

PD_EXPORT int pd_tri_attention(const pd_tri_attn_args* a, void* stream) {
    if (!a || !a->z2 || !a->W2 || !a->w_inv || !a->bias || !a->o || !a->qkv_amax) return PD_ERR_ARG;
    if (a->T <= 0 || a->Treal <= 0 || a->Treal > a->T) return PD_ERR_ARG;
    if (a->C != CZ || a->nheads != CZ / 32 || a->T > NTILE * KT || a->T % 4 != 0 || !(a->bias_prescale > 0.f) || !(a->zn_amax > 0.f))
        return PD_ERR_UNSUPPORTED;
    if (a->bias_prescale > 0x1p90f) return PD_ERR_UNSUPPORTED;         // (masked entries must stay finite: pd_attention_pipe_ok)
    if ((((uintptr_t)a->z2 | (uintptr_t)a->W2 | (uintptr_t)a->o | (uintptr_t)a->w_inv) & 15) != 0) return PD_ERR_UNSUPPORTED;
    if ((long long)a->T * a->T * CZ * 4 >= 0xffffff00ll) return PD_ERR_UNSUPPORTED;
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(tri_attn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
            return PD_ERR_LAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL(tri_attn_kernel, dim3((unsigned)(a->T / RPB), CZ / 32), dim3(NTHR), LDS_BYTES, (hipStream_t)stream, *a);
    return pd_check_launch();
}
