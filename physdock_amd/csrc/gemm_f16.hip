// fp32-accurate GEMM on the fp16 matrix pipe: two-part operand splitting with power-of-two scales ("f16 x 3").
//
// Second operand format of the split-operand GEMM (the first, three bf16 parts / six products, is gemm_split.hip, whose
// structure - persistent 128 x 128 tiles on eight waves, XCD-aware tile order, pre-split fragment-major weights fetched
// straight into MFMA registers, A split while it is staged, epilogues from gemm_tile_common.h - this file shares):
//   * every operand value times a power of two is split into a = h + l, h = fp16(a), l = fp16(a - h): 22 significand bits
//     (pd_split2h, common.h), and a product is the sum of three partial products h.h + h.l + l.h, each accumulated in fp32
//     by v_mfma_f32_32x32x16_f16 - three MFMAs per 32 x 32 x 16 block instead of six, two thirds of the operand bytes through
//     LDS / registers, 6 instead of 11 VALU operations per split pair.  Effective peak 2.5 PF / 3 = 839 TF;
//   * measured against float64 (tools/micro/f16x2_probe.hip, profiles/r03_f16x2_probe.txt; tests/test_gemm_f16_gpu.py) the
//     result is at least as accurate as v_mfma_f32_32x32x2_f32's for every K >= 32 and operand ranges up to 2^+-20 inside
//     a row - the fp32 MFMA rounds once per product, this one once per 16 products;
//   * fp16's exponent range is the price: a scaled value above 65504 would overflow.  So the format is used only where an
//     UPPER BOUND of |A| is known before the launch: the weights are scaled per output row at pack time (exact, w_inv[n]
//     undoes it in the epilogue), A by 2^e with e from the bound `a_amax` (a device scalar, so launches stay graph-capturable).
//     In the DiT blocks every A operand has a rigorous bound that follows from LayerNorm + the AdaLN table alone
//     (pd_dit_bounds, sampler.hip); everything else stays on the bf16 x 6 kernel, which needs no bound.
// A arrives as fp32 (split while it is staged, after the norm prologue and the scale) or pre-split by pd_norm_split2
// ([2][M][K] fp16, PRO == 3: the staging is a 16-byte copy).
#include <stdlib.h>
#include <type_traits>
#define PD_EPILOGUE_Y2 1      // the head-norm epilogue of THIS family can write k | v pre-split for the attention kernel (pd_gemm_args.Y2)
#include "gemm_tile_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#ifndef PD_F16_MIN_TILES
#define PD_F16_MIN_TILES 256
#endif
#ifndef PD_F16_MIN_TILES_SMALL
#define PD_F16_MIN_TILES_SMALL 160
#endif
#ifndef PD_F16_MIN_TILES_SMALL_LONGK
#define PD_F16_MIN_TILES_SMALL_LONGK 96
#endif
// lab ablations of the direct-W main loop (timing only, wrong results): 1 every block stages the A rows of tile 0 (L2-hot operand),
// 2 every wave loads the W fragments of column block 0 / k-step 0, 4 no block barrier inside the slice loop, 8 no A requests in the loop,
// 16 the A fragments are read from LDS once per tile instead of once per k-step (no LDS reads in the loop), 32 no LDS staging stores,
// 64 no W fragment requests in the loop
#ifdef PD_F16_ABL
constexpr int F16_ABL = PD_F16_ABL;
#else
constexpr int F16_ABL = 0;
#endif
#ifdef PD_LAB      // lab build only (tools/gemm_f16_trace.py): in-kernel phase trace of the direct-W main loop
__device__ unsigned long long* g_f16_trace = nullptr;
#define PD_F16_TRACE_PTR g_f16_trace
#else
#define PD_F16_TRACE_PTR ((unsigned long long*)nullptr)
#endif
constexpr int NPARTS = 2;            // operand parts: (hi, lo) fp16
constexpr int PITCH = 24;            // LDS-W tiles: 16 k per row, 48 bytes apart
constexpr int PITCH2 = 40;           // DW tiles: 32 k per row, 80 bytes apart (conflict-free ds_read_b128 fragments)

template <int BM_, int BN_, int WM_, int NWAVES_, bool DW_, int BPC_ = 0>
struct FTile {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = NWAVES_ / WM_, NT = 64 * NWAVES_;
    static constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    static constexpr bool DW = DW_;      // direct W: the B fragments go from global memory straight into MFMA registers
    static constexpr int STAGE = DW ? NPARTS * BM * PITCH2 : NPARTS * (BM + BN) * PITCH;      // fp16 elements per stage
    static constexpr int LDS_BYTES = 2 * STAGE * 2;
    static constexpr int BLOCKS_PER_CU = BPC_ ? BPC_ : (NWAVES_ == 8 ? 2 : 4);
    static constexpr int WAVES_PER_SIMD = NWAVES_ * BLOCKS_PER_CU / 4;
    static constexpr int GRID = 256 * BLOCKS_PER_CU;
};

__device__ __forceinline__ void lds_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0); outstanding global loads stay in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// four scaled values -> (hi, lo) fp16 parts, two values per packed conversion
__device__ __forceinline__ void split4h(const f32x4& v, f16x4& h, f16x4& l) {
    const pd_parts2 p0 = pd_split2h(v[0], v[1]), p1 = pd_split2h(v[2], v[3]);
    h = __builtin_bit_cast(f16x4, (u32x2){p0.h, p1.h});
    l = __builtin_bit_cast(f16x4, (u32x2){p0.l, p1.l});
}

template <int PRO, int EPI, class TL>
__global__ __launch_bounds__(TL::NT) __attribute__((amdgpu_waves_per_eu(TL::WAVES_PER_SIMD, TL::WAVES_PER_SIMD)))
void gemm_f16_kernel(const pd_gemm_args p) {
    constexpr int BM = TL::BM, BN = TL::BN, TM = TL::TM, TN = TL::TN;
    constexpr int XSLOTS = TL::GRID / 8;
    constexpr int SNT = TL::NT;
    constexpr int TPR_A = SNT / BM;              // threads per A row (4); a row of a 32-k slice = 8 f32x4 chunks
    constexpr int CPH_A = 4 / TPR_A;             // chunks per thread per 16-k half
    constexpr int TPR_W = SNT / BN;              // threads per W row; a row of a 32-k slice of one part = 4 f16x8 chunks
    constexpr int NW = 4 / TPR_W;                // W chunks per thread per part per slice
    constexpr bool DW = TL::DW;
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the fragment addresses become SGPR base + lane offset
    const int wm = wave / TL::WN, wn = wave % TL::WN;
    const int l31 = lane & 31, hh = lane >> 5;
    const int nMb = p.M / BM, nNb = p.N / BN;
    const int ntiles = nMb * nNb;
    const int nk = (p.K + 31) / 32;
    const int nks = 2 * nk;                       // 16-k steps per (zero padded) weight row
    // pre-split scaled weights, fragment-major: [2 parts][N / 32 row blocks][nks k-steps][64 lanes][8 fp16]
    const f16x8* __restrict__ W2 = reinterpret_cast<const f16x8*>(p.W2);
    const long long wpart = (long long)((p.N + 31) / 32) * nks * 64;          // in 16-byte units
    auto w2_chunk = [&](int part, int n, int kchunk) {                          // kchunk = k / 8
        return W2 + part * wpart + ((long long)(n >> 5) * nks + (kchunk >> 1)) * 64 + (kchunk & 1) * 32 + (n & 31);
    };
    // power-of-two scale of A from its magnitude bound; the epilogue undoes it together with the weights' row scales
    const float a_s = pd_pow2_scale(*p.a_amax);
    const float inv_a_s = 1.0f / a_s;

    auto sA = [&](int s, int part) { return lds + s * TL::STAGE + part * BM * PITCH; };
    auto sW = [&](int s, int part) { return lds + s * TL::STAGE + NPARTS * BM * PITCH + part * BN * PITCH; };

    const int a_row = tid / TPR_A, a_q = tid % TPR_A;
    const int w_row = tid / TPR_W, w_q = tid % TPR_W;
    // PRO == 3: A arrives pre-split and pre-scaled (pd_norm_split2, [2][M][Kp] fp16): a thread copies one 16-byte chunk per part
    constexpr bool AS = PRO == 3;
    constexpr int CPA = 4 / TPR_A;               // pre-split A: 8-k chunks per thread, part and slice (four per row)
    const _Float16* __restrict__ A2 = reinterpret_cast<const _Float16*>(p.A2);
    const long long apart = (long long)p.M * (nk * 32);
    f16x8 ra2[NPARTS][AS ? CPA : 1];
    f32x4 ra[2][CPH_A];                          // [half][i]: chunk 4*half + a_q + TPR_A*i of the thread's row
    f16x8 rw[NPARTS][NW];

    auto gload = [&](int bm0, int bn0, int k0) {
        if constexpr (F16_ABL & 1) bm0 = 0;
        const int r = bm0 + a_row;                // full tiles only: always < M
        if constexpr (AS) {
            const _Float16* ap2 = A2 + (long long)r * (nk * 32) + k0 + 8 * a_q;
#pragma unroll
            for (int part = 0; part < NPARTS; ++part)
#pragma unroll
                for (int i = 0; i < CPA; ++i) ra2[part][i] = *reinterpret_cast<const f16x8*>(ap2 + part * apart + 8 * TPR_A * i);
        }
        const float* ap = p.A + (long long)r * p.lda + k0;
#pragma unroll
        for (int h = 0; h < (AS ? 0 : 2); ++h)
#pragma unroll
            for (int i = 0; i < CPH_A; ++i) {
                int kc = 16 * h + 4 * (a_q + TPR_A * i);
                kc = k0 + kc < p.K ? kc : 0;      // clamped address; zeroed in the staging (K % 4 == 0 is required)
                ra[h][i] = *reinterpret_cast<const f32x4*>(ap + kc);
            }
        if constexpr (!DW) {
#pragma unroll
            for (int part = 0; part < NPARTS; ++part)
#pragma unroll
                for (int i = 0; i < NW; ++i)
                    rw[part][i] = *w2_chunk(part, bn0 + w_row, (k0 >> 3) + (NW == 2 ? w_q + 2 * i : w_q));
        }
    };
    // DW: this wave's B fragments of 16-k step `ks` of column block bn0, straight into MFMA operand registers
    f16x8 wf[2][TN][NPARTS];
    auto wfrag = [&](int buf, int bn0, int ks) {
        if constexpr (F16_ABL & 2) { bn0 = 0; ks = 0; }
        if constexpr (F16_ABL & 64) { if (ks > 1) return; }      // no B requests inside the slice loop
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const f16x8* base = W2 + ((long long)((bn0 + wn * (32 * TN) + j * 32) >> 5) * nks + ks) * 64 + lane;
#pragma unroll
            for (int part = 0; part < NPARTS; ++part) wf[buf][j][part] = base[part * wpart];
        }
    };

    f32x16 acc[TM][TN];
    const bool grouped = gridDim.x == TL::GRID && nMb >= 8;
    TileOrder ord;
    ord.init(nMb, nNb, grouped ? blockIdx.x & 7 : 0, grouped ? 8 : 1, XSLOTS);
    const int t_step = grouped ? XSLOTS : gridDim.x;
    const int t_end = grouped ? ord.ntiles : ntiles;
    int tile = grouped ? blockIdx.x >> 3 : blockIdx.x;
    if (tile >= t_end) return;
    auto coords = [&](int t, int& bm0, int& bn0) {
        int mb, nb;
        if (grouped) ord.get(t, mb, nb);
        else { mb = t % nMb; nb = t / nMb; }
        bm0 = mb * BM; bn0 = nb * BN;
    };
    int bm0, bn0;
    coords(tile, bm0, bn0);
    gload(bm0, bn0, 0);
    if constexpr (DW) { wfrag(0, bn0, 0); wfrag(1, bn0, 1); }

    for (; tile < t_end; tile += t_step) {
        const int n0 = bn0 + wn * (32 * TN) + l31;
        float c0[TN], c1[TN], cs[TN];
        const int gate_row = bm0 + wm * (32 * TM);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            c0[j] = p.bias ? p.bias[n0 + 32 * j] : 0.f;
            c1[j] = 1.f;
            cs[j] = p.w_inv[n0 + 32 * j] * inv_a_s;          // undoes both operand scales (powers of two: exact)
            if constexpr (EPI == EPI_HN) c1[j] = p.hn_w[((n0 + 32 * j) / p.hn_split) * 32 + l31];
            if constexpr (EPI == EPI_GATERES)
                c1[j] = p.mul ? p.mul[(long long)(gate_row / p.mul_rows_per_group) * p.mul_gstride + n0 + 32 * j] : 1.f;
        }
        // prologue state of the one A row this thread stages
        float st_mean = 0.f, st_rstd = 1.f;
        int grp_off = 0;
        if constexpr (PRO != 0 && !AS) {
            const int m = bm0 + a_row;
            st_mean = p.stats[2 * (long long)m];
            st_rstd = p.stats[2 * (long long)m + 1];
            if constexpr (PRO == 2) grp_off = (m / p.pro_rows_per_group) * p.pro_gstride;
        }
        // norm prologue, scale, k-tail zeroing and split of one f32x4 chunk
        auto prep = [&](f32x4 v, int kc, f16x4& ph, f16x4& pl) {
            if constexpr (PRO != 0) {
                const int kl = kc < p.K ? kc : 0;
                const f32x4 pw = *reinterpret_cast<const f32x4*>(p.pro_w + grp_off + kl);
                const f32x4 pb = *reinterpret_cast<const f32x4*>(p.pro_b + grp_off + kl);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (v[e] - st_mean) * st_rstd * pw[e] + pb[e];
            }
            if (p.pro_act == PD_ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (p.pro_act == PD_ACT_SILU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = pd_silu(v[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= a_s;
            if (kc >= p.K) v = f32x4{0.f, 0.f, 0.f, 0.f};
            split4h(v, ph, pl);
        };
        // ---- LDS-W tiles (GLU: a wave owns both columns of a pair): one 16-k half of the slice held in ra / rw
        auto stage = [&](int h, int k0) {
            if constexpr (AS) {
                if ((a_q >> 1) == h) {
#pragma unroll
                    for (int part = 0; part < NPARTS; ++part)
                        *reinterpret_cast<f16x8*>(sA(h, part) + a_row * PITCH + 8 * (a_q & 1)) = ra2[part][0];
                }
            }
#pragma unroll
            for (int i = 0; i < (AS ? 0 : CPH_A); ++i) {
                const int c = a_q + TPR_A * i;                    // chunk inside the half
                f16x4 ph, pl;
                prep(ra[h][i], k0 + 16 * h + 4 * c, ph, pl);
                const int o = a_row * PITCH + 4 * c;
                *reinterpret_cast<f16x4*>(sA(h, 0) + o) = ph;
                *reinterpret_cast<f16x4*>(sA(h, 1) + o) = pl;
            }
            if (!DW && (NW == 2 || (w_q >> 1) == h)) {
                const int i = NW == 2 ? h : 0;
                const int c = NW == 2 ? w_q : (w_q & 1);          // 16-byte chunk inside the half
#pragma unroll
                for (int part = 0; part < NPARTS; ++part)
                    *reinterpret_cast<f16x8*>(sW(h, part) + w_row * PITCH + 8 * c) = rw[part][i];
            }
        };
        // three partial products per (i, j) fragment pair of one 16-k stage, smallest first
        auto mma = [&](int s) {
            f16x8 fa[TM][NPARTS], fw[TN][NPARTS];
#pragma unroll
            for (int part = 0; part < NPARTS; ++part) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[i][part] = *reinterpret_cast<const f16x8*>(sA(s, part) + (wm * (32 * TM) + i * 32 + l31) * PITCH + 8 * hh);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fw[j][part] = *reinterpret_cast<const f16x8*>(sW(s, part) + (wn * (32 * TN) + j * 32 + l31) * PITCH + 8 * hh);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x16 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fw[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][1], fw[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fw[j][0], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        };
        // ---- DW tiles: the whole 32-k slice in ra goes to LDS stage s; fragments of k-step ks come from there and from wf[ks]
        auto stage2 = [&](int s, int k0) {
            if constexpr (F16_ABL & 32) { if (k0 > 0) return; }
            _Float16* base = lds + s * TL::STAGE + a_row * PITCH2;
            if constexpr (AS) {
#pragma unroll
                for (int part = 0; part < NPARTS; ++part)
#pragma unroll
                    for (int i = 0; i < CPA; ++i) *reinterpret_cast<f16x8*>(base + part * BM * PITCH2 + 8 * (a_q + TPR_A * i)) = ra2[part][i];
            }
#pragma unroll
            for (int h = 0; h < (AS ? 0 : 2); ++h)
#pragma unroll
                for (int i = 0; i < CPH_A; ++i) {
                    const int c = a_q + TPR_A * i;
                    f16x4 ph, pl;
                    prep(ra[h][i], k0 + 16 * h + 4 * c, ph, pl);
                    const int o = 16 * h + 4 * c;
                    *reinterpret_cast<f16x4*>(base + o) = ph;
                    *reinterpret_cast<f16x4*>(base + BM * PITCH2 + o) = pl;
                }
        };
        f16x8 fa_abl[TM][NPARTS];
        bool fa_abl_have = false;
        auto mma2 = [&](int s, int ks) {
            f16x8 fa[TM][NPARTS];
            const _Float16* base = lds + s * TL::STAGE + (wm * (32 * TM) + l31) * PITCH2 + 16 * ks + 8 * hh;
            if (!(F16_ABL & 16) || !fa_abl_have) {
#pragma unroll
                for (int part = 0; part < NPARTS; ++part)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        fa[i][part] = *reinterpret_cast<const f16x8*>(base + part * BM * PITCH2 + i * 32 * PITCH2);
            }
            if constexpr (F16_ABL & 16) {
                if (!fa_abl_have) {
#pragma unroll
                    for (int part = 0; part < NPARTS; ++part)
#pragma unroll
                        for (int i = 0; i < TM; ++i) fa_abl[i][part] = fa[i][part];
                    fa_abl_have = true;
                } else {
#pragma unroll
                    for (int part = 0; part < NPARTS; ++part)
#pragma unroll
                        for (int i = 0; i < TM; ++i) fa[i][part] = fa_abl[i][part];
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x16 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], wf[ks][j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][1], wf[ks][j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], wf[ks][j][0], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        };
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // slice 0 of this tile is in ra / rw (requested before the previous tile's epilogue)
        lds_barrier();                            // every wave has finished the previous tile's last stage
        if constexpr (DW) {
            unsigned long long* dbg = nullptr;
            if (PD_F16_TRACE_PTR && lane == 0 && blockIdx.x < 64 && tile < t_step)
                dbg = PD_F16_TRACE_PTR + ((long long)blockIdx.x * 8 + wave) * (6 * 64);
#define PD_STAMP(slot) if (dbg && kt < 64) dbg[kt * 6 + slot] = __builtin_amdgcn_s_memtime()
            stage2(0, 0);
            lds_barrier();
            // one 32-k slice; `more` is a compile-time flag and the last slice is peeled off the loop: with the requests inside
            // run-time conditionals the compiler's s_waitcnt insertion falls back to vmcnt(0) at the first MFMA of every slice -
            // every wave then sat out the full latency of the A loads it had just issued (phase trace: tools/gemm_f16_trace.py)
            auto slice = [&](int kt, auto more_c) {
                constexpr bool more = decltype(more_c)::value;
                const int st = kt & 1;
                PD_STAMP(0);
                if constexpr (more && !(F16_ABL & 8)) { gload(bm0, bn0, (kt + 1) * 32); __builtin_amdgcn_sched_barrier(0); }   // requests stay where they are written
                mma2(st, 0);
                PD_STAMP(1);
                // every B buffer is re-requested right after its last use
                if constexpr (more) { __builtin_amdgcn_sched_barrier(0); wfrag(0, bn0, 2 * kt + 2); __builtin_amdgcn_sched_barrier(0); }
                mma2(st, 1);
                PD_STAMP(2);
                if constexpr (more) {
                    __builtin_amdgcn_sched_barrier(0);
                    wfrag(1, bn0, 2 * kt + 3);
                    __builtin_amdgcn_sched_barrier(0);
                    PD_STAMP(3);
                    // the other stage was last read in the previous iteration, which every wave left through its barrier
                    stage2(st ^ 1, (kt + 1) * 32);
                    PD_STAMP(4);
                    if constexpr (!(F16_ABL & 4)) lds_barrier();
                    PD_STAMP(5);
                }
            };
            for (int kt = 0; kt + 1 < nk; ++kt) slice(kt, std::true_type{});
            slice(nk - 1, std::false_type{});
#undef PD_STAMP
        } else {
            stage(0, 0);
            stage(1, 0);
            lds_barrier();
            for (int kt = 0; kt < nk; ++kt) {
                const bool more = kt + 1 < nk;
                if (more) gload(bm0, bn0, (kt + 1) * 32);
                mma(0);
                // every wave has read stage 0, and the stage-1 stores of the previous iteration (issued after ITS second
                // barrier) become visible to the mma(1) below - so this barrier is needed in the last iteration too
                if (nk > 1) lds_barrier();
                if (more) stage(0, (kt + 1) * 32);
                mma(1);
                if (more) {
                    lds_barrier();                    // stage 1 read by every wave; the stage-0 stores above are visible
                    stage(1, (kt + 1) * 32);
                }
            }
        }
        const int cur_bm0 = bm0, cur_bn0 = bn0;
        if (tile + t_step < t_end) {
            coords(tile + t_step, bm0, bn0);
            gload(bm0, bn0, 0);
            if constexpr (DW) { wfrag(0, bn0, 0); wfrag(1, bn0, 1); }
        }
        // back to the units of A . W^T (exact: powers of two), then the shared epilogues
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= cs[j];
        epilogue<EPI, TM, TN>(p, acc, c0, c1, cur_bm0, cur_bn0, wm, wn, l31, hh);
    }
}

#ifdef PD_LAB
}  // namespace
extern "C" __attribute__((visibility("default"))) int pd_lab_set_f16_trace(void* buf) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_f16_trace), &q, sizeof(q)) == hipSuccess ? PD_OK : PD_ERR_LAUNCH;
}
namespace {
#endif

template <int PRO, int EPI, class TL>
int run_f16(int op, const pd_gemm_args* p, hipStream_t s) {
    auto k = gemm_f16_kernel<PRO, EPI, TL>;
    constexpr int lds = TL::LDS_BYTES;
    if (op == 1)
        return hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess
                   ? PD_OK : PD_ERR_LAUNCH;
    const long long ntiles = (long long)(p->M / TL::BM) * (p->N / TL::BN);
    hipLaunchKernelGGL(k, dim3((unsigned)(ntiles < TL::GRID ? ntiles : TL::GRID)), dim3(TL::NT), lds, s, *p);
    return pd_check_launch();
}


// ---- "rows" kernel: K = 128 (the atom-level q | k | v projection of a DiT block, transformers.py:149-175 at c_a = 128).
// As statistics launch + gemm_f16_kernel<1 | 2, HN> this projection read its 67 MB of rows four times (statistics, then once per
// 128-column tile, each pass normalising and splitting them again in its staging) and ran at 96.7 + 14.7 us for 12.9 GFLOP /
// 268 MB.  Here a block owns 128 whole rows, as transition_f16.hip does: four threads per row compute its statistics (the
// arithmetic of pd_rowstats: mean, then centred squares), apply the norm / AdaLN prologue and the operand scale, split, and leave
// the two fp16 parts of the tile in LDS for ALL column tiles; the weights stream from L2 as MFMA fragments, three 16-k steps
// ahead.  No barrier after the prologue: the waves walk the column tiles and their epilogues at their own pace.
constexpr int RLP = 136;                 // LDS row pitch in fp16 (272 bytes = 17 x 16: conflict-free ds_read_b128 fragments)
// rows per block BM: 128 on eight waves (two blocks per CU: 88 vs 94 us at 64 samples) or 64 on four (four blocks per CU: launches
// of 256 - 1023 row tiles of 64, 10 - 31 samples: 31.8 vs 42.3 us at 20)
template <int PRO, int EPI, int BM>
__global__ __launch_bounds__(4 * BM) __attribute__((amdgpu_waves_per_eu(4, 4)))
void gemm_f16_rows_kernel(const pd_gemm_args p) {
    constexpr int TM = 2, TN = 1, PART = BM * RLP, NKS = 8, PF = 3, KC = 128;
#ifdef PD_F16_ROWS_NO_XPF      // lab: the next column tile's first W fragments requested after, not before, this tile's epilogue
    constexpr bool XPF = false;
#else
    constexpr bool XPF = true;
#endif
    static_assert(NKS % (PF + 1) == 0, "the fragment ring must be back at buffer 0 when a column tile ends");
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hh = lane >> 5;
    const float a_s = pd_pow2_scale(*p.a_amax);
    const float inv_a_s = 1.0f / a_s;
    const int ntiles = p.M / BM, nNb = p.N / 128;
    const int wpart = (p.N >> 5) * NKS * 1024;                       // bytes per part of the fragment-major weights
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W2), 0, 2 * wpart, 0x00020000);
    const int loff = lane * 16;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * BM;
        {   // ---- prologue: statistics + norm + scale + split of the tile's rows (four threads per row, 16-byte chunks interleaved)
            const int r = tid >> 2, q = tid & 3;
            const int m = row0 + r;
            const float* xr = p.A + (long long)m * p.lda;
            const int goff = PRO == 2 ? (m / p.pro_rows_per_group) * p.pro_gstride : 0;
            // (the gain / shift rows are read chunk by chunk in the write loop - cache hits: holding them across the two reductions
            //  as transition_f16.hip does at 256 registers spills 100 registers at this kernel's 128)
            f32x4 v[8];
            float s1 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * (q + 4 * i));
                s1 += v[i][0] + v[i][1] + v[i][2] + v[i][3];
            }
            s1 += __shfl_xor(s1, 1);
            s1 += __shfl_xor(s1, 2);
            float mean = p.stats_inline == 2 ? s1 * (1.0f / KC) : 0.f;
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; sq += d * d; }
            sq += __shfl_xor(sq, 1);
            sq += __shfl_xor(sq, 2);
            float rstd = rsqrtf(sq * (1.0f / KC) + p.stats_eps) * a_s;           // the operand scale rides on rstd and on the shift
            if (p.stats) {                             // statistics supplied (a by-product of another pass over the rows, e.g. pd_pair_bias)
                mean = p.stats[2 * (long long)m];
                rstd = p.stats[2 * (long long)m + 1] * a_s;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = 4 * (q + 4 * i);
                const f32x4 gw = *reinterpret_cast<const f32x4*>(p.pro_w + goff + c), gb = *reinterpret_cast<const f32x4*>(p.pro_b + goff + c);
                float t[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = (v[i][e] - mean) * rstd * gw[e] + gb[e] * a_s;
                const pd_parts2 p0 = pd_split2h(t[0], t[1]), p1 = pd_split2h(t[2], t[3]);
                *reinterpret_cast<u32x2*>(lds + r * RLP + c) = u32x2{p0.h, p1.h};
                *reinterpret_cast<u32x2*>(lds + PART + r * RLP + c) = u32x2{p0.l, p1.l};
            }
        }
        lds_barrier();
        const _Float16* abase = lds + (64 * wm + l31) * RLP + 8 * hh;
        f16x8 wf[PF + 1][NPARTS];
        auto wload = [&](int cb, int buf, int ks) {                       // cb: the wave's 32-column block of W
            const int so = (cb * NKS + ks) * 1024;
            wf[buf][0] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, loff, so, 0));
            wf[buf][1] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, loff, so + wpart, 0));
        };
#pragma unroll
        for (int ks = 0; ks < PF; ++ks) wload(wn, ks, ks);
#pragma unroll 1
        for (int nt = 0; nt < nNb; ++nt) {
            const int bn0 = nt * 128;
            const int cb = (bn0 >> 5) + wn;
            if (!XPF && nt > 0) {
#pragma unroll
                for (int ks = 0; ks < PF; ++ks) wload(cb, ks, ks);
            }
            const int n0 = bn0 + wn * 32 + l31;
            float c0[TN], c1[TN];
            c0[0] = p.bias ? p.bias[n0] : 0.f;
            c1[0] = 1.f;
            if constexpr (EPI == EPI_HN) c1[0] = p.hn_w[(n0 / p.hn_split) * 32 + l31];
            const float cs = p.w_inv[n0] * inv_a_s;
            f32x16 acc[TM][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                if (ks + PF < NKS) wload(cb, (ks + PF) % (PF + 1), ks + PF);
                // (without the fence hipcc hoists the fragment reads of all eight k-steps - 128 registers - to the head of the tile and spills them)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const f16x8 a0 = *reinterpret_cast<const f16x8*>(abase + 32 * i * RLP + 16 * ks);
                    const f16x8 a1 = *reinterpret_cast<const f16x8*>(abase + PART + 32 * i * RLP + 16 * ks);
                    f32x16 t = acc[i][0];
                    t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, wf[ks % (PF + 1)][1], t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, wf[ks % (PF + 1)][0], t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, wf[ks % (PF + 1)][0], t, 0, 0, 0);
                    acc[i][0] = t;
                }
            }
            // the first fragments of the next column tile travel during this tile's epilogue (NKS % (PF + 1) == 0: buffers 0 .. PF - 1 again)
            if (XPF && nt + 1 < nNb) {
#pragma unroll
                for (int ks = 0; ks < PF; ++ks) wload(cb + 4, ks, ks);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][0][r] *= cs;
            epilogue<EPI, TM, TN>(p, acc, c0, c1, row0, bn0, wm, wn, l31, hh);
        }
        lds_barrier();                               // every wave has read the tile: the next one may overwrite it
    }
}


// ---- "wide rows" kernel: K = 512 (the token-level q | k | v projection of a DiT block, transformers.py:149-175 at c_s = 512).
// The A-stationary form of the rows kernel above for rows that fill LDS: a block of SIXTEEN waves owns 64 whole rows (2 x 64 x
// 1040 bytes = 130 KB of LDS, one block per CU, four waves per SIMD), sixteen threads per row compute the statistics, normalise,
// modulate, scale and split them ONCE; then every wave takes every sixteenth 32-column block of the output and runs its whole
// K = 512 contraction against the resident tile - 192 MFMAs per block with nothing but LDS fragment reads and W fragments from
// L2 (four 16-k steps in flight, the ring runs on into the wave's next column block) - and the shared epilogue.  Against
// pd_norm_split2 + gemm_f16_kernel<3, .>: no normalised copy of the activations through HBM (33 MB out, 12 x re-read), no A
// requests, LDS stores or block barriers in the main loop.  The price: every CU streams the whole weight matrix once per 64 rows.
constexpr int WLP = 520;                 // LDS row pitch in fp16 (1040 bytes = 65 x 16: conflict-free ds_read_b128 fragments)
constexpr int WROWS_LDS_BYTES = 2 * 64 * WLP * 2;

// Wave tile: 64 rows x 32 columns (TM = 2, TN = 1) for the plain / head-norm epilogues, 64 x 64 packed columns (TM = 2, TN = 2: a GLU
// pair needs both of its columns in one wave) for the SwiGLU up-projection.  Work items (row group, column unit) are dealt to the
// waves round-robin; GK = depth of the W fragment ring.
template <int PRO, int EPI, int TM, int TN, int GK = 4, int NWV = 16>
__global__ __launch_bounds__(64 * NWV) __attribute__((amdgpu_waves_per_eu(NWV / 4, 4)))
void gemm_f16_wrows_kernel(const pd_gemm_args p) {
    constexpr int BM = 64, KC = 512, PART = BM * WLP, NKS = KC / 16, PF = GK - 1, RG = 2 / TM;
    constexpr int RPP = 4 * NWV;                   // rows per prologue pass (sixteen threads per row)
    static_assert(NKS % (PF + 1) == 0, "the fragment ring must be back at buffer 0 when a work item ends");
    static_assert(TM * RG == 2 && NWV % RG == 0, "a wave keeps its row group");
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const float a_s = pd_pow2_scale(*p.a_amax);
    const float inv_a_s = 1.0f / a_s;
    const int ntiles = p.M / BM, ncb = p.N >> 5;
    const int nitems = RG * (ncb / TN);
    const int wpart = ncb * NKS * 1024;                              // bytes per part of the fragment-major weights
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W2), 0, 2 * wpart, 0x00020000);
    const int loff = lane * 16;
    const int rh = wave % RG;                                        // this wave's row group (32 TM rows)

    // Launches with fewer row tiles than CUs: nsplit blocks share a row tile - each stages the rows (the norm is cheap) and takes its
    // share of the column units - so that 80 tiles (20 samples) still put 240 blocks on the chip.  The launcher's grid says so.
    const int nsplit = (int)gridDim.x > ntiles ? (int)gridDim.x / ntiles : 1;
    for (int work = blockIdx.x; work < ntiles * nsplit; work += gridDim.x) {
        const int tile = work / nsplit, part = work - tile * nsplit;
        const int item0 = part * nitems / nsplit, item1 = (part + 1) * nitems / nsplit;
        const int row0 = tile * BM;
        if constexpr (PRO == 3) {
            // A arrives pre-split and pre-scaled ([2][M][512] fp16: pd_norm_split2, or the attention kernel's O2): 16-byte copies
            const int q = tid & 15;
#pragma unroll 1
            for (int r = tid >> 4; r < BM; r += RPP) {
            const _Float16* a2 = reinterpret_cast<const _Float16*>(p.A2) + (long long)(row0 + r) * KC;
            f16x8 c[NPARTS][4];
#pragma unroll
            for (int part = 0; part < NPARTS; ++part)
#pragma unroll
                for (int i = 0; i < 4; ++i) c[part][i] = *reinterpret_cast<const f16x8*>(a2 + (long long)part * p.M * KC + 8 * (q + 16 * i));
#pragma unroll
            for (int part = 0; part < NPARTS; ++part)
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<f16x8*>(lds + part * PART + r * WLP + 8 * (q + 16 * i)) = c[part][i];
            }
        } else if constexpr (PRO == 0) {
            // plain fp32 rows (no norm: the attention output in front of linear_o at a handful of samples): scale, split, stage
            const int q = tid & 15;
#pragma unroll 1
            for (int r = tid >> 4; r < BM; r += RPP) {
                const float* xr = p.A + (long long)(row0 + r) * p.lda;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int c = 4 * (q + 16 * i);
                    const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
                    const pd_parts2 p0 = pd_split2h(v[0] * a_s, v[1] * a_s), p1 = pd_split2h(v[2] * a_s, v[3] * a_s);
                    *reinterpret_cast<u32x2*>(lds + r * WLP + c) = u32x2{p0.h, p1.h};
                    *reinterpret_cast<u32x2*>(lds + PART + r * WLP + c) = u32x2{p0.l, p1.l};
                }
            }
        } else {   // ---- prologue: sixteen threads (one DPP row) per row, 16-byte chunks interleaved; 4 NWV rows per pass
            const int q = tid & 15;
#pragma unroll 1
            for (int r = tid >> 4; r < BM; r += RPP) {
            const int m = row0 + r;
            const float* xr = p.A + (long long)m * p.lda;
            const int goff = PRO == 2 ? (m / p.pro_rows_per_group) * p.pro_gstride : 0;
            f32x4 v[8];
            float s1 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * (q + 16 * i));
                s1 += v[i][0] + v[i][1] + v[i][2] + v[i][3];
            }
            auto row16 = [](float x) {                  // sum over the 16 lanes of a DPP row, in every lane
                x += pd_dpp<0xB1>(x); x += pd_dpp<0x4E>(x); x += pd_dpp<0x141>(x); x += pd_dpp<0x140>(x);
                return x;
            };
            s1 = row16(s1);
            float mean = p.stats_inline == 2 ? s1 * (1.0f / KC) : 0.f;
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; sq += d * d; }
            sq = row16(sq);
            float rstd = rsqrtf(sq * (1.0f / KC) + p.stats_eps) * a_s;           // the operand scale rides on rstd and on the shift
            if (p.stats) {
                mean = p.stats[2 * (long long)m];
                rstd = p.stats[2 * (long long)m + 1] * a_s;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = 4 * (q + 16 * i);
                const f32x4 gw = *reinterpret_cast<const f32x4*>(p.pro_w + goff + c), gb = *reinterpret_cast<const f32x4*>(p.pro_b + goff + c);
                float t[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = (v[i][e] - mean) * rstd * gw[e] + gb[e] * a_s;
                const pd_parts2 p0 = pd_split2h(t[0], t[1]), p1 = pd_split2h(t[2], t[3]);
                *reinterpret_cast<u32x2*>(lds + r * WLP + c) = u32x2{p0.h, p1.h};
                *reinterpret_cast<u32x2*>(lds + PART + r * WLP + c) = u32x2{p0.l, p1.l};
            }
            }
        }
        lds_barrier();
        const _Float16* abase = lds + (32 * TM * rh + l31) * WLP + 8 * hh;
        f16x8 wf[PF + 1][TN][NPARTS];
        auto wload = [&](int cu, int buf, int ks) {                       // cu: column unit = TN consecutive 32-column blocks of W
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int so = ((cu * TN + j) * NKS + ks) * 1024;
                wf[buf][j][0] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, loff, so, 0));
                wf[buf][j][1] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, loff, so + wpart, 0));
            }
        };
        if (item0 + wave < item1) {
#pragma unroll
            for (int ks = 0; ks < PF; ++ks) wload((item0 + wave) / RG, ks, ks);
        }
#pragma unroll 1
        for (int item = item0 + wave; item < item1; item += NWV) {
            const int cu = item / RG;
            const int n0 = cu * (32 * TN) + l31;
            float c0[TN], c1[TN], cs[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                c0[j] = p.bias ? p.bias[n0 + 32 * j] : 0.f;
                c1[j] = 1.f;
                if constexpr (EPI == EPI_HN) c1[j] = p.hn_w[((n0 + 32 * j) / p.hn_split) * 32 + l31];
                if constexpr (EPI == EPI_GATERES)        // one gate row per row group (groups are whole 64-row tiles)
                    c1[j] = p.mul ? p.mul[(long long)((row0 + 32 * TM * rh) / p.mul_rows_per_group) * p.mul_gstride + n0 + 32 * j] : 1.f;
                cs[j] = p.w_inv[n0 + 32 * j] * inv_a_s;
            }
            f32x16 acc[TM][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            // four 16-k steps per trip (one turn of the fragment ring); the last trip's requests are the next work item's first
            auto group = [&](int ks0, auto last_c) {
                constexpr bool last = decltype(last_c)::value;
#pragma unroll
                for (int jj = 0; jj < PF + 1; ++jj) {
                    const int ks = ks0 + jj;
                    if constexpr (!last) wload(cu, (jj + PF) % (PF + 1), ks + PF);
                    else if (jj == 0) wload(cu, PF % (PF + 1), ks + PF);              // ks0 + PF = NKS - 1: still this item
                    else if (item + NWV < item1) wload((item + NWV) / RG, (jj + PF) % (PF + 1), jj - 1);
                    __builtin_amdgcn_sched_barrier(0);          // keeps the fragment reads of later steps where they are (registers)
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const f16x8 a0 = *reinterpret_cast<const f16x8*>(abase + 32 * i * WLP + 16 * ks);
                        const f16x8 a1 = *reinterpret_cast<const f16x8*>(abase + PART + 32 * i * WLP + 16 * ks);
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            f32x16 t = acc[i][j];
                            t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, wf[jj][j][1], t, 0, 0, 0);
                            t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, wf[jj][j][0], t, 0, 0, 0);
                            t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, wf[jj][j][0], t, 0, 0, 0);
                            acc[i][j] = t;
                        }
                    }
                }
            };
#pragma unroll 1
            for (int ks0 = 0; ks0 < NKS - (PF + 1); ks0 += PF + 1) group(ks0, std::false_type{});
            group(NKS - (PF + 1), std::true_type{});
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] *= cs[j];
            epilogue<EPI, TM, TN>(p, acc, c0, c1, row0, cu * (32 * TN), rh, 0, l31, hh);
        }
        lds_barrier();                               // every wave has read the tile: the next one may overwrite it
    }
}


// ---- wide rows, 64 x 64 wave units with a K-SPLIT TAIL (round 5): the head-norm / plain / gate-residual projections on the unit shape
// that took the SwiGLU projection from 0.33 to 0.44 of the pipe (an A fragment pair feeds two column blocks, a W fragment pair two row
// blocks: 12 MFMAs per 16-k step, per four ds_read_b128 and four 16-byte W requests - the 64 x 32 items issue the same requests per SIX).
// N = 1536 is 24 such units for sixteen waves - one and a half rounds, which is why round 4 left q | k | v on 64 x 32 items.  Here every
// wave takes full units while whole rounds of sixteen last and the remaining EIGHT units are shared by wave pairs (w, w + 8): each wave
// contracts one K half (256) of the pair's unit, the partial tiles meet in LDS - in the A tile's space, which is dead once every wave has
// left its main loop - and each wave of a pair finishes one 32-column half of the unit (sum of two numbers: order-free, bit-reproducible).
// Every wave issues the same MFMA count (N = 1536: 576, N = 512: 192).  Units: N / 64 = 16 f + {0, 8}.
template <int PRO, int EPI>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4)))
void gemm_f16_wrows_ks_kernel(const pd_gemm_args p) {
    constexpr int BM = 64, KC = 512, PART = BM * WLP, NKS = KC / 16, NWV = 16, RPP = 4 * NWV;
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const float a_s = pd_pow2_scale(*p.a_amax);
    const float inv_a_s = 1.0f / a_s;
    const int ntiles = p.M / BM, ncb = p.N >> 5, nunits = ncb >> 1;
    const int nfull = (nunits >> 4) << 4;                            // units taken whole, sixteen per round
    const bool tail = nunits > nfull;                                // eight more, one per wave pair
    const int wpart = ncb * NKS * 1024;
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W2), 0, 2 * wpart, 0x00020000);
    const int loff = lane * 16;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * BM;
        if constexpr (PRO == 3) {
            const int q = tid & 15;
#pragma unroll 1
            for (int r = tid >> 4; r < BM; r += RPP) {
                const _Float16* a2 = reinterpret_cast<const _Float16*>(p.A2) + (long long)(row0 + r) * KC;
                f16x8 c[NPARTS][4];
#pragma unroll
                for (int part = 0; part < NPARTS; ++part)
#pragma unroll
                    for (int i = 0; i < 4; ++i) c[part][i] = *reinterpret_cast<const f16x8*>(a2 + (long long)part * p.M * KC + 8 * (q + 16 * i));
#pragma unroll
                for (int part = 0; part < NPARTS; ++part)
#pragma unroll
                    for (int i = 0; i < 4; ++i) *reinterpret_cast<f16x8*>(lds + part * PART + r * WLP + 8 * (q + 16 * i)) = c[part][i];
            }
        } else {   // sixteen threads (one DPP row) per row: statistics, LayerNorm / RMSNorm, modulation, operand scale, split - once
            const int q = tid & 15;
#pragma unroll 1
            for (int r = tid >> 4; r < BM; r += RPP) {
                const int m = row0 + r;
                const float* xr = p.A + (long long)m * p.lda;
                const int goff = PRO == 2 ? (m / p.pro_rows_per_group) * p.pro_gstride : 0;
                f32x4 v[8];
                float s1 = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * (q + 16 * i));
                    s1 += v[i][0] + v[i][1] + v[i][2] + v[i][3];
                }
                auto row16 = [](float x) {
                    x += pd_dpp<0xB1>(x); x += pd_dpp<0x4E>(x); x += pd_dpp<0x141>(x); x += pd_dpp<0x140>(x);
                    return x;
                };
                s1 = row16(s1);
                float mean = p.stats_inline == 2 ? s1 * (1.0f / KC) : 0.f;
                float sq = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; sq += d * d; }
                sq = row16(sq);
                const float rstd = rsqrtf(sq * (1.0f / KC) + p.stats_eps) * a_s;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int c = 4 * (q + 16 * i);
                    const f32x4 gw = *reinterpret_cast<const f32x4*>(p.pro_w + goff + c), gb = *reinterpret_cast<const f32x4*>(p.pro_b + goff + c);
                    float t[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = (v[i][e] - mean) * rstd * gw[e] + gb[e] * a_s;
                    const pd_parts2 p0 = pd_split2h(t[0], t[1]), p1 = pd_split2h(t[2], t[3]);
                    *reinterpret_cast<u32x2*>(lds + r * WLP + c) = u32x2{p0.h, p1.h};
                    *reinterpret_cast<u32x2*>(lds + PART + r * WLP + c) = u32x2{p0.l, p1.l};
                }
            }
        }
        lds_barrier();
        const _Float16* abase = lds + l31 * WLP + 8 * hh;
        f16x8 wf[2][2][NPARTS];                                          // [ring buffer][column block of the unit][part]
        auto wload = [&](int cu, int buf, int ks) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int so = ((cu * 2 + j) * NKS + ks) * 1024;
                wf[buf][j][0] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, loff, so, 0));
                wf[buf][j][1] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, loff, so + wpart, 0));
            }
        };
        f32x16 acc[2][2];
        auto zero = [&]() {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        };
        auto kstep = [&](int buf, int ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f16x8 a0 = *reinterpret_cast<const f16x8*>(abase + 32 * i * WLP + 16 * ks);
                const f16x8 a1 = *reinterpret_cast<const f16x8*>(abase + PART + 32 * i * WLP + 16 * ks);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16 t = acc[i][j];
                    t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, wf[buf][j][1], t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, wf[buf][j][0], t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, wf[buf][j][0], t, 0, 0, 0);
                    acc[i][j] = t;
                }
            }
        };
        // k-steps ks0 .. ks0 + n (n even) of unit cu against the resident rows; buffer 0 holds step ks0 on entry; the last step's request
        // is the FIRST step of the wave's next segment (ncu, nks0), if any - the ring runs on across units
        auto contract = [&](int cu, int ks0, int n, bool more, int ncu, int nks0) {
#pragma unroll 1
            for (int ks = ks0; ks < ks0 + n; ks += 2) {
                wload(cu, 1, ks + 1);
                __builtin_amdgcn_sched_barrier(0);
                kstep(0, ks);
                if (ks + 2 < ks0 + n) wload(cu, 0, ks + 2);
                else if (more) wload(ncu, 0, nks0);
                __builtin_amdgcn_sched_barrier(0);
                kstep(1, ks + 1);
            }
        };
        auto consts = [&](int col0, int nj, float (&c0)[2], float (&c1)[2], float (&cs)[2]) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n0 = col0 + 32 * j + l31;
                const bool on = j < nj;
                c0[j] = (on && p.bias) ? p.bias[n0] : 0.f;
                c1[j] = 1.f;
                if constexpr (EPI == EPI_HN) c1[j] = on ? p.hn_w[(n0 / p.hn_split) * 32 + l31] : 1.f;
                if constexpr (EPI == EPI_GATERES)
                    c1[j] = (on && p.mul) ? p.mul[(long long)(row0 / p.mul_rows_per_group) * p.mul_gstride + n0] : 1.f;
                cs[j] = on ? p.w_inv[n0] * inv_a_s : 0.f;
            }
        };
        const int pr = wave & 7, kh = wave >> 3;                          // tail: pair and K half
        const int tcu = nfull + pr;
        int cu = wave;
        if (cu < nfull) wload(cu, 0, 0);
        else if (tail) wload(tcu, 0, kh * (NKS / 2));
        for (; cu < nfull; cu += NWV) {
            zero();
            const bool more_full = cu + NWV < nfull;
            contract(cu, 0, NKS, more_full || tail, more_full ? cu + NWV : tcu, more_full ? 0 : kh * (NKS / 2));
            float c0[2], c1[2], cs[2];
            consts(cu * 64, 2, c0, c1, cs);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] *= cs[j];
            epilogue<EPI, 2, 2>(p, acc, c0, c1, row0, cu * 64, 0, 0, l31, hh);
        }
        if (tail) {
            zero();
            contract(tcu, kh * (NKS / 2), NKS / 2, false, 0, 0);
        }
        lds_barrier();                               // every wave has left its main loop: the A tile's space is free
        if (tail) {
            // the partial tile of the column block the PARTNER finishes goes to this wave's 8 KB slot: [row block][register quad][lane]
            // (kh as a compile-time constant: a run-time index into the accumulator array would move it to scratch memory)
            auto exchange = [&](auto khc) {
                constexpr int KH = decltype(khc)::value;
                float* xch = reinterpret_cast<float*>(lds) + wave * 2048;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x16& t = acc[i][1 - KH];
                        *reinterpret_cast<f32x4*>(xch + ((i * 4 + g) * 64 + lane) * 4) = f32x4{t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]};
                    }
                lds_barrier();
                const float* xin = reinterpret_cast<const float*>(lds) + (wave ^ 8) * 2048;
                float c0[2], c1[2], cs[2];
                consts(tcu * 64 + 32 * KH, 1, c0, c1, cs);
                f32x16 fin[2][1];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 o = *reinterpret_cast<const f32x4*>(xin + ((i * 4 + g) * 64 + lane) * 4);
                        const f32x16& t = acc[i][KH];
#pragma unroll
                        for (int e = 0; e < 4; ++e) fin[i][0][4 * g + e] = (t[4 * g + e] + o[e]) * cs[0];
                    }
                const float d0[1] = {c0[0]}, d1[1] = {c1[0]};
                epilogue<EPI, 2, 1>(p, fin, d0, d1, row0, tcu * 64 + 32 * KH, 0, 0, l31, hh);
            };
            if (kh == 0) exchange(std::integral_constant<int, 0>{}); else exchange(std::integral_constant<int, 1>{});
            lds_barrier();                           // the slots are read: the next tile's rows may overwrite them
        }
    }
}

template <int PRO, int EPI>
int run_f16_wrows_ks(int op, const pd_gemm_args* p, hipStream_t s) {
    auto k = gemm_f16_wrows_ks_kernel<PRO, EPI>;
    if (op == 1)
        return hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, WROWS_LDS_BYTES) == hipSuccess
                   ? PD_OK : PD_ERR_LAUNCH;
    const int ntiles = p->M / 64;
    hipLaunchKernelGGL(k, dim3((unsigned)(ntiles < 256 ? ntiles : 256)), dim3(1024), WROWS_LDS_BYTES, s, *p);
    return pd_check_launch();
}

// the K-split-tail kernel takes a launch when the row tiles fill the chip by themselves and N is 16 f + {0, 8} units of 64 columns
#ifndef PD_F16_WROWS_KS
#define PD_F16_WROWS_KS 0      // measured slower (round 5: q | k | v 104 -> 127 us, linear_o 35 -> 49 us, 157.9 -> 147.5 poses/s): lab knob
#endif
inline bool wrows_ks_shape(const pd_gemm_args* p) {
    const int units = p->N / 64;
    return PD_F16_WROWS_KS && p->N % 64 == 0 && (units % 16 == 0 || units % 16 == 8) && p->M / 64 >= 256;
}

// lab knobs of the column split: most blocks per row tile, fewest work items a block may be left with
#ifndef PD_F16_WROWS_MAX_SPLIT
#define PD_F16_WROWS_MAX_SPLIT 4
#endif
#ifndef PD_F16_WROWS_MIN_ITEMS
#define PD_F16_WROWS_MIN_ITEMS 8
#endif
#ifndef PD_F16_WROWS_ROUNDS
#define PD_F16_WROWS_ROUNDS 1      // lab: 0 = the round-5 split rule (256 / ntiles) for 129 - 255 row tiles
#endif
template <int PRO, int EPI, int TM, int TN, int GK = 4, int NWV = 16>
int run_f16_wrows(int op, const pd_gemm_args* p, hipStream_t s) {
    auto k = gemm_f16_wrows_kernel<PRO, EPI, TM, TN, GK, NWV>;
    if (op == 1)
        return hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, WROWS_LDS_BYTES) == hipSuccess
                   ? PD_OK : PD_ERR_LAUNCH;
    const int ntiles = p->M / 64;
    // fewer row tiles than CUs: up to four blocks share a tile's columns (norm-prologue forms: re-staging the rows is cheap)
    const int nitems = (2 / TM) * ((p->N >> 5) / TN);
    // (measured per call at 1 / 2 / 3 / 5 / 8 / 12 samples of 256 tokens: up to 4 blocks with >= 8 items each 89 / 107 / 127 / 136 / 138 /
    //  143 ms; up to 16 blocks with >= 2 items 89 / 100 / 110 / 113 / 135 / 145 ms: the deep split below 32 row tiles only)
    const bool few = ntiles < 32;
    int nsplit = (PRO != 3 && ntiles < 256) ? 256 / ntiles : 1;
    const int max_split = few ? 4 * PD_F16_WROWS_MAX_SPLIT : PD_F16_WROWS_MAX_SPLIT;
    nsplit = nsplit > max_split ? max_split : nsplit;
    while (nsplit > 1 && nitems / nsplit < (few ? PD_F16_WROWS_MIN_ITEMS / 4 : PD_F16_WROWS_MIN_ITEMS)) --nsplit;
    // Round 6 (the sample-count cliffs): between 129 and 255 row tiles (33 - 63 samples of 256 tokens) 256 / ntiles is 1 and
    // every block walks ALL columns while up to half the CUs idle - 36 samples ran the SwiGLU projection in 92 us against 68 us at 32.  One
    // block per CU means whole rounds: s blocks per row tile cost ceil(ntiles s / 256) rounds of (staging + 1 / s of the columns), staging
    // ~ 15 % of a full tile (fitted: with 8 % the model picked s = 4 for 48 samples, measured 326.8 -> 333.8 ms per call; 36 / 40 samples
    // take s = 3: 291.1 -> 277.8 / 299.4 -> 288.9 ms, profiles/r06_ab_wrows_rounds.txt); take the cheapest s (the smallest on a tie).
    if (PRO != 3 && ntiles > 128 && ntiles < 256 && PD_F16_WROWS_ROUNDS) {
        float best = 1e9f;
        for (int sp = 1; sp <= 6 && nitems / sp >= PD_F16_WROWS_MIN_ITEMS; ++sp) {
            const float c = (float)((ntiles * sp + 255) / 256) * (0.15f + 1.0f / (float)sp);
            if (c < best - 1e-4f) { best = c; nsplit = sp; }
        }
    }
    hipLaunchKernelGGL(k, dim3((unsigned)(nsplit > 1 ? ntiles * nsplit : (ntiles < 256 ? ntiles : 256))), dim3(64 * NWV), WROWS_LDS_BYTES, s, *p);
    return pd_check_launch();
}

int dispatch_f16_wrows(int op, int pro, int epi, const pd_gemm_args* p, hipStream_t s) {
#ifdef PD_F16_WROWS_12
    // lab: head-norm / plain epilogues with N a multiple of 768 (q | k | v: 1536 = 24 units of 64 columns) on TWELVE waves (three per
    // SIMD) x two 64 x 64 units each.  Measured 88.9 - 91.9 us against 92.6 - 94.9 us for sixteen waves x three 64 x 32 items, bench
    // unchanged: what the larger unit gains per wave (+ 25 % MFMA throughput) the fourth wave per SIMD gave anyway.  Off.
    if ((pro == 1 || pro == 2) && (epi == EPI_HN || epi == EPI_PLAIN) && (op == 1 || p->N % 768 == 0)) {
        int r = PD_ERR_UNSUPPORTED;
        if (pro == 1 && epi == EPI_HN) r = run_f16_wrows<1, EPI_HN, 2, 2, 2, 12>(op, p, s);
        if (pro == 2 && epi == EPI_HN) r = run_f16_wrows<2, EPI_HN, 2, 2, 2, 12>(op, p, s);
        if (pro == 1 && epi == EPI_PLAIN) r = run_f16_wrows<1, EPI_PLAIN, 2, 2, 2, 12>(op, p, s);
        if (pro == 2 && epi == EPI_PLAIN) r = run_f16_wrows<2, EPI_PLAIN, 2, 2, 2, 12>(op, p, s);
        if (op != 1 || r != PD_OK) return r;
    }
#endif
#if PD_F16_WROWS_KS
    if (op == 1 || wrows_ks_shape(p)) {       // 64 x 64 units with a K-split tail (chip-filling launches; set-up covers every form)
        int r = PD_ERR_UNSUPPORTED;
        if (pro == 1 && epi == EPI_HN) r = run_f16_wrows_ks<1, EPI_HN>(op, p, s);
        if (pro == 2 && epi == EPI_HN) r = run_f16_wrows_ks<2, EPI_HN>(op, p, s);
        if (pro == 1 && epi == EPI_PLAIN) r = run_f16_wrows_ks<1, EPI_PLAIN>(op, p, s);
        if (pro == 2 && epi == EPI_PLAIN) r = run_f16_wrows_ks<2, EPI_PLAIN>(op, p, s);
        if (pro == 3 && epi == EPI_GATERES) r = run_f16_wrows_ks<3, EPI_GATERES>(op, p, s);
        if (pro == 3 && epi == EPI_PLAIN) r = run_f16_wrows_ks<3, EPI_PLAIN>(op, p, s);
        if (op != 1 && r != PD_ERR_UNSUPPORTED) return r;
        if (op == 1 && r != PD_OK && r != PD_ERR_UNSUPPORTED) return r;
    }
#endif
    if (pro == 1 && epi == EPI_HN) return run_f16_wrows<1, EPI_HN, 2, 1>(op, p, s);
    if (pro == 2 && epi == EPI_HN) return run_f16_wrows<2, EPI_HN, 2, 1>(op, p, s);
    if (pro == 1 && epi == EPI_PLAIN) return run_f16_wrows<1, EPI_PLAIN, 2, 1>(op, p, s);
    if (pro == 2 && epi == EPI_PLAIN) return run_f16_wrows<2, EPI_PLAIN, 2, 1>(op, p, s);
    if (pro == 3 && epi == EPI_GATERES) return run_f16_wrows<3, EPI_GATERES, 2, 1>(op, p, s);
    if (pro == 3 && epi == EPI_PLAIN) return run_f16_wrows<3, EPI_PLAIN, 2, 1>(op, p, s);
    if (pro == 0 && epi == EPI_GATERES) return run_f16_wrows<0, EPI_GATERES, 2, 1>(op, p, s);
    if (pro == 0 && epi == EPI_PLAIN) return run_f16_wrows<0, EPI_PLAIN, 2, 1>(op, p, s);
#ifdef PD_F16_WROWS_GLU12      // lab: 32 x 64 wave tiles, four-deep fragment ring (163 vs 128 us at 64 samples)
    if (pro == 1 && epi == EPI_GLU) return run_f16_wrows<1, EPI_GLU, 1, 2>(op, p, s);
    if (pro == 2 && epi == EPI_GLU) return run_f16_wrows<2, EPI_GLU, 1, 2>(op, p, s);
#endif
    // SwiGLU: 64 x 64 wave tiles (a W fragment pair feeds two row blocks, an A fragment two column blocks: 24 MFMAs per 16-k step
    // and wave), two-deep fragment ring (64 accumulator registers leave room for no more; four waves per SIMD cover the L2 latency)
    if (pro == 1 && epi == EPI_GLU) return run_f16_wrows<1, EPI_GLU, 2, 2, 2>(op, p, s);
    if (pro == 2 && epi == EPI_GLU) return run_f16_wrows<2, EPI_GLU, 2, 2, 2>(op, p, s);

    return PD_ERR_UNSUPPORTED;
}


// ---- "wide rows, chunked K" kernel: N = 512, K > 512 (the token-level down-projection w2 of a DiT block, K = 1408,
// feed_forward.py:30-31 behind transitions.py:27-30).  A row of K values does not fit LDS, but N / 32 = 16 column blocks are exactly
// the sixteen waves of a block: every wave keeps ONE 64 x 32 accumulator tile for the whole launch and the block walks K in chunks
// of 256 - scaled, split and staged ONCE per 64 rows (the 128 x 128 tile kernel does that once per column tile: four times, with
// a barrier every 32 k), double-buffered in LDS with one block barrier per chunk (96 MFMAs per wave).
constexpr int CLP = 264;                 // LDS row pitch in fp16 of a 256-k chunk (528 bytes = 33 x 16)
constexpr int WCHUNK_LDS_BYTES = 2 * 2 * 64 * CLP * 2;

template <int EPI>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4)))
void gemm_f16_wchunk_kernel(const pd_gemm_args p) {
    constexpr int BM = 64, KCH = 256, TM = 2, TN = 1, PART = BM * CLP, BUF = NPARTS * PART, PF = 3, GK = PF + 1;
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const float a_s = pd_pow2_scale(*p.a_amax);
    const float inv_a_s = 1.0f / a_s;
    const int ntiles = p.M / BM;
    const int nks = p.K >> 4;                                        // 16-k steps (K % 64 == 0: whole groups of four)
    const int nch = (p.K + KCH - 1) / KCH;
    const int wpart = (p.N >> 5) * nks * 1024;                       // bytes per part of the fragment-major weights
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W2), 0, 2 * wpart, 0x00020000);
    const int loff = lane * 16;
    const int sr = tid >> 4, sq = tid & 15;                          // staging: row, 16-byte chunk phase

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * BM;
        const float* xr = p.A + (long long)(row0 + sr) * p.lda;
        f32x4 ra[4];
        auto gload = [&](int c) {                                    // chunk c of the thread's row: columns 256 c + 4 (sq + 16 i)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = KCH * c + 4 * (sq + 16 * i);
                ra[i] = col < p.K ? *reinterpret_cast<const f32x4*>(xr + col) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        auto sstore = [&](int buf) {
            _Float16* base = lds + buf * BUF + sr * CLP;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = 4 * (sq + 16 * i);
                const pd_parts2 p0 = pd_split2h(ra[i][0] * a_s, ra[i][1] * a_s), p1 = pd_split2h(ra[i][2] * a_s, ra[i][3] * a_s);
                *reinterpret_cast<u32x2*>(base + c) = u32x2{p0.h, p1.h};
                *reinterpret_cast<u32x2*>(base + PART + c) = u32x2{p0.l, p1.l};
            }
        };
        f16x8 wf[GK][NPARTS];
        auto wload = [&](int buf, int ks) {                           // this wave's column block = its index
            const int so = (wave * nks + ks) * 1024;
            wf[buf][0] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, loff, so, 0));
            wf[buf][1] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, loff, so + wpart, 0));
        };
        gload(0);
#pragma unroll
        for (int ks = 0; ks < PF; ++ks) wload(ks, ks);
        const int n0 = wave * 32 + l31;
        float c0[TN], c1[TN];
        c0[0] = p.bias ? p.bias[n0] : 0.f;
        c1[0] = 1.f;
        if constexpr (EPI == EPI_GATERES) c1[0] = p.mul ? p.mul[(long long)(row0 / p.mul_rows_per_group) * p.mul_gstride + n0] : 1.f;
        const float cs = p.w_inv[n0] * inv_a_s;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
        sstore(0);
        lds_barrier();
#pragma unroll 1
        for (int c = 0; c < nch; ++c) {
            gload(c + 1);                                            // the next chunk travels under this one's MFMAs (columns >= K: zeros, not stored)
            __builtin_amdgcn_sched_barrier(0);
            const _Float16* abase = lds + (c & 1) * BUF + l31 * CLP + 8 * hh;
            const int ng = (c + 1 < nch ? KCH / 16 : nks - c * (KCH / 16)) / GK;      // groups of four 16-k steps in this chunk
#pragma unroll 1
            for (int gi = 0; gi < ng; ++gi) {
                const int ksl = gi * GK, ksg = c * (KCH / 16) + ksl;                  // k-step inside the chunk / of the launch
#pragma unroll
                for (int jj = 0; jj < GK; ++jj) {
                    wload((jj + PF) % GK, ksg + jj + PF);      // (beyond the last k-step: another block's fragment or, past the range, zeros - never used; no run-time condition around a request)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const f16x8 a0 = *reinterpret_cast<const f16x8*>(abase + 32 * i * CLP + 16 * (ksl + jj));
                        const f16x8 a1 = *reinterpret_cast<const f16x8*>(abase + PART + 32 * i * CLP + 16 * (ksl + jj));
                        f32x16 t = acc[i][0];
                        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, wf[jj][1], t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, wf[jj][0], t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, wf[jj][0], t, 0, 0, 0);
                        acc[i][0] = t;
                    }
                }
            }
            if (c + 1 < nch) sstore((c + 1) & 1);                    // (that buffer was last read in chunk c - 1: every wave is past its barrier)
            lds_barrier();
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][0][r] *= cs;
        epilogue<EPI, TM, TN>(p, acc, c0, c1, row0, wave * 32, 0, 0, l31, hh);
    }
}

template <int EPI>
int run_f16_wchunk(int op, const pd_gemm_args* p, hipStream_t s) {
    auto k = gemm_f16_wchunk_kernel<EPI>;
    if (op == 1)
        return hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, WCHUNK_LDS_BYTES) == hipSuccess
                   ? PD_OK : PD_ERR_LAUNCH;
    const int ntiles = p->M / 64;
    hipLaunchKernelGGL(k, dim3((unsigned)(ntiles < 256 ? ntiles : 256)), dim3(1024), WCHUNK_LDS_BYTES, s, *p);
    return pd_check_launch();
}

int dispatch_f16_wchunk(int op, int epi, const pd_gemm_args* p, hipStream_t s) {
    if (epi == EPI_GATERES) return run_f16_wchunk<EPI_GATERES>(op, p, s);
    if (epi == EPI_PLAIN) return run_f16_wchunk<EPI_PLAIN>(op, p, s);
    return PD_ERR_UNSUPPORTED;
}

#ifndef PD_F16_WCHUNK
#define PD_F16_WCHUNK 1                // lab: 0 = those launches stay on the tile kernel
#endif

#ifndef PD_F16_WROWS_A2
#define PD_F16_WROWS_A2 1              // lab: 0 = pre-split A stays on the tile kernel
#endif
#ifndef PD_F16_WROWS_PLAIN
#define PD_F16_WROWS_PLAIN 1            // lab: 0 = plain fp32 rows (K = 512, few samples) stay on the fp32 streaming kernel
#endif
#ifndef PD_F16_WROWS_PLAIN_MAX_TILES
#define PD_F16_WROWS_PLAIN_MAX_TILES 39  // up to 9 samples of 256 tokens (from 40 tiles on the 64 x 128 tile kernel takes the launch)
#endif
#ifndef PD_F16_WROWS_TINY
#define PD_F16_WROWS_TINY 1             // lab: 0 = launches below the tile kernels' thresholds never reach the wide-rows kernels
#endif
#ifndef PD_F16_WROWS_MIN_TILES_SPLIT
#define PD_F16_WROWS_MIN_TILES_SPLIT 4    // norm-prologue forms (blocks may share a row tile): from one sample of 256 tokens on
#endif
#ifndef PD_F16_WROWS_MIN_TILES
#define PD_F16_WROWS_MIN_TILES 128     // 64-row tiles, one block per CU: from half the chip on (32 samples of 256 tokens: 72 -> 66 us + the split pass; 48: 105 -> 74; 20: 49 -> 61, stays on the tile kernel)
#endif

template <int PRO, int EPI, int BM>
int run_f16_rows(int op, const pd_gemm_args* p, hipStream_t s) {
    auto k = gemm_f16_rows_kernel<PRO, EPI, BM>;
    constexpr int lds = 2 * BM * RLP * 2;
    if (op == 1)
        return hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess
                   ? PD_OK : PD_ERR_LAUNCH;
    const int ntiles = p->M / BM, grid = 512 * (128 / BM);
    hipLaunchKernelGGL(k, dim3((unsigned)(ntiles < grid ? ntiles : grid)), dim3(4 * BM), lds, s, *p);
    return pd_check_launch();
}

// big: 128-row tiles (op 1 = raise the LDS limit of both instances)
int dispatch_f16_rows(int op, int pro, int epi, const pd_gemm_args* p, hipStream_t s, bool big) {
#define PD_RCASE(P, E) if (pro == P && epi == E) { \
        if (op == 1) { const int r = run_f16_rows<P, E, 64>(1, p, s); return r != PD_OK ? r : run_f16_rows<P, E, 128>(1, p, s); } \
        return big ? run_f16_rows<P, E, 128>(op, p, s) : run_f16_rows<P, E, 64>(op, p, s); }
    PD_RCASE(1, EPI_HN) PD_RCASE(2, EPI_HN) PD_RCASE(1, EPI_PLAIN) PD_RCASE(2, EPI_PLAIN)
#undef PD_RCASE
    return PD_ERR_UNSUPPORTED;
}

#ifndef PD_F16_ROWS_MIN_TILES
#define PD_F16_ROWS_MIN_TILES 512      // 128-row tiles from here on (two per block slot) ...
#endif
#ifndef PD_F16_ROWS_GIVEN_STATS
#define PD_F16_ROWS_GIVEN_STATS 1      // also for launches that bring their statistics (lab: 0 = those stay on the tile kernel)
#endif
#ifndef PD_F16_ROWS_MIN_TILES64
#define PD_F16_ROWS_MIN_TILES64 320    // ... else 64-row tiles if there are this many (10 samples of 2048 atoms)
#endif

using F128 = FTile<128, 128, 2, 8, true>;    // 2 x 4 waves of 64 x 32, direct W
using F128G = FTile<128, 128, 4, 8, false>;  // 4 x 2 waves of 32 x 64 (GLU: a wave owns both columns of a pair), W through LDS
#ifdef PD_F16_GLU_LDSW
using F128GD = F128G;
#else
// ... direct W: with two parts and a pre-split A the two B fragment buffers fit (124 VGPRs, no spill; with the in-kernel norm
// prologue they spill 4-8 registers, so those cases keep W in LDS): token SwiGLU projection 196 -> 190 us
using F128GD = FTile<128, 128, 4, 8, true>;
#endif

// 64 x 128 tiles on four waves (1 x 4, 64 x 32 each: the register profile of the 128 x 128 tile), four blocks per CU: launches whose 128 x 128 tile count leaves CUs idle (the
// token projections at ~8-30 samples: w2 / linear_o at 20 samples = 160 tiles of 128 x 128, 320 of these)
using F64 = FTile<64, 128, 1, 4, true>;

// lab: 256 x 128 tiles, 2 x 4 waves of 128 x 32 (each W fragment feeds four row blocks: half the W requests per MFMA)
#ifndef PD_F16_T256_BPC
#define PD_F16_T256_BPC 1
#endif
using F256 = FTile<256, 128, 2, 8, true, PD_F16_T256_BPC>;

int dispatch_f16(int op, int pro, int epi, const pd_gemm_args* p, hipStream_t s, bool small) {
#define PD_FCASE(P, E, TL) if (pro == P && epi == E) return run_f16<P, E, TL>(op, p, s);
#ifdef PD_F16_T256
    if (!small && (op == 1 || p->M % 256 == 0)) {
        PD_FCASE(0, EPI_PLAIN, F256) PD_FCASE(3, EPI_PLAIN, F256)
    }
#endif
    if (small) {
        PD_FCASE(0, EPI_PLAIN, F64) PD_FCASE(1, EPI_PLAIN, F64) PD_FCASE(3, EPI_PLAIN, F64)
        PD_FCASE(1, EPI_HN, F64) PD_FCASE(2, EPI_HN, F64) PD_FCASE(3, EPI_HN, F64)
        PD_FCASE(0, EPI_GATERES, F64) PD_FCASE(3, EPI_GATERES, F64) PD_FCASE(0, EPI_TGATERES, F64)
        return PD_ERR_UNSUPPORTED;
    }
    PD_FCASE(0, EPI_PLAIN, F128) PD_FCASE(1, EPI_PLAIN, F128) PD_FCASE(3, EPI_PLAIN, F128)
    PD_FCASE(1, EPI_HN, F128) PD_FCASE(2, EPI_HN, F128) PD_FCASE(3, EPI_HN, F128)
    PD_FCASE(1, EPI_GLU, F128G) PD_FCASE(2, EPI_GLU, F128G) PD_FCASE(3, EPI_GLU, F128GD) PD_FCASE(1, EPI_GLUT, F128G)
    PD_FCASE(0, EPI_GATERES, F128) PD_FCASE(3, EPI_GATERES, F128) PD_FCASE(0, EPI_TGATERES, F128)
#undef PD_FCASE
    return PD_ERR_UNSUPPORTED;
}

}  // namespace

// Same contract as pd_gemm_split_try (gemm_split.hip); needs the fp16-split weights with their row scales (args->W2, w_inv) and
// the magnitude bound of A (args->a_amax).  init_only: 0 launch, 1 raise the LDS limits, 2 query (returns the EPI kind).
extern "C" int pd_gemm_f16_try(const pd_gemm_args* args, int pro, int tile, void* stream, int init_only) {
    if (init_only == 1) {
        int rc = PD_OK;
        for (int P = 0; P < 4; ++P)
            for (int E = 0; E < 6; ++E) {
                for (int small = 0; small < 2; ++small) {
                    const int r = dispatch_f16(1, P, E, nullptr, nullptr, small != 0);
                    if (r != PD_OK && r != PD_ERR_UNSUPPORTED) rc = r;
                }
                const int r = dispatch_f16_rows(1, P, E, nullptr, nullptr, true);
                if (r != PD_OK && r != PD_ERR_UNSUPPORTED) rc = r;
                const int rw = dispatch_f16_wrows(1, P, E, nullptr, nullptr);
                if (rw != PD_OK && rw != PD_ERR_UNSUPPORTED) rc = rw;
                const int rc2 = P == 0 ? dispatch_f16_wchunk(1, E, nullptr, nullptr) : PD_ERR_UNSUPPORTED;
                if (rc2 != PD_OK && rc2 != PD_ERR_UNSUPPORTED) rc = rc2;
            }
        return rc;
    }
    const pd_gemm_args& p = *args;
    if (!p.W2 || !p.w_inv || !p.a_amax || p.K % 4 != 0 || (tile != 128 && tile != 64)) return PD_ERR_UNSUPPORTED;
    if (p.A2) {                                  // pre-split A: whole 32-k slices, 16-byte aligned, prologue and scale already applied
        if (pro != 0 || p.pro_act != PD_ACT_NONE || p.K % 32 != 0 || ((uintptr_t)p.A2 & 15)) return PD_ERR_UNSUPPORTED;
        pro = 3;
    }
    // transposed GLU output (the triangle update's gated q | k projections): the rule of gemm_split.hip
    const bool glut = p.out_mode == PD_OUT_TRANSPOSED && p.glu && !p.hn_w && !p.mul && !p.res && !p.act && !p.rowscale_acc &&
                      !p.maskadd && p.out_scale == 1.f && p.vecY && (!p.rowscale || ((uintptr_t)p.rowscale & 15) == 0) && !p.A2 && !p.Y2;
    if (p.a_kmajor || p.w_kmajor || !p.vecA || p.batch != 1 || (p.out_mode != PD_OUT_ROWMAJOR && !glut)) return PD_ERR_UNSUPPORTED;
    if (((uintptr_t)p.W2 & 15) != 0 || p.N % 128 != 0 || p.M % 64 != 0) return PD_ERR_UNSUPPORTED;
    // 128 x 128 tiles when they fill the chip, else 64 x 128 tiles if there are enough of THOSE; smaller launches are latency-bound
    // (k-split fp32 kernel).  The caller's tile (its 64 x 64 choice below 192 row/column blocks) only says the rows come in 64s.
    const long long t128 = p.M % 128 == 0 ? (long long)(p.M / 128) * (p.N / 128) : 0, t64 = (long long)(p.M / 64) * (p.N / 128);
    // (the tile kernels' own threshold; the wide-rows kernels below share a row tile among several blocks and take fewer: PD_F16_TINY)
    // (long K - the token w2, K = 1408 - is worth the 64 x 128 tile from 96 tiles on: 31 us at 10 samples of 256 tokens against 39 us for the
    //  K-split fp32 pair at 6 - 7 samples and 49 us for the bf16 x 6 64 x 64 tile that took 8 - 9; tools/w2_paths.py, round 5)
    const bool tiny = t128 < PD_F16_MIN_TILES && t64 < (p.K >= 1024 ? PD_F16_MIN_TILES_SMALL_LONGK : PD_F16_MIN_TILES_SMALL);
    // plain fp32 rows with K = 512 at a handful of samples (linear_o behind an attention that writes fp32: one launch whose blocks share
    // the row tiles, instead of the K-split pair of the fp32 streaming kernel: 7 + 7 us)
    const bool plain512 = PD_F16_WROWS_PLAIN && p.K == 512 && !p.A2 && pro == 0 && !p.stats && !p.stats_inline && p.pro_act == PD_ACT_NONE &&
                          p.M / 64 >= PD_F16_WROWS_MIN_TILES_SPLIT && p.M / 64 <= PD_F16_WROWS_PLAIN_MAX_TILES && t128 < PD_F16_MIN_TILES;
    if (tiny && !(PD_F16_WROWS_TINY && p.K == 512 && !p.A2 && !p.stats && (p.stats_inline || plain512))) return PD_ERR_UNSUPPORTED;
    const bool small = t128 < PD_F16_MIN_TILES;
    if (p.rowscale_acc || (p.rowscale && !glut) || p.maskadd || p.out_scale != 1.f) return PD_ERR_UNSUPPORTED;
    int epi;
    if (glut) epi = small ? -1 : EPI_GLUT;
    else if (p.glu) epi = (p.hn_w || p.mul || p.res || p.act) ? -1 : EPI_GLU;
    else if (p.hn_w) epi = (p.mul || p.res || p.act) ? -1 : EPI_HN;
    else if (p.res) {
        epi = (p.mul && p.mul_rows_per_group <= 0) ? EPI_TGATERES : EPI_GATERES;      // gate: one row per group, or a tensor (ldmul)
        if (p.act || p.res_row_mod > 0) epi = -1;
        if (p.mul && p.mul_rows_per_group > 0 && p.mul_rows_per_group % 64 != 0) epi = -1;
    } else epi = p.mul ? -1 : EPI_PLAIN;
    if (epi < 0) return PD_ERR_UNSUPPORTED;
    if (p.Y2) {          // k | v written pre-split for the attention kernel: head-norm epilogue only, whole 32-column heads
        if (epi != EPI_HN || !p.y2_amax || p.hn_split <= 0 || p.hn_split % 32 != 0 || p.y2_col0 % 32 != 0 ||
            p.ldy2 < 2 * (p.N - p.y2_col0) || p.ldy2 % 8 != 0 || ((uintptr_t)p.Y2 & 15))
            return PD_ERR_UNSUPPORTED;
    }
    // row statistics computed by the kernel itself (pd_gemm_args.stats_inline): the rows kernel (K = 128, whole 128-row tiles,
    // enough of them for two blocks per CU), or nothing
    // ... or that the caller supplies (K = 128 with a norm prologue: the rows kernel reads the rows once for all column tiles)
    const bool rows_big = p.M % 128 == 0 && p.M / 128 >= PD_F16_ROWS_MIN_TILES;
    const bool rows_ok = p.K == 128 && !p.A2 && (pro == 1 || pro == 2) && p.pro_act == PD_ACT_NONE && p.act == PD_ACT_NONE &&
                         (rows_big || p.M / 64 >= PD_F16_ROWS_MIN_TILES64) && (epi == EPI_HN || epi == EPI_PLAIN) &&
                         (p.stats || p.stats_inline == 1 || p.stats_inline == 2) && p.lda % 4 == 0 && ((uintptr_t)p.A & 15) == 0 &&
                         (((uintptr_t)p.pro_w | (uintptr_t)p.pro_b) & 15) == 0 && p.pro_gstride % 4 == 0 &&
                         (pro != 2 || p.pro_rows_per_group > 0) && (long long)(p.N / 32) * 8 * 1024 * 2 < 0x7fffffffll;
    // K = 512: the wide-rows kernel (64-row tiles on sixteen waves; only with inline statistics: callers that pre-split A keep that path)
    const bool wrows_ok = p.K == 512 && !p.A2 && !p.stats && (pro == 1 || pro == 2) && p.pro_act == PD_ACT_NONE && p.act == PD_ACT_NONE &&
                          p.M % 64 == 0 && p.M / 64 >= PD_F16_WROWS_MIN_TILES_SPLIT && (epi == EPI_HN || epi == EPI_PLAIN || (epi == EPI_GLU && p.N % 64 == 0)) &&
                          (p.stats_inline == 1 || p.stats_inline == 2) && p.lda % 4 == 0 && ((uintptr_t)p.A & 15) == 0 &&
                          (((uintptr_t)p.pro_w | (uintptr_t)p.pro_b) & 15) == 0 && p.pro_gstride % 4 == 0 &&
                          (pro != 2 || p.pro_rows_per_group > 0) && (long long)(p.N / 32) * 32 * 1024 * 2 < 0x7fffffffll;
    // ... or with A already split (K = 512: linear_o behind the attention kernel's split output), plain / gate + residual epilogue
    const bool wrows_a2 = PD_F16_WROWS_A2 && p.K == 512 && p.A2 && pro == 3 && p.act == PD_ACT_NONE && p.M / 64 >= PD_F16_WROWS_MIN_TILES && p.M % 64 == 0 &&
                          (epi == EPI_PLAIN || (epi == EPI_GATERES && (!p.mul || p.mul_rows_per_group % 64 == 0))) &&
                          (long long)(p.N / 32) * 32 * 1024 * 2 < 0x7fffffffll;
    // N = 512 with a long K and fp32 rows (the token w2): one accumulator tile per wave, K in double-buffered chunks
    const bool wchunk_ok = PD_F16_WCHUNK && p.N == 512 && p.K > 512 && p.K % 64 == 0 && !p.A2 && !p.stats && !p.stats_inline && pro == 0 &&
                           p.pro_act == PD_ACT_NONE && p.act == PD_ACT_NONE && p.M % 64 == 0 && p.M / 64 >= PD_F16_WROWS_MIN_TILES &&
                           p.lda % 4 == 0 && ((uintptr_t)p.A & 15) == 0 &&
                           (epi == EPI_PLAIN || (epi == EPI_GATERES && (!p.mul || p.mul_rows_per_group % 64 == 0))) &&
                           (long long)16 * (p.K / 16) * 1024 * 2 < 0x7fffffffll;
    if (wchunk_ok) {
        if (init_only == 2) {
            const int r = dispatch_f16_wchunk(1, epi, nullptr, nullptr);
            return r == PD_OK ? epi + 0x600 : r;          // tile code 6: the chunked wide-rows kernel
        }
        return dispatch_f16_wchunk(0, epi, &p, (hipStream_t)stream);
    }
    const bool wrows_plain = plain512 && p.act == PD_ACT_NONE && p.M % 64 == 0 && p.lda % 4 == 0 && ((uintptr_t)p.A & 15) == 0 &&
                             (epi == EPI_PLAIN || (epi == EPI_GATERES && (!p.mul || p.mul_rows_per_group % 64 == 0))) &&
                             (long long)(p.N / 32) * 32 * 1024 * 2 < 0x7fffffffll;
    if (wrows_ok || wrows_a2 || wrows_plain) {
        if (init_only == 2) {
            const int r = dispatch_f16_wrows(1, pro, epi, nullptr, nullptr);
            return r == PD_OK ? epi + 0x500 : r;          // tile code 5: the wide-rows kernel
        }
        return dispatch_f16_wrows(0, pro, epi, &p, (hipStream_t)stream);
    }
    if (tiny) return PD_ERR_UNSUPPORTED;
    if ((!p.stats && p.stats_inline) || (rows_ok && PD_F16_ROWS_GIVEN_STATS)) {
        const bool big = rows_big;
        if (!rows_ok) return PD_ERR_UNSUPPORTED;
        if (init_only == 2) {
            const int r = dispatch_f16_rows(1, pro, epi, nullptr, nullptr, big);
            return r == PD_OK ? epi + (big ? 0x300 : 0x400) : r;          // tile codes 3 / 4: the rows kernel on 128- / 64-row tiles
        }
        return dispatch_f16_rows(0, pro, epi, &p, (hipStream_t)stream, big);
    }
    if (init_only == 2) {        // query: the EPI kind, + 0x100 when the launch takes the 64 x 128 tile
        const int r = dispatch_f16(1, pro, epi, nullptr, nullptr, small);
        return r == PD_OK ? epi + (small ? 0x100 : 0) : r;
    }
    return dispatch_f16(0, pro, epi, &p, (hipStream_t)stream, small);
}
