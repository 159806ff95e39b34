// fp32-accurate GEMM on the bf16 matrix pipe: error-free 3-way operand splitting ("bf16 x 6").
//
// Every fp32 operand is split into three bf16 values a = a_h + a_m + a_l (exact for normal fp32: 3 x 8 significand
// bits + the residual signs cover the 24-bit significand); a product a.b is then the sum of the partial products of the
// parts, each EXACT in fp32 (8 x 8 significand bits).  Dropping the three terms below 2^-24 |a b| (m.l, l.m, l.l) leaves six
// v_mfma_f32_32x32x16_bf16 per 32 x 32 x 16 block, accumulated in fp32 by the matrix pipe.  Measured on MI355X
// (tools/micro/bf16x3_probe.hip, K = 512, wide dynamic range): max error / sum|a b| = 7.3e-7 against 1.09e-6 for
// v_mfma_f32_32x32x2_f32 (which is bitwise an fmaf chain: one rounding per product; the bf16 pipe rounds once per 16
// products), rms 7.8e-8 against 1.0e-7 - i.e. at least fp32-MFMA accuracy, and nine products add nothing measurable.
// Rate: six bf16 MFMAs of 32 cycles replace eight fp32 MFMAs of 64 cycles per 32 x 32 x 16 block: 2.67 x the fp32
// matrix peak (2.5 PF / 6 = 419 TF effective).  The reference computes these contractions with fp32 F.linear on CPU
// and with TF32 on its own GPUs (models/model.py:5); parity is held to the same 1e-3 A bar as before.
//
// Structure = gemm_stream.hip (persistent blocks, XCD-aware tile order, norm prologue applied while staging A,
// compile-time specialised epilogues straight from the accumulator fragments - the C fragment layout of the bf16 MFMA is
// the same 32 x 32 layout, so gemm_tile_common.h is shared).  What differs:
//   * W arrives PRE-SPLIT and FRAGMENT-MAJOR (packing.split3_bf16: [3][N/32][Kp/16][64 lanes][8] bf16, Kp = K rounded up to
//     32, zero padded) - weights are constants.  In the "direct W" tiles (struct STile, DW) a wave fetches its B fragments
//     with one coalesced 16-byte load per lane straight into MFMA registers, one k-step ahead; W never touches LDS;
//   * A is split while it is staged: global fp32 -> registers -> (norm prologue) -> 3 x bf16 -> LDS, two values per packed
//     conversion;
//   * DW tiles: an LDS stage holds a whole 32-k slice of A (80-byte rows: conflict-free ds_read_b128 fragments), two stages,
//     ONE block barrier per slice.  The GLU tile (two B fragments per wave: no register room for direct buffers) keeps the
//     first version of the loop: A and W through LDS in 16-k stages (48-byte rows), two barriers per slice;
//   * the next slice is requested from global memory before the MFMA block and written to the other stage after it
//     (LDS-only barriers: no vmcnt drain).
#include <stdlib.h>
#include "gemm_tile_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef PD_SPLIT_MIN_TILES
#define PD_SPLIT_MIN_TILES 256
#endif
#ifndef PD_SPLIT_MIN_TILES_SMALL
#define PD_SPLIT_MIN_TILES_SMALL 256
#endif
#if defined(PD_ABL) && PD_ABL == 6
#define NPARTS_IS_3 0
constexpr int NPARTS = 2;            // ablation: two operand parts / three products (timing estimate of a 2-part format; wrong results)
#else
#define NPARTS_IS_3 1
constexpr int NPARTS = 3;
#endif
constexpr int KS = 16;               // k per LDS stage = one v_mfma_f32_32x32x16_bf16 step
constexpr int PITCH = 24;            // bf16 per LDS row (48 bytes)
constexpr int PITCH2 = 40;           // DW tiles: 32 k per row, 80 bytes apart (20 r mod 64 hits 16 distinct 4-bank groups: conflict-free b128 reads)

// Block tile BM x BN computed by NWAVES waves laid out WM x WN.  The hot 128 x 128 tile runs on EIGHT waves (64 x 32 per
// wave; 32 x 64 for GLU epilogues, where a wave must own both columns of a pair): with two blocks per CU that is four
// waves per SIMD at <= 128 VGPRs, so the matrix pipe always has a wave with MFMAs to issue while others stage operands -
// the six short bf16 MFMAs per block leave 2.67 x less matrix time per staged byte than the fp32 kernel has to hide it in.
// DW ("direct W"): the pre-split weight fragments do not pass through LDS at all.  packing.split3_bf16 stores them
// FRAGMENT-MAJOR - the 64 lanes' 16-byte operands of one (32-row block, 16-k step) are one contiguous 1 KB block - so a
// wave fetches its B operand with one fully coalesced global_load_dwordx4 per part, one k-step ahead of its use.  LDS then
// carries only the A tile (which needs the norm prologue and the split, done once for the four waves that share it): per
// block and k-step 12 KB of stores + 48 KB of reads instead of 24 + 72, the LDS port being what bounded the loop.
template <int BM_, int BN_, int WM_, int NWAVES_, bool DW_ = false>
struct STile {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = NWAVES_ / WM_, NT = 64 * NWAVES_;
    static constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    static constexpr bool DW = DW_;
    // DW: a stage is a whole 32-k slice of A (rows PITCH2 apart) -> one block barrier per slice instead of two
    static constexpr int STAGE = DW ? 3 * BM * PITCH2 : 3 * (BM + BN) * PITCH;      // bf16 elements per stage
    static constexpr int LDS_BYTES = 2 * STAGE * 2;
    static constexpr int BLOCKS_PER_CU = NWAVES_ == 8 ? 2 : LDS_BYTES > 60000 ? 2 : LDS_BYTES > 40000 ? 3 : 4;
    static constexpr int WAVES_PER_SIMD = NWAVES_ * BLOCKS_PER_CU / 4;
    static constexpr int GRID = 256 * BLOCKS_PER_CU;
};


__device__ __forceinline__ void lds_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0); outstanding global loads stay in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// a = h + m + l with bf16 parts (round-to-nearest-even conversions; the residuals are exact in fp32)
__device__ __forceinline__ void split4(const f32x4& v, bf16x4& h, bf16x4& m, bf16x4& l) {
    const pd_parts p0 = pd_split2(v[0], v[1]), p1 = pd_split2(v[2], v[3]);       // two values per packed conversion
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    h = __builtin_bit_cast(bf16x4, (u32x2){p0.h, p1.h});
    m = __builtin_bit_cast(bf16x4, (u32x2){p0.m, p1.m});
    l = __builtin_bit_cast(bf16x4, (u32x2){p0.l, p1.l});
}

template <int PRO, int EPI, class TL>
__global__ __launch_bounds__(TL::NT) __attribute__((amdgpu_waves_per_eu(TL::WAVES_PER_SIMD, TL::WAVES_PER_SIMD)))
void gemm_split_kernel(const pd_gemm_args p) {
    constexpr int BM = TL::BM, BN = TL::BN, TM = TL::TM, TN = TL::TN;
    constexpr int XSLOTS = TL::GRID / 8;
    constexpr int SNT = TL::NT;
    constexpr int TPR_A = SNT / BM;              // threads per A row (2 or 4); a row of a 32-k slice = 8 f32x4 chunks
    constexpr int CPH_A = 4 / TPR_A;             // chunks per thread per 16-k half
    constexpr int TPR_W = SNT / BN;              // threads per W row; a row of a 32-k slice of one part = 4 bf16x8 chunks
    constexpr int NW = 4 / TPR_W;                // W chunks per thread per part per slice (2: one per half; 1: one half only)
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / TL::WN, wn = wave % TL::WN;
    const int l31 = lane & 31, hh = lane >> 5;
    const int nMb = p.M / BM, nNb = p.N / BN;
    const int ntiles = nMb * nNb;
    constexpr bool DW = TL::DW;
    const int nk = (p.K + 31) / 32;
    const int nks = 2 * nk;                       // 16-k steps per (zero padded) weight row
    // pre-split weights, fragment-major: [3 parts][N / 32 row blocks][nks k-steps][64 lanes][8 bf16], lane = 32 * (k / 8 & 1) + n % 32
    const bf16x8* __restrict__ W3 = reinterpret_cast<const bf16x8*>(p.W3);
    const long long wpart = (long long)((p.N + 31) / 32) * nks * 64;          // in 16-byte units
    auto w3_chunk = [&](int part, int n, int kchunk) {                          // kchunk = k / 8
        return W3 + part * wpart + ((long long)(n >> 5) * nks + (kchunk >> 1)) * 64 + (kchunk & 1) * 32 + (n & 31);
    };

    // stage s: [3 parts][BM rows][PITCH] for A, then [3][BN][PITCH] for W
    auto sA = [&](int s, int part) { return lds + s * TL::STAGE + part * BM * PITCH; };
    auto sW = [&](int s, int part) { return lds + s * TL::STAGE + 3 * BM * PITCH + part * BN * PITCH; };

    const int a_row = tid / TPR_A, a_q = tid % TPR_A;
    const int w_row = tid / TPR_W, w_q = tid % TPR_W;
    // PRO == 3: A arrives pre-split (pd_norm_split, [3][M][Kp] bf16): a thread copies one 16-byte chunk (8 k) per part
    constexpr bool AS = PRO == 3;
    static_assert(!AS || TPR_A == 4, "pre-split A: four 8-k chunks per row slice");
    const __bf16* __restrict__ A3 = reinterpret_cast<const __bf16*>(p.A3);
    const long long apart = (long long)p.M * (nk * 32);
    bf16x8 ra3[3];
    f32x4 ra[2][CPH_A];                          // [half][i]: chunk 4*half + a_q + TPR_A*i of the thread's row
    bf16x8 rw[3][NW];                            // [part][i]: NW == 2 -> chunk w_q + 2 i (half i); NW == 1 -> chunk w_q (half w_q >> 1)

    auto gload = [&](int bm0, int bn0, int k0) {
#if defined(PD_ABL) && (PD_ABL == 1 || PD_ABL == 5)
        if (k0 != 0) return;                      // ablation: no A traffic inside a tile
#endif
        const int r = bm0 + a_row;                // full tiles only: always < M
        if constexpr (AS) {
            const __bf16* ap3 = A3 + (long long)r * (nk * 32) + k0 + 8 * a_q;
#pragma unroll
            for (int part = 0; part < NPARTS; ++part) ra3[part] = *reinterpret_cast<const bf16x8*>(ap3 + part * apart);
        }
        const float* ap = p.A + (long long)r * p.lda + k0;
#pragma unroll
        for (int h = 0; h < (AS ? 0 : 2); ++h)
#pragma unroll
            for (int i = 0; i < CPH_A; ++i) {
                int kc = 16 * h + 4 * (a_q + TPR_A * i);
                kc = k0 + kc < p.K ? kc : 0;      // clamped address; zeroed in `stage` (K % 4 == 0 is required)
                ra[h][i] = *reinterpret_cast<const f32x4*>(ap + kc);
            }
        if constexpr (!DW) {
#pragma unroll
            for (int part = 0; part < NPARTS; ++part)
#pragma unroll
                for (int i = 0; i < NW; ++i)
                    rw[part][i] = *w3_chunk(part, bn0 + w_row, (k0 >> 3) + (NW == 2 ? w_q + 2 * i : w_q));
        }
    };
    // DW: this wave's B fragments of 16-k step `ks` of column block bn0, straight into MFMA operand registers
    bf16x8 wf[2][TN][3];
    auto wfrag = [&](int buf, int bn0, int ks) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#if defined(PD_ABL) && (PD_ABL == 2 || PD_ABL == 5)
            ks = 0;                               // ablation: B operands always the same (L1-resident) block
#endif
            const bf16x8* base = W3 + ((long long)((bn0 + wn * (32 * TN) + j * 32) >> 5) * nks + ks) * 64 + lane;
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
        float c0[TN], c1[TN];
        const int gate_row = bm0 + wm * (32 * TM);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            c0[j] = p.bias ? p.bias[n0 + 32 * j] : 0.f;
            c1[j] = 1.f;
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
        // norm prologue + k-tail zeroing + split + LDS store of one 16-k half of the slice held in ra / rw
        auto stage = [&](int h, int k0) {
            if constexpr (AS) {
                if ((a_q >> 1) == h) {
#pragma unroll
                    for (int part = 0; part < NPARTS; ++part)
                        *reinterpret_cast<bf16x8*>(sA(h, part) + a_row * PITCH + 8 * (a_q & 1)) = ra3[part];
                }
            }
#pragma unroll
            for (int i = 0; i < (AS ? 0 : CPH_A); ++i) {
                const int c = a_q + TPR_A * i;                    // chunk inside the half
                const int kc = k0 + 16 * h + 4 * c;
                f32x4 v = ra[h][i];
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
                if (kc >= p.K) v = f32x4{0.f, 0.f, 0.f, 0.f};
                bf16x4 ph, pm, pl;
                split4(v, ph, pm, pl);
                const int o = a_row * PITCH + 4 * c;
                *reinterpret_cast<bf16x4*>(sA(h, 0) + o) = ph;
                *reinterpret_cast<bf16x4*>(sA(h, 1) + o) = pm;
                if (NPARTS == 3) *reinterpret_cast<bf16x4*>(sA(h, 2) + o) = pl;
            }
            if (!DW && (NW == 2 || (w_q >> 1) == h)) {
                const int i = NW == 2 ? h : 0;
                const int c = NW == 2 ? w_q : (w_q & 1);          // 16-byte chunk inside the half
#pragma unroll
                for (int part = 0; part < NPARTS; ++part)
                    *reinterpret_cast<bf16x8*>(sW(h, part) + w_row * PITCH + 8 * c) = rw[part][i];
            }
        };
        // 6 partial products per (i, j) fragment pair of one 16-k stage
        auto mma = [&](int s) {
            bf16x8 fa[TM][3], fw[TN][3];
#pragma unroll
            for (int part = 0; part < NPARTS; ++part) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[i][part] = *reinterpret_cast<const bf16x8*>(sA(s, part) + (wm * (32 * TM) + i * 32 + l31) * PITCH + 8 * hh);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (DW) fw[j][part] = wf[s][j][part];
                    else fw[j][part] = *reinterpret_cast<const bf16x8*>(sW(s, part) + (wn * (32 * TN) + j * 32 + l31) * PITCH + 8 * hh);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x16 c = acc[i][j];
#if NPARTS_IS_3
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fw[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fw[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fw[j][1], c, 0, 0, 0);
#endif
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fw[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fw[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fw[j][0], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        };
        // ---- DW variants: the whole 32-k slice in ra goes to LDS stage s; fragments of k-step ks come from there and from wf[ks]
        auto stage2 = [&](int s, int k0) {
            __bf16* base = lds + s * TL::STAGE + a_row * PITCH2;
            if constexpr (AS) {
#pragma unroll
                for (int part = 0; part < NPARTS; ++part) *reinterpret_cast<bf16x8*>(base + part * BM * PITCH2 + 8 * a_q) = ra3[part];
            }
#pragma unroll
            for (int h = 0; h < (AS ? 0 : 2); ++h)
#pragma unroll
                for (int i = 0; i < CPH_A; ++i) {
                    const int c = a_q + TPR_A * i;
                    const int kc = k0 + 16 * h + 4 * c;
                    f32x4 v = ra[h][i];
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
                    if (kc >= p.K) v = f32x4{0.f, 0.f, 0.f, 0.f};
                    bf16x4 ph, pm, pl;
#if defined(PD_ABL) && (PD_ABL == 3 || PD_ABL == 5)
                    ph = pm = pl = __builtin_bit_cast(bf16x4, f32x2{ra[h][i][0], ra[h][i][1]});      // ablation: no prologue / split VALU
#else
                    split4(v, ph, pm, pl);
#endif
                    const int o = 16 * h + 4 * c;
                    *reinterpret_cast<bf16x4*>(base + o) = ph;
                    *reinterpret_cast<bf16x4*>(base + BM * PITCH2 + o) = pm;
                    if (NPARTS == 3) *reinterpret_cast<bf16x4*>(base + 2 * BM * PITCH2 + o) = pl;
                }
        };
        auto mma2 = [&](int s, int ks) {
            bf16x8 fa[TM][3];
            const __bf16* base = lds + s * TL::STAGE + (wm * (32 * TM) + l31) * PITCH2 + 16 * ks + 8 * hh;
#pragma unroll
            for (int part = 0; part < NPARTS; ++part)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[i][part] = *reinterpret_cast<const bf16x8*>(base + part * BM * PITCH2 + i * 32 * PITCH2);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x16 c = acc[i][j];
#if NPARTS_IS_3
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], wf[ks][j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], wf[ks][j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], wf[ks][j][1], c, 0, 0, 0);
#endif
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], wf[ks][j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], wf[ks][j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], wf[ks][j][0], c, 0, 0, 0);
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
            stage2(0, 0);
            lds_barrier();
            for (int kt = 0; kt < nk; ++kt) {
                const int st = kt & 1;
                const bool more = kt + 1 < nk;
                if (more) gload(bm0, bn0, (kt + 1) * 32);
                mma2(st, 0);
                if (more) wfrag(0, bn0, 2 * kt + 2);          // every B buffer is re-requested right after its last use:
                mma2(st, 1);                                  // a full k-step (+ the staging and the barrier) ahead
                if (more) {
                    wfrag(1, bn0, 2 * kt + 3);
                    // the other stage was last read in the previous iteration, which every wave left through its barrier
                    stage2(st ^ 1, (kt + 1) * 32);
#if !defined(PD_ABL) || (PD_ABL != 4 && PD_ABL != 5)
                    lds_barrier();
#endif
                }
            }
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
                stage(1, (kt + 1) * 32);          // (visible to the next mma(1) through the barrier after the next mma(0))
            }
        }
        }
        const int cur_bm0 = bm0, cur_bn0 = bn0;
        if (tile + t_step < t_end) {
            coords(tile + t_step, bm0, bn0);
            gload(bm0, bn0, 0);
            if constexpr (DW) { wfrag(0, bn0, 0); wfrag(1, bn0, 1); }
        }
        epilogue<EPI, TM, TN>(p, acc, c0, c1, cur_bm0, cur_bn0, wm, wn, l31, hh);
    }
}

template <int PRO, int EPI, class TL>
int run_split(int op, const pd_gemm_args* p, hipStream_t s) {
    auto k = gemm_split_kernel<PRO, EPI, TL>;
    constexpr int lds = TL::LDS_BYTES;
    if (op == 1)
        return hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess
                   ? PD_OK : PD_ERR_LAUNCH;
    const long long ntiles = (long long)(p->M / TL::BM) * (p->N / TL::BN);
    hipLaunchKernelGGL(k, dim3((unsigned)(ntiles < TL::GRID ? ntiles : TL::GRID)), dim3(TL::NT), lds, s, *p);
    return pd_check_launch();
}

using S128 = STile<128, 128, 2, 8, true>;  // 2 x 4 waves of 64 x 32
using S128G = STile<128, 128, 4, 8, false>; // 4 x 2 waves of 32 x 64 (GLU): two B fragments per wave do not fit the register budget directly (the 4-wave 64 x 64 direct layout: same GEMM rate, -2 % end to end)
using S128W4 = STile<128, 128, 2, 4>;      // 2 x 2 waves of 64 x 64
using S64 = STile<64, 64, 2, 4, true>;
using S12864 = STile<128, 64, 4, 4, true>;

int dispatch_split(int op, int pro, int epi, int tile, const pd_gemm_args* p, hipStream_t s) {
#define PD_SCASE(P, E, C, TL) if (pro == P && epi == E && tile == C) return run_split<P, E, TL>(op, p, s);
    PD_SCASE(0, EPI_PLAIN, 128, S128) PD_SCASE(1, EPI_PLAIN, 128, S128)
    PD_SCASE(1, EPI_HN, 128, S128) PD_SCASE(2, EPI_HN, 128, S128)
    PD_SCASE(1, EPI_GLU, 128, S128G) PD_SCASE(2, EPI_GLU, 128, S128G) PD_SCASE(1, EPI_GLUT, 128, S128G)
    PD_SCASE(0, EPI_GATERES, 128, S128) PD_SCASE(0, EPI_TGATERES, 128, S128)
    PD_SCASE(3, EPI_PLAIN, 128, S128) PD_SCASE(3, EPI_HN, 128, S128) PD_SCASE(3, EPI_GLU, 128, S128G)      // pre-split A
    PD_SCASE(0, EPI_PLAIN, 64, S64) PD_SCASE(1, EPI_PLAIN, 64, S64)
    PD_SCASE(1, EPI_HN, 64, S64) PD_SCASE(2, EPI_HN, 64, S64)
    PD_SCASE(0, EPI_GATERES, 64, S64) PD_SCASE(0, EPI_TGATERES, 64, S64)
    PD_SCASE(1, EPI_GLU, 12864, S12864) PD_SCASE(2, EPI_GLU, 12864, S12864)
#ifdef PD_LAB      // experiments: 32 x 64 wave tiles with a plain epilogue; the 4-wave 64 x 64 layout with a GLU epilogue
    PD_SCASE(1, EPI_PLAIN, 1284, S128G) PD_SCASE(1, EPI_GLU, 1282, S128W4)
#endif
#undef PD_SCASE
    return PD_ERR_UNSUPPORTED;
}

}  // namespace


// Same contract as pd_gemm_stream_try (gemm_stream.hip); additionally needs the pre-split weights (args->W3) and
// 16-byte aligned A rows with K % 4 == 0.  init_only: 0 launch, 1 raise the LDS limits, 2 query (returns the EPI kind).
extern "C" int pd_gemm_split_try(const pd_gemm_args* args, int pro, int tile, void* stream, int init_only) {
    if (init_only == 1) {
        int rc = PD_OK;
        for (int T : {128, 64, 12864})
            for (int P = 0; P < 4; ++P)
                for (int E = 0; E < 6; ++E) {
                    const int r = dispatch_split(1, P, E, T, nullptr, nullptr);
                    if (r != PD_OK && r != PD_ERR_UNSUPPORTED) rc = r;
                }
        return rc;
    }
    const pd_gemm_args& p = *args;
    if (!p.W3 || p.K % 4 != 0) return PD_ERR_UNSUPPORTED;
    if (p.A3) {                                  // pre-split A: whole 32-k slices, 16-byte aligned, prologue already applied
        if (pro != 0 || p.pro_act != PD_ACT_NONE || p.K % 32 != 0 || ((uintptr_t)p.A3 & 15) || tile != 128) return PD_ERR_UNSUPPORTED;
        pro = 3;
    }
#ifdef PD_LAB
    if (const char* f = getenv("PD_SPLIT_TILE")) { if (tile == 128) tile = atoi(f); }
    if (tile == 1284 || tile == 1282) {
        if (init_only == 2) return p.glu ? EPI_GLU : EPI_PLAIN;
        return dispatch_split(0, pro, p.glu ? EPI_GLU : EPI_PLAIN, tile, &p, (hipStream_t)stream);
    }
#endif
    if (tile != 128 && tile != 64 && tile != 12864) return PD_ERR_UNSUPPORTED;
    const int tbm = tile == 64 ? 64 : 128, tbn = tile == 128 ? 128 : 64;
    const bool glut = p.out_mode == PD_OUT_TRANSPOSED && p.glu && !p.hn_w && !p.mul && !p.res && !p.act && !p.rowscale_acc &&
                      !p.maskadd && p.out_scale == 1.f && p.vecY && (!p.rowscale || ((uintptr_t)p.rowscale & 15) == 0) && tile == 128;
    if (p.a_kmajor || p.w_kmajor || !p.vecA || p.batch != 1 || (p.out_mode != PD_OUT_ROWMAJOR && !glut)) return PD_ERR_UNSUPPORTED;
    if (((uintptr_t)p.W3 & 15) != 0 || p.N % 32 != 0) return PD_ERR_UNSUPPORTED;
    if (p.M % tbm != 0 || p.N % tbn != 0) return PD_ERR_UNSUPPORTED;
    // Launches that do not fill the chip are latency-bound (two block barriers per 32-k slice here, one in gemm_stream.hip):
    // measured at 1-4 samples the fp32 kernel is faster on every DiT shape, from ~256 tiles on the split kernel wins.
    if ((long long)(p.M / tbm) * (p.N / tbn) < (tile == 128 ? PD_SPLIT_MIN_TILES : PD_SPLIT_MIN_TILES_SMALL)) return PD_ERR_UNSUPPORTED;
    if (p.rowscale_acc || (p.rowscale && !glut) || p.maskadd || p.out_scale != 1.f) return PD_ERR_UNSUPPORTED;
    int epi;
    if (glut) epi = EPI_GLUT;
    else if (p.glu) epi = (p.hn_w || p.mul || p.res || p.act) ? -1 : EPI_GLU;
    else if (p.hn_w) epi = (p.mul || p.res || p.act) ? -1 : EPI_HN;
    else if (p.res) {
        epi = (p.mul && p.mul_rows_per_group <= 0) ? EPI_TGATERES : EPI_GATERES;
        if (p.act || p.res_row_mod > 0) epi = -1;
        if (p.mul && p.mul_rows_per_group > 0 && p.mul_rows_per_group % 64 != 0) epi = -1;
    } else epi = p.mul ? -1 : EPI_PLAIN;
    if (epi < 0) return PD_ERR_UNSUPPORTED;
    if (init_only == 2) {
        const int r = dispatch_split(1, pro, epi, tile, nullptr, nullptr);
        return r == PD_OK ? epi : r;
    }
    return dispatch_split(0, pro, epi, tile, &p, (hipStream_t)stream);
}
