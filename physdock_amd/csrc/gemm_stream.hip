// Persistent fp32 MFMA GEMM with a direct-from-fragment, compile-time specialised epilogue (the variant of gemm.hip
// for the hot full-tile row-major shapes).
//
// Why: measurements on gemm.hip (NOTES.md "epilogue finding, revisited") show that its flag-driven LDS-staged epilogue
// costs ~20 % of a K=128 tile and ~7 % of a K=512 tile, and that this cost is NOT the stores (suppressing only the
// stores changes the time by 6 %): it is the instruction / latency chain park -> barrier -> re-read -> flag tests ->
// 64-bit address arithmetic, repeated per 16-byte chunk.  Here
//   * the epilogue kind is a template parameter and tiles are always full, so an element costs 2-6 instructions,
//   * values go straight from the MFMA accumulator fragment to memory (lane = column: every store instruction writes
//     two full 128-byte row segments), no LDS round trip, no barrier,
//   * SGPR base pointers + 32-bit lane offsets,
//   * a block walks several output tiles; the first k-slice of the next tile and the per-column constants are
//     requested before the epilogue of the current one, so their latency hides under it.
//   * the persistent grid walks the tiles in XCD-aware patches (struct TileOrder) so that A / W panels are shared
//     through each XCD's L2 instead of being re-fetched per column block.
// A register-parked DEFERRED epilogue (issued inside the next tile's main loop) was built first: it needs 64 more
// registers than the 256 available at two waves per SIMD and the compiler spills (NOTES.md).
//
// Scope: A [M,K] and W [N,K] row-major with 16-byte aligned rows, row-major Y, one batch, M % 128 == N % 128 == 0
// (pd_gemm peels a ragged row remainder off to gemm.hip), epilogue kinds below.  Everything else stays on gemm.hip (pd_gemm dispatches).
#include <stdlib.h>
#include "common.h"
#include "physdock_hip.h"

#ifdef PD_STREAM_SAMETILE
#define PD_LT(x) ((x) & 0)          // experiment: every tile re-reads tile 0 (all loads hit L2 / TLB)
#else
#define PD_LT(x) (x)
#endif

#include "gemm_tile_common.h"

namespace {

#ifndef PD_KSPLIT_MAX_BYTES
#define PD_KSPLIT_MAX_BYTES (12 << 20)
#endif
constexpr int BK = 32, LDK = BK + 4, NT = 256;

// Block tile BM x BN, 4 waves laid out WM x WN, each wave TM x TN MFMA fragments of 32 x 32:
//   Tile<128,128,2> 2 x 2 waves of 64 x 64     the hot shapes
//   Tile<64,64,2>   2 x 2 waves of 32 x 32     problems that do not fill the chip with 128 x 128 tiles
//   Tile<128,64,4>  4 x 1 waves of 32 x 64     the same for GLU epilogues (a wave must own both columns of a pair)
template <int BM_, int BN_, int WM_>
struct Tile {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = 4 / WM_;
    static constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    static constexpr int A_TILE = BM * LDK, W_TILE = BN * LDK;
    static constexpr int LDS_BYTES = 4 * 2 * (A_TILE + W_TILE);
    static constexpr int BLOCKS_PER_CU = LDS_BYTES > 60000 ? 2 : LDS_BYTES > 40000 ? 3 : 4;     // 73.7 / 55.3 / 36.9 KB of LDS
    static constexpr int GRID = 256 * BLOCKS_PER_CU;                // persistent grid
};

template <int ROWS>
struct Loader {     // ROWS rows x 32 k, [row][k] layout; thread -> rows (tid>>3)+32i, 16-byte chunk tid&7
    static constexpr int NP = ROWS / 32;
    f32x4 reg[NP];
    __device__ __forceinline__ void load(const float* __restrict__ base, int ld, int r0, int rows, int k0, int K, int tid) {
        const int kc = k0 + (tid & 7) * 4;
        const int kcl = kc < K ? kc : 0;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int r = r0 + (tid >> 3) + 32 * i;
            r = r < rows ? r : rows - 1;
            reg[i] = *reinterpret_cast<const f32x4*>(base + (long long)r * ld + kcl);
        }
    }
    __device__ __forceinline__ void mask(int r0, int rows, int k0, int K, int tid) {
        if (r0 + ROWS <= rows && k0 + BK <= K) return;
        const int kc = k0 + (tid & 7) * 4;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const bool rok = r0 + (tid >> 3) + 32 * i < rows;
#pragma unroll
            for (int e = 0; e < 4; ++e) reg[i][e] = (rok && kc + e < K) ? reg[i][e] : 0.f;
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ s, int tid) const {
#pragma unroll
        for (int i = 0; i < NP; ++i) *reinterpret_cast<f32x4*>(s + ((tid >> 3) + 32 * i) * LDK + (tid & 7) * 4) = reg[i];
    }
};

template <int PRO, int EPI, class TL>
__global__ __launch_bounds__(NT, TL::BLOCKS_PER_CU) void gemm_stream_kernel(const pd_gemm_args p) {
    constexpr int BM = TL::BM, BN = TL::BN, TM = TL::TM, TN = TL::TN;
    constexpr int A_TILE = TL::A_TILE, W_TILE = TL::W_TILE;
    constexpr int XSLOTS = TL::GRID / 8;               // resident blocks per XCD
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;
    float* sW = smem + 2 * A_TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / TL::WN, wn = wave % TL::WN;
    const int l31 = lane & 31, hh = lane >> 5;
    const int nMb = p.M / BM, nNb = p.N / BN;             // full tiles only (launcher)
    const int ntiles = nMb * nNb;
    const int nk = (p.K + BK - 1) / BK;

    Loader<BM> la;
    Loader<BN> lw;
    f32x16 acc[TM][TN];

    // tile sequence of this block: XCD-aware patches for the persistent grid, plain M-fastest order otherwise
    const bool grouped = gridDim.x == TL::GRID && nMb >= 8;
    TileOrder ord;
    ord.init(nMb, nNb, grouped ? blockIdx.x & 7 : 0, grouped ? 8 : 1, XSLOTS);
    int t_step = grouped ? XSLOTS : gridDim.x;
    const int t_end = grouped ? ord.ntiles : ntiles;
    int tile = grouped ? blockIdx.x >> 3 : blockIdx.x;
    // K-split launches (few tiles, long K): one (tile, k-part) per block, all parts of a tile on one XCD (block b -> XCD
    // b & 7) so that the partial sums meet in that XCD's L2:  b = xcd + 8 (q ks + part),  tile = 8 q + xcd
    // Two launches: ksplit = ks > 1: every block computes one k-part of one tile and stores its partial accumulators in the
    // scratch; ksplit = -ks: one block per tile adds the parts IN ORDER (bit-reproducible, no atomics, the kernel boundary is
    // the only synchronisation) and runs the epilogue.
    const int ks = p.ksplit > 1 ? p.ksplit : (p.ksplit < -1 ? -p.ksplit : 1);
    const bool reduce_only = p.ksplit < -1;
    int kt_beg = 0, kt_end = nk, part = 0;
    if (ks > 1) {
        if (reduce_only) {
            tile = blockIdx.x;
            kt_end = 0;
        } else {
            const int rest = blockIdx.x >> 3;
            part = rest % ks;
            tile = (rest / ks) * 8 + (blockIdx.x & 7);
            const int nkp = (nk + ks - 1) / ks;
            kt_beg = part * nkp;
            kt_end = kt_beg + nkp < nk ? kt_beg + nkp : nk;
        }
        t_step = 1 << 30;                                   // a single pass through the tile loop
    }
    if (tile >= t_end) return;
    auto coords = [&](int t, int& bm0, int& bn0) {
        int mb, nb;
        if (grouped) ord.get(t, mb, nb);
        else { mb = t % nMb; nb = t / nMb; }
        bm0 = mb * BM; bn0 = nb * BN;
    };
    int bm0, bn0;
    coords(tile, bm0, bn0);
    if (!reduce_only) {
        la.load(p.A, p.lda, PD_LT(bm0), p.M, kt_beg * BK, p.K, tid);
        lw.load(p.W, p.ldw, PD_LT(bn0), p.N, kt_beg * BK, p.K, tid);
    }

    for (; tile < t_end; tile += t_step) {
        // per-lane column constants of this tile (column group j = packed columns n0 + 32 j); consumed in the epilogue
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
        // per-tile prologue state (rows this thread stages)
        constexpr int NPA = Loader<BM>::NP;
        float st_mean[NPA], st_rstd[NPA];
        int grp_off[NPA];
#pragma unroll
        for (int i = 0; i < NPA; ++i) { st_mean[i] = 0.f; st_rstd[i] = 1.f; grp_off[i] = 0; }
        if constexpr (PRO != 0) {
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                const int m = bm0 + (tid >> 3) + 32 * i;
                if (p.stats) {
                    st_mean[i] = p.stats[2 * (long long)m];
                    st_rstd[i] = p.stats[2 * (long long)m + 1];
                } else if (!reduce_only) {
                    // pd_gemm_args.stats_inline: the eight threads that stage row m compute its statistics themselves (the row is K
                    // floats, L2-resident in the launches that come here) - mean first, then centred squares, as pd_rowstats does
                    const float* rowp = p.A + (long long)m * p.lda + (tid & 7) * 4;
                    float s1 = 0.f;
                    if (p.stats_inline == 2) {
                        for (int kc = (tid & 7) * 4; kc < p.K; kc += 32) {
                            const f32x4 v = *reinterpret_cast<const f32x4*>(rowp + (kc - (tid & 7) * 4));
                            s1 += (v[0] + v[1]) + (v[2] + v[3]);
                        }
                        s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2); s1 += __shfl_xor(s1, 4);
                    }
                    const float mean = p.stats_inline == 2 ? s1 / (float)p.K : 0.f;
                    float q = 0.f;
                    for (int kc = (tid & 7) * 4; kc < p.K; kc += 32) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(rowp + (kc - (tid & 7) * 4));
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q += d * d; }
                    }
                    q += __shfl_xor(q, 1); q += __shfl_xor(q, 2); q += __shfl_xor(q, 4);
                    st_mean[i] = mean;
                    st_rstd[i] = rsqrtf(q / (float)p.K + p.stats_eps);
                }
                if constexpr (PRO == 2) grp_off[i] = (m / p.pro_rows_per_group) * p.pro_gstride;
            }
        }
        auto transform_A = [&](int k0) {
            if constexpr (PRO != 0) {
                int kc = k0 + (tid & 7) * 4;
                kc = kc < p.K ? kc : 0;
                f32x4 pw, pb;
                if constexpr (PRO == 1) {
                    pw = *reinterpret_cast<const f32x4*>(p.pro_w + kc);
                    pb = *reinterpret_cast<const f32x4*>(p.pro_b + kc);
                }
#pragma unroll
                for (int i = 0; i < NPA; ++i) {
                    if constexpr (PRO == 2) {
                        pw = *reinterpret_cast<const f32x4*>(p.pro_w + grp_off[i] + kc);
                        pb = *reinterpret_cast<const f32x4*>(p.pro_b + grp_off[i] + kc);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) la.reg[i][e] = (la.reg[i][e] - st_mean[i]) * st_rstd[i] * pw[e] + pb[e];
                }
            }
            if (p.pro_act == PD_ACT_RELU) {
#pragma unroll
                for (int i = 0; i < NPA; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) la.reg[i][e] = fmaxf(la.reg[i][e], 0.f);
            } else if (p.pro_act == PD_ACT_SILU) {
#pragma unroll
                for (int i = 0; i < NPA; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) la.reg[i][e] = pd_silu(la.reg[i][e]);
            }
        };
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // slice 0 of this tile is already in la/lw (requested before the previous tile's epilogue)
        if (!reduce_only) {
            transform_A(kt_beg * BK);
            la.mask(bm0, p.M, kt_beg * BK, p.K, tid);
            lw.mask(bn0, p.N, kt_beg * BK, p.K, tid);
            __syncthreads();                     // previous tile's last slice has been read by every wave
            la.store(sA, tid);
            lw.store(sW, tid);
            __syncthreads();
        }

        for (int kt = kt_beg; kt < kt_end; ++kt) {
            const int cur = (kt - kt_beg) & 1;
            const bool more = kt + 1 < kt_end;
            if (more) {
                la.load(p.A, p.lda, PD_LT(bm0), p.M, (kt + 1) * BK, p.K, tid);
                lw.load(p.W, p.ldw, PD_LT(bn0), p.N, (kt + 1) * BK, p.K, tid);
            }
            const float* a = sA + cur * A_TILE;
            const float* w = sW + cur * W_TILE;
#pragma unroll
            for (int g = 0; g < BK / 8; ++g) {
                f32x4 fa[TM], fw[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(a + (wm * (32 * TM) + i * 32 + l31) * LDK + g * 8 + 4 * hh);
#pragma unroll
                for (int j = 0; j < TN; ++j) fw[j] = *reinterpret_cast<const f32x4*>(w + (wn * (32 * TN) + j * 32 + l31) * LDK + g * 8 + 4 * hh);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fw[j][e], acc[i][j], 0, 0, 0);
            }
            if (more) {
                transform_A((kt + 1) * BK);
                la.mask(bm0, p.M, (kt + 1) * BK, p.K, tid);
                lw.mask(bn0, p.N, (kt + 1) * BK, p.K, tid);
                la.store(sA + (cur ^ 1) * A_TILE, tid);
                lw.store(sW + (cur ^ 1) * W_TILE, tid);
                __syncthreads();
            }
        }
        // request the next tile's first slice: its latency hides under this tile's epilogue
        const int cur_bm0 = bm0, cur_bn0 = bn0;
        if (tile + t_step < t_end) {
            coords(tile + t_step, bm0, bn0);
            la.load(p.A, p.lda, PD_LT(bm0), p.M, 0, p.K, tid);
            lw.load(p.W, p.ldw, PD_LT(bn0), p.N, 0, p.K, tid);
        }

        if (ks > 1) {
            float* __restrict__ parts = reinterpret_cast<float*>(p.ksplit_ws) + (long long)tile * ks * (BM * BN);
            if (!reduce_only) {
                float* mine = parts + (long long)part * (BM * BN);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *reinterpret_cast<f32x4*>(mine + (((i * TN + j) * 4 + q) * NT + tid) * 4) =
                                f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                return;
            }
            for (int pp = 0; pp < ks; ++pp) {
                const float* src = parts + (long long)pp * (BM * BN);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 v = *reinterpret_cast<const f32x4*>(src + (((i * TN + j) * 4 + q) * NT + tid) * 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += v[e];
                        }
            }
        }
        epilogue<EPI, TM, TN>(p, acc, c0, c1, cur_bm0, cur_bn0, wm, wn, l31, hh);
    }
}

}  // namespace

// (PRO, EPI, tile) instantiations: op 0 launch, 1 raise the dynamic-LDS limit
template <int PRO, int EPI, class TL>
static int run_stream(int op, const pd_gemm_args* p, hipStream_t s) {
    auto k = gemm_stream_kernel<PRO, EPI, TL>;
    constexpr int lds = TL::LDS_BYTES;
    if (op == 1)
        return hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess
                   ? PD_OK : PD_ERR_LAUNCH;
    const long long ntiles = (long long)(p->M / TL::BM) * (p->N / TL::BN);
    if (p->ksplit > 1) {
        hipLaunchKernelGGL(k, dim3((unsigned)(((ntiles + 7) / 8) * 8 * p->ksplit)), dim3(NT), lds, s, *p);
        pd_gemm_args r = *p;
        r.ksplit = -p->ksplit;
        hipLaunchKernelGGL(k, dim3((unsigned)ntiles), dim3(NT), lds, s, r);
        return pd_check_launch();
    }
    hipLaunchKernelGGL(k, dim3((unsigned)(ntiles < TL::GRID ? ntiles : TL::GRID)), dim3(NT), lds, s, *p);
    return pd_check_launch();
}

// tile codes of the C interface between gemm.hip and this file
using T128 = Tile<128, 128, 2>;
using T64 = Tile<64, 64, 2>;
using T12864 = Tile<128, 64, 4>;
static void tile_dims(int tile, int& bm, int& bn) { bm = tile == 64 ? 64 : 128; bn = tile == 128 ? 128 : 64; }

static int dispatch_stream(int op, int pro, int epi, int tile, const pd_gemm_args* p, hipStream_t s) {
#define PD_SCASE(P, E, C, TL) if (pro == P && epi == E && tile == C) return run_stream<P, E, TL>(op, p, s);
    PD_SCASE(0, EPI_PLAIN, 128, T128) PD_SCASE(1, EPI_PLAIN, 128, T128)
    PD_SCASE(1, EPI_HN, 128, T128) PD_SCASE(2, EPI_HN, 128, T128)
    PD_SCASE(1, EPI_GLU, 128, T128) PD_SCASE(2, EPI_GLU, 128, T128) PD_SCASE(1, EPI_GLUT, 128, T128)
    PD_SCASE(0, EPI_GATERES, 128, T128) PD_SCASE(0, EPI_TGATERES, 128, T128)
    // smaller tiles for problems that do not fill the chip with 128 x 128 ones (few samples, trunk side tracks)
    PD_SCASE(0, EPI_PLAIN, 64, T64) PD_SCASE(1, EPI_PLAIN, 64, T64)
    PD_SCASE(1, EPI_HN, 64, T64) PD_SCASE(2, EPI_HN, 64, T64)
    PD_SCASE(0, EPI_GATERES, 64, T64) PD_SCASE(0, EPI_TGATERES, 64, T64)
    PD_SCASE(1, EPI_GLU, 12864, T12864) PD_SCASE(2, EPI_GLU, 12864, T12864)
#undef PD_SCASE
    return PD_ERR_UNSUPPORTED;
}

// tile: 128, 64 or 12864 = 128 x 64 (the block tile pd_gemm's heuristic picked).  init_only: 0 launch, 1 raise the LDS limits (pd_init),
// 2 query only (returns the EPI_* kind).  Returns PD_ERR_UNSUPPORTED when the arguments are outside this file's scope
// (pd_gemm then uses gemm.hip).
extern "C" int pd_gemm_stream_try(const pd_gemm_args* args, int pro, int tile, void* stream, int init_only) {
    if (init_only == 1) {
        int rc = PD_OK;
        for (int T : {128, 64, 12864})
            for (int P = 0; P < 3; ++P)
                for (int E = 0; E < 6; ++E) {
                    const int r = dispatch_stream(1, P, E, T, nullptr, nullptr);
                    if (r != PD_OK && r != PD_ERR_UNSUPPORTED) rc = r;
                }
        return rc;
    }
    const pd_gemm_args& p = *args;
    if (tile != 128 && tile != 64 && tile != 12864) return PD_ERR_UNSUPPORTED;
    int tbm, tbn;
    tile_dims(tile, tbm, tbn);
    const bool glut = p.out_mode == PD_OUT_TRANSPOSED && p.glu && !p.hn_w && !p.mul && !p.res && !p.act && !p.rowscale_acc &&
                      !p.maskadd && p.out_scale == 1.f && p.vecY && (!p.rowscale || ((uintptr_t)p.rowscale & 15) == 0) && tile == 128;
    if (p.a_kmajor || p.w_kmajor || !p.vecA || !p.vecW || p.batch != 1 || (p.out_mode != PD_OUT_ROWMAJOR && !glut)) return PD_ERR_UNSUPPORTED;
    if (p.M % tbm != 0 || p.N % tbn != 0) return PD_ERR_UNSUPPORTED;        // full tiles only
    if (!p.stats && p.stats_inline && (pro == 0 || p.K % 32 != 0 || (p.stats_inline != 1 && p.stats_inline != 2))) return PD_ERR_UNSUPPORTED;
    if (p.rowscale_acc || (p.rowscale && !glut) || p.maskadd || p.out_scale != 1.f) return PD_ERR_UNSUPPORTED;
    int epi;
    if (glut) epi = EPI_GLUT;
    else if (p.glu) epi = (p.hn_w || p.mul || p.res || p.act) ? -1 : EPI_GLU;
    else if (p.hn_w) epi = (p.mul || p.res || p.act) ? -1 : EPI_HN;
    else if (p.res) {
        epi = (p.mul && p.mul_rows_per_group <= 0) ? EPI_TGATERES : EPI_GATERES;
        if (p.act || p.res_row_mod > 0) epi = -1;
        if (p.mul && p.mul_rows_per_group > 0 && p.mul_rows_per_group % 64 != 0) epi = -1;   // one gate row per wave
    } else epi = p.mul ? -1 : EPI_PLAIN;
    if (epi < 0) return PD_ERR_UNSUPPORTED;
    if (init_only == 2) {                                     // query: epilogue kind (>= 0) of the instantiation
        const int r = dispatch_stream(1, pro, epi, tile, nullptr, nullptr);
        return r == PD_OK ? epi : r;
    }
    // K-split for launches that leave most of the chip idle: parts of >= 4 slices, up to ~512 blocks
    pd_gemm_args q = p;
    q.ksplit = 1;
    if (p.ksplit_ws && ((uintptr_t)p.ksplit_ws & 15) == 0) {
        const long long ntiles = (long long)(p.M / tbm) * (p.N / tbn);
        const int nk = (p.K + BK - 1) / BK;
        long long ks = nk / 4;
        if (ks > 8) ks = 8;
        if (ks > 512 / ntiles) ks = 512 / ntiles;
        while (ks >= 2 && ntiles * ks * tbm * tbn * 4 > PD_KSPLIT_MAX_BYTES) --ks;      // the partial sums are written and read back: keep them L2-sized
        const long long need = ntiles * ks * tbm * tbn * 4;
        if (ks >= 2 && ntiles < 192 && need <= p.ksplit_ws_bytes) q.ksplit = (int)ks;
    }
    return dispatch_stream(0, pro, epi, tile, &q, (hipStream_t)stream);
}
