// Shared by gemm_stream.hip (fp32 MFMA main loop) and gemm_split.hip (3 x bf16 split-operand main loop): both leave the
// block tile in the SAME accumulator fragments (32 x 32 C tiles: lane = column, register r = row (r&3)+8(r>>2)+4*half), so
// the compile-time specialised epilogues and the XCD-aware tile order are common code.
#pragma once
#include "common.h"
#include "physdock_hip.h"

#ifdef PD_STREAM_NOEMIT
#define PD_ST(dst, val) do { const float v_ = (val); if (v_ == 123.456f) (dst) = v_; } while (0)   // experiment: no stores
#else
#define PD_ST(dst, val) (dst) = (val)
#endif

namespace {

// Epilogue kinds (compile-time, so that the epilogue is a few straight-line instructions per element; a flag-driven
// fragment epilogue cost ~75 instructions per element = 12 us per 128x128 tile).
//   PLAIN   : Y = act(acc + bias)
//   HN      : Y = headnorm(acc + bias) on columns < hn_cols (q | k), plain beyond (v)        [no act]
//   GLU     : Y = silu(a + ba) * (b + bb)  |  (a + ba) * sigmoid(b + bb)   on packed column pairs
//   GATERES : Y = (acc + bias) * gate[row group] + res          (gate optional; res may alias Y)
//   TGATERES: Y = (acc + bias) * gate[row, col] + res           (gate tensor, e.g. the sigmoid gate of an attention)
//   GLUT    : GLU as above, times rowscale[row], stored TRANSPOSED  Y[col][row]   (the channel-major q | k operands of the
//             triangle multiplication, attentions.py:161-162: four consecutive rows of one column = one 16-byte store)
enum { EPI_PLAIN = 0, EPI_HN = 1, EPI_GLU = 2, EPI_GATERES = 3, EPI_TGATERES = 4, EPI_GLUT = 5 };

// Epilogue of one block tile straight from the accumulator fragments: lane = column, register r = row
// (r&3)+8(r>>2)+4*half.  Tiles are always full (the launcher peels ragged rows off to gemm.hip).  Addresses are
// (uniform row pointer)[lane offset]: SGPR base + one shared 32-bit VGPR offset per array, nothing per row in VGPRs.
template <int EPI, int TM, int TN>
__device__ __forceinline__ void epilogue(const pd_gemm_args& p, const f32x16 (&acc)[TM][TN], const float (&c0)[TN],
                                         const float (&c1)[TN], int bm0, int bn0, int wm, int wn, int l31, int hh) {
    static_assert((EPI != EPI_GLU && EPI != EPI_GLUT) || TN == 2, "a GLU pair needs both column fragments in one wave");
    const int ldy = p.ldy, ldres = p.ldres, ldmul = p.ldmul;
    const int yoff = hh * 4 * ldy + l31;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mb = bm0 + wm * (32 * TM) + i * 32;
        if constexpr (EPI == EPI_GLUT) {
            // output column n = packed column pair index; rows of register group g: mb + 8 g + 4 hh + (0..3)
            float* __restrict__ Yo = p.Y + (long long)(((bn0 + wn * (32 * TN)) >> 1) + l31) * ldy + mb + 4 * hh;
            const float* __restrict__ rs = p.rowscale + mb + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 sc = p.rowscale ? *reinterpret_cast<const f32x4*>(rs + 8 * g) : f32x4{1.f, 1.f, 1.f, 1.f};
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = acc[i][0][4 * g + e] + c0[0], b = acc[i][TN - 1][4 * g + e] + c0[TN - 1];
                    o[e] = (p.glu == 1 ? pd_silu_r(a) * b : a * pd_sigmoid_r(b)) * sc[e];
                }
                *reinterpret_cast<f32x4*>(Yo + 8 * g) = o;
            }
        } else if constexpr (EPI == EPI_GLU) {
            float* __restrict__ Yo = p.Y + (long long)mb * ldy + ((bn0 + wn * (32 * TN)) >> 1);
            if (p.glu == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    PD_ST((Yo + pd_frag_row(r, 0) * ldy)[yoff], pd_silu_r(acc[i][0][r] + c0[0]) * (acc[i][TN - 1][r] + c0[TN - 1]));
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    PD_ST((Yo + pd_frag_row(r, 0) * ldy)[yoff], (acc[i][0][r] + c0[0]) * pd_sigmoid_r(acc[i][TN - 1][r] + c0[TN - 1]));
            }
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int ncol0 = bn0 + wn * (32 * TN) + j * 32;
                float* __restrict__ Yo = p.Y + (long long)mb * ldy + ncol0;
                if constexpr (EPI == EPI_GATERES || EPI == EPI_TGATERES) {
                    const float* __restrict__ Ro = p.res + (long long)mb * ldres + ncol0;
                    const int roff = hh * 4 * ldres + l31;
                    float gv[16];
                    if constexpr (EPI == EPI_TGATERES) {       // gate pass first: 16 loads in flight, not 32
                        const float* __restrict__ Go = p.mul + (long long)mb * ldmul + ncol0;
                        const int goff = hh * 4 * ldmul + l31;
#pragma unroll
                        for (int r = 0; r < 16; ++r) gv[r] = (Go + pd_frag_row(r, 0) * ldmul)[goff];
#pragma unroll
                        for (int r = 0; r < 16; ++r) gv[r] = (acc[i][j][r] + c0[j]) * gv[r];
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) gv[r] = (acc[i][j][r] + c0[j]) * c1[j];
                    }
                    float rv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[r] = (Ro + pd_frag_row(r, 0) * ldres)[roff];
#pragma unroll
                    for (int r = 0; r < 16; ++r) PD_ST((Yo + pd_frag_row(r, 0) * ldy)[yoff], gv[r] + rv[r]);
                    __builtin_amdgcn_sched_barrier(0);      // keep the 16/32 loads of one fragment from piling up with the next
                } else if constexpr (EPI == EPI_HN) {
                    // pd_gemm_args.Y2: this fragment (one head's 32 columns of k or v) goes to the attention kernel already scaled
                    // and split into two fp16 parts instead of to Y (the launcher admits Y2 only on the fp16-format kernel)
#ifdef PD_EPILOGUE_Y2                 // defined by gemm_f16.hip only: the other kernel families never get a launch with Y2
                    const bool to_y2 = p.Y2 != nullptr && ncol0 >= p.y2_col0;
#else
                    constexpr bool to_y2 = false;
#endif
                    float y2s = 1.f;
                    unsigned short* __restrict__ Y2o = nullptr;
                    if (to_y2) {
                        y2s = pd_pow2_scale(p.y2_amax[(ncol0 - p.y2_col0) / p.hn_split]);
                        // row layout: groups of four columns as (4 high parts, 4 low parts): column c = 4 g + e -> 8 g + e, low + 4
                        Y2o = reinterpret_cast<unsigned short*>(p.Y2) + (long long)(mb + hh * 4) * p.ldy2 + 2 * (ncol0 - p.y2_col0) +
                              8 * (l31 >> 2) + (l31 & 2);          // (the lane pair's even column: 32-bit stores, see below)
                    }
                    float ov[16];
                    if (ncol0 < p.hn_cols) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float v = acc[i][j][r] + c0[j];
                            const float ss = pd_half_sum32(v * v);        // a head = the 32 lanes of a wave half
                            ov[r] = v * rsqrtf(ss * (1.0f / 32.0f) + p.hn_eps) * c1[j];
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) ov[r] = acc[i][j][r] + c0[j];
                    }
                    if (to_y2) {
                        // two rows per split: a lane holds (row r, row r + 1) of ITS column as packed pairs; one exchange with the
                        // neighbouring column's lane (quad_perm 1,0,3,2) turns that into (column e, column e + 1) pairs - the even
                        // lane keeps row r, the odd lane row r + 1 - so every lane issues two 32-bit stores per row pair (high
                        // parts, low parts) instead of four 16-bit ones
                        const bool odd = (l31 & 1) != 0;
                        const unsigned sel = odd ? 0x07060302u : 0x01000504u;       // v_perm_b32(own, neighbour, sel)
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {
                            const pd_parts2 q2 = pd_split2h(ov[r] * y2s, ov[r + 1] * y2s);
                            const unsigned nh = (unsigned)__builtin_amdgcn_mov_dpp((int)q2.h, 0xB1, 0xf, 0xf, true);
                            const unsigned nl = (unsigned)__builtin_amdgcn_mov_dpp((int)q2.l, 0xB1, 0xf, 0xf, true);
                            const unsigned dh = __builtin_amdgcn_perm(q2.h, nh, sel), dl = __builtin_amdgcn_perm(q2.l, nl, sel);
                            unsigned* d = reinterpret_cast<unsigned*>(Y2o + (long long)pd_frag_row(odd ? r + 1 : r, 0) * p.ldy2);
                            d[0] = dh;
                            d[2] = dl;                                        // low parts: four fp16 elements further
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) PD_ST((Yo + pd_frag_row(r, 0) * ldy)[yoff], ov[r]);
                    }
                } else {
                    if (p.act == PD_ACT_SILU) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            PD_ST((Yo + pd_frag_row(r, 0) * ldy)[yoff], pd_silu_r(acc[i][j][r] + c0[j]));
                    } else if (p.act == PD_ACT_NONE) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            PD_ST((Yo + pd_frag_row(r, 0) * ldy)[yoff], acc[i][j][r] + c0[j]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            PD_ST((Yo + pd_frag_row(r, 0) * ldy)[yoff], pd_act(acc[i][j][r] + c0[j], p.act));
                    }
                }
            }
        }
    }
}

// XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8), each with its own 4 MB L2.
// A persistent grid of 512 (1024 for 64x64 tiles) blocks gives every XCD 64 (128) concurrently running tiles; they are chosen
// as a compact PATCH of gm row blocks x gn column blocks inside a contiguous range of row blocks owned by that XCD, so that
// an A panel fetched by one tile is an L2 hit for the gn tiles beside it and a W panel for the gm tiles below it.  Traffic
// per XCD ~ tiles x (W-panel bytes / gm + A-panel bytes / gn): a square patch (8 x 8) minimises it.  (M-fastest order
// re-fetches A once per column block: measured 8x the algorithmic reads for N = 2816; full-width patches of 2 row blocks
// streamed the 8.6 MB of split weights of that shape once per patch: 445 MB per launch against 134 MB algorithmic.)
struct TileOrder {
    int nMb, nNb, mb_lo, nmb, gm, gn, per_group, ntiles;      // this XCD's row-block range, patch height / width, tiles per row group
    __device__ __forceinline__ void init(int nMb_, int nNb_, int xcd, int nxcd, int slots) {
        nMb = nMb_; nNb = nNb_;
        mb_lo = (int)((long long)nMb * xcd / nxcd);
        nmb = (int)((long long)nMb * (xcd + 1) / nxcd) - mb_lo;
        int g0 = 1;
        while ((g0 + 1) * (g0 + 1) <= slots) ++g0;            // floor(sqrt(slots))
        gn = nNb < g0 ? nNb : g0;
        gm = slots / gn;
        gm = gm < 1 ? 1 : gm;
        if (gm > nmb) {                                       // few row blocks: spend the slots on width instead
            gm = nmb > 0 ? nmb : 1;
            gn = slots / gm;
            gn = gn > nNb ? nNb : (gn < 1 ? 1 : gn);
        }
        per_group = gm * nNb;
        ntiles = nmb * nNb;
    }
    // t-th tile of this XCD -> (row block, column block): row groups of gm row blocks (the last may be shorter), inside a
    // group column chunks of gn (the last may be narrower), inside a patch row-fastest
    __device__ __forceinline__ void get(int t, int& mb, int& nb) const {
        int g = t / per_group;
        const int ngroups = (nmb + gm - 1) / gm;
        g = g < ngroups - 1 ? g : ngroups - 1;
        const int q = t - g * per_group;
        const int h = nmb - g * gm < gm ? nmb - g * gm : gm;
        const int nchunks = (nNb + gn - 1) / gn;
        int c = q / (h * gn);
        c = c < nchunks - 1 ? c : nchunks - 1;
        const int r = q - c * h * gn;
        mb = mb_lo + g * gm + r % h;
        nb = c * gn + r / h;
    }
};

}  // namespace
