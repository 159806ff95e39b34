// Atom-level DiT transition in ONE kernel (reference primitives/transitions.py:27-30 = AdaLN-Zero + SwiGLU feed-forward + gate +
// residual; C = 128 channels, hidden = 384):      x += gate * W2 . ( silu(W1 y) * (W3 y) ),   y = (1 + scale) LayerNorm(x) + shift.
//
// As three launches (row statistics, SwiGLU up-projection, down-projection) this step moves the hidden tensor through HBM
// twice - 201 MB out and 201 MB back in at 64 samples - around two K = 128 / K = 384 GEMMs whose tiles are prologue / epilogue
// bound (MfmaUtil 0.2).  Here a block owns 128 whole rows: it normalises them itself (a row is 128 values: four threads), keeps
// the two-part fp16 operand in LDS for the whole tile, and walks the hidden dimension in three chunks of 128: GLU chunk ->
// split -> LDS -> accumulated into the 128 x 128 output tile.  The hidden activations never leave the CU; HBM sees x in, x out.
// Operand format, scales and accuracy: gemm_f16.hip (two fp16 parts, three products; bounds from pd_dit_bounds).
// One block per CU (139 KB of LDS: the A tile and one hidden chunk, both as two fp16 parts with 272-byte rows = conflict-free
// ds_read_b128 fragments), eight waves with up to 256 VGPRs: the weights come straight from global memory in fragment-major
// order (packing.split2_f16), one 16-k step ahead.
#include "gemm_tile_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int C_ = 128;                 // channels = K of the up-projection = N of the down-projection
#ifndef PD_TRANSITION_BM
#define PD_TRANSITION_BM 64
#endif
constexpr int CH = 128;                 // hidden columns per chunk
constexpr int LP = 136;                 // LDS row pitch in fp16 (272 bytes: 17 x 16)
// rows per block tile: 128 rows on eight waves, one block per CU - or 64 rows on four waves, TWO blocks per CU: the same wave
// tiles and the same 139 KB of LDS per CU, but one block's LayerNorm / GLU epilogue / read-modify-write phases overlap the
// other's matrix phases (the weights are streamed twice as often: 1.2 GB per launch through the L2s instead of 0.6)
template <int BM> struct TT {
    static constexpr int NT = 4 * BM;                       // threads: four per row in phase 0
    static constexpr int PART = BM * LP;                    // fp16 elements per part of a tile
    static constexpr int LDS_BYTES = 2 * 2 * PART * 2;      // (A tile + hidden chunk) x 2 parts
    static constexpr int BLOCKS_PER_CU = 128 / BM;
};

__device__ __forceinline__ void block_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int NCH, int BM>              // hidden = 128 * NCH
__global__ __launch_bounds__(4 * BM) __attribute__((amdgpu_waves_per_eu(2, 2)))
void transition_f16_kernel(const pd_transition_args p) {
    constexpr int PART = TT<BM>::PART;
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    _Float16* sA = lds;                          // [2][128][LP]   y, scaled and split
    _Float16* sH = lds + 2 * PART;               // [2][128][LP]   hidden chunk, scaled and split
    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index as a scalar: the fragment block offsets derived from it must be SGPR operands of the buffer loads (from the
    // VGPR tid >> 6 hipcc wraps every load in a v_readfirstlane waterfall loop)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    constexpr int NKS1 = C_ / 16;                // 16-k steps of the up-projection (8)
    constexpr int NKS2 = CH * NCH / 16;          // ... of the down-projection (24)
    constexpr int W13PART = (2 * CH * NCH / 32) * NKS1 * 1024;       // bytes per part (1 KB fragment blocks)
    constexpr int W2PART = (C_ / 32) * NKS2 * 1024;
    constexpr int PF = 3;                        // weight fragments are requested PF 16-k steps ahead (L2 latency >> one step's MFMAs)
    // Weight fragments through buffer loads: descriptor in SGPRs, ONE shared VGPR offset (lane * 16), the fragment block as a
    // scalar offset (with flat 64-bit per-lane addresses hipcc hoists ~30 of them out of the chunk loop and spills them).
    const auto rs13 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W13), 0, 2 * W13PART, 0x00020000);
    const auto rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W2), 0, 2 * W2PART, 0x00020000);
    const int loff = lane * 16;
    auto wfrag13 = [&](int block, int part) {
        return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs13, loff, block * 1024 + part * W13PART, 0));
    };
    auto wfrag2 = [&](int block, int part) {
        return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs2, loff, block * 1024 + part * W2PART, 0));
    };
    const float y_s = pd_pow2_scale(*p.y_amax), h_s = pd_pow2_scale(*p.h_amax);
    const float inv_y_s = 1.0f / y_s, inv_h_s = 1.0f / h_s;
    // up-projection: 2 x 4 waves, 64 rows x 64 packed columns (one GLU pair = 32 hidden) each: a weight fragment feeds two row
    // fragments (half the weight loads per MFMA of the 32 x 128 layout)
    const int wm1 = wave >> 2, wn1 = wave & 3;
    // down-projection: 2 x 4 waves, 64 rows x 32 output columns each
    const int wm2 = wave >> 2, wn2 = wave & 3;
    const int ntiles = p.M / BM;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long row0 = (long long)tile * BM;
        // ---- phase 0: LayerNorm + AdaLN + scale + split of the tile's 128 rows into LDS (four threads per row)
        {
            const int r = tid >> 2, q = tid & 3;
            const float* xr = p.x + (row0 + r) * C_;
            // the AdaLN gain / shift rows of this row's group first: they do not depend on the statistics, so their latency
            // overlaps the x loads and the two reductions
            const long long goff = p.rows_per_group > 0 ? ((row0 + r) / p.rows_per_group) * (long long)p.gstride : 0;
            f32x4 v[8], gw[8], gb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                gw[i] = *reinterpret_cast<const f32x4*>(p.scale1p + goff + 4 * (q + 4 * i));
                gb[i] = *reinterpret_cast<const f32x4*>(p.shift + goff + 4 * (q + 4 * i));
            }
            float s1 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * (q + 4 * i));
                s1 += v[i][0] + v[i][1] + v[i][2] + v[i][3];
            }
            s1 += __shfl_xor(s1, 1);
            s1 += __shfl_xor(s1, 2);
            const float mean = p.rms ? 0.f : s1 * (1.0f / C_);
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; sq += d * d; }
            sq += __shfl_xor(sq, 1);
            sq += __shfl_xor(sq, 2);
            const float rstd = rsqrtf(sq * (1.0f / C_) + p.eps) * y_s;      // the operand scale rides on rstd ...
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = 4 * (q + 4 * i);
                const f32x4 w = gw[i], b = gb[i];
                float t[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = (v[i][e] - mean) * rstd * w[e] + b[e] * y_s;      // ... and on the shift
                const pd_parts2 p0 = pd_split2h(t[0], t[1]), p1 = pd_split2h(t[2], t[3]);
                *reinterpret_cast<u32x2*>(sA + r * LP + c) = u32x2{p0.h, p1.h};
                *reinterpret_cast<u32x2*>(sA + PART + r * LP + c) = u32x2{p0.l, p1.l};
            }
        }
        block_barrier();

        f32x16 acc2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][r] = 0.f;

#pragma unroll 1
        for (int c = 0; c < NCH; ++c) {
            // ---- phase 1: packed GLU columns [256 c + 64 wn1, +64) of W13 against rows [64 wm1, +64) of the A tile
            f32x16 acc1[2][2];                                     // [row fragment][a | b]
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc1[i][j][r] = 0.f;
            const int nb = 2 * CH * c + 64 * wn1;                  // first packed column of this wave
            f16x8 wf[PF + 1][2][2];
            auto wload = [&](int buf, int ks) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int block = ((nb + 32 * j) >> 5) * NKS1 + ks;
                    wf[buf][j][0] = wfrag13(block, 0);
                    wf[buf][j][1] = wfrag13(block, 1);
                }
            };
#pragma unroll
            for (int ks = 0; ks < PF; ++ks) wload(ks, ks);
            const _Float16* abase = sA + (64 * wm1 + l31) * LP + 8 * hh;
#pragma unroll
            for (int ks = 0; ks < NKS1; ++ks) {
                if (ks + PF < NKS1) wload((ks + PF) % (PF + 1), ks + PF);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const f16x8 a0 = *reinterpret_cast<const f16x8*>(abase + 32 * i * LP + 16 * ks);
                    const f16x8 a1 = *reinterpret_cast<const f16x8*>(abase + PART + 32 * i * LP + 16 * ks);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x16 t = acc1[i][j];
                        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, wf[ks % (PF + 1)][j][1], t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, wf[ks % (PF + 1)][j][0], t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, wf[ks % (PF + 1)][j][0], t, 0, 0, 0);
                        acc1[i][j] = t;
                    }
                }
            }
            // GLU in registers, then scale + split into the hidden chunk tile: lane = hidden column, register = row; a register
            // pair (two adjacent rows) shares one split, its halves go to the two rows
            {
                const float ca = p.w13_inv[nb + l31] * inv_y_s, cb = p.w13_inv[nb + 32 + l31] * inv_y_s;
                const int hid = 32 * wn1 + l31;                    // column inside the chunk
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    unsigned short* hs = reinterpret_cast<unsigned short*>(sH) + (64 * wm1 + 32 * i + 4 * hh) * LP + hid;
#ifndef PD_TR_SILU
#define PD_TR_SILU 0      // 0: IEEE division (shipped: measured fastest), 1: v_rcp_f32 in place, 2: v_rcp_f32 with the 16 gates of a fragment first, 3: v_rcp_f32 + a Newton step
#endif
#if PD_TR_SILU == 2
                    float gate[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) gate[r] = __builtin_amdgcn_rcpf(1.0f + __expf(-acc1[i][0][r] * ca));
#endif
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
#if PD_TR_SILU == 0
                        const float h0 = pd_silu(acc1[i][0][r] * ca) * (acc1[i][1][r] * cb) * h_s;
                        const float h1 = pd_silu(acc1[i][0][r + 1] * ca) * (acc1[i][1][r + 1] * cb) * h_s;
#elif PD_TR_SILU == 1
                        const float h0 = pd_silu_r(acc1[i][0][r] * ca) * (acc1[i][1][r] * cb) * h_s;
                        const float h1 = pd_silu_r(acc1[i][0][r + 1] * ca) * (acc1[i][1][r + 1] * cb) * h_s;
#elif PD_TR_SILU == 2
                        const float h0 = (acc1[i][0][r] * ca) * gate[r] * (acc1[i][1][r] * cb) * h_s;
                        const float h1 = (acc1[i][0][r + 1] * ca) * gate[r + 1] * (acc1[i][1][r + 1] * cb) * h_s;
#else       // 3: reciprocal + one Newton step (four instructions instead of the division's ten, 0.5 ulp)
                        auto silu_n = [](float x) {
                            const float d = 1.0f + __expf(-x);
                            const float r0 = __builtin_amdgcn_rcpf(d);
                            return x * __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
                        };
                        const float h0 = silu_n(acc1[i][0][r] * ca) * (acc1[i][1][r] * cb) * h_s;
                        const float h1 = silu_n(acc1[i][0][r + 1] * ca) * (acc1[i][1][r + 1] * cb) * h_s;
#endif
                        const pd_parts2 s2 = pd_split2h(h0, h1);
                        const int ro = ((r & 3) + 8 * (r >> 2)) * LP;      // row of register r (the lane half's 4 hh is in hs)
                        hs[ro] = (unsigned short)s2.h;
                        hs[ro + LP] = (unsigned short)(s2.h >> 16);
                        hs[PART + ro] = (unsigned short)s2.l;
                        hs[PART + ro + LP] = (unsigned short)(s2.l >> 16);
                    }
                }
            }
            block_barrier();
            // ---- phase 2: output tile += hidden chunk . W2[:, 128 c .. +128]^T
            {
                f16x8 vf[PF + 1][2];
                auto vload = [&](int buf, int ks) {
                    const int block = wn2 * NKS2 + (CH / 16) * c + ks;
                    vf[buf][0] = wfrag2(block, 0);
                    vf[buf][1] = wfrag2(block, 1);
                };
#pragma unroll
                for (int ks = 0; ks < PF; ++ks) vload(ks, ks);
                const _Float16* hbase = sH + (64 * wm2 + l31) * LP + 8 * hh;
#pragma unroll
                for (int ks = 0; ks < CH / 16; ++ks) {
                    if (ks + PF < CH / 16) vload((ks + PF) % (PF + 1), ks + PF);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const f16x8 a0 = *reinterpret_cast<const f16x8*>(hbase + 32 * i * LP + 16 * ks);
                        const f16x8 a1 = *reinterpret_cast<const f16x8*>(hbase + PART + 32 * i * LP + 16 * ks);
                        f32x16 t = acc2[i];
                        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, vf[ks % (PF + 1)][1], t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, vf[ks % (PF + 1)][0], t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, vf[ks % (PF + 1)][0], t, 0, 0, 0);
                        acc2[i] = t;
                    }
                }
            }
            block_barrier();                          // the chunk tile is free for the next chunk (and sA for the next tile)
        }
        // ---- epilogue: x = x + gate * (acc / scales)
        {
            const int n = 32 * wn2 + l31;
            const float cs = p.w2_inv[n] * inv_h_s;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const long long rb = row0 + 64 * wm2 + 32 * i + 4 * hh;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long row = rb + (r & 3) + 8 * (r >> 2);
                    const long long goff = p.rows_per_group > 0 ? (row / p.rows_per_group) * (long long)p.gstride : 0;
                    float* xp = p.x + row * C_ + n;
                    *xp = *xp + acc2[i][r] * cs * p.gate[goff + n];
                }
            }
        }
    }
}

}  // namespace

// x [M][128] updated in place; see include/physdock_hip.h pd_transition_args.  PD_ERR_UNSUPPORTED for other shapes (the caller
// then runs the three-launch form).  init: M <= 0 with args == nullptr raises the dynamic-LDS limit.
#ifndef PD_TRANSITION_MIN128
#define PD_TRANSITION_MIN128 16       // (round 5: 256 -> 16; same-box ms per call at 2 / 4 / 8 / 12 samples 94.4 / 104.9 / 140.0 / 144.1 -> 92.5 / 102.4 / 136.4 / 138.8)
#endif
PD_EXPORT int pd_transition_f16(const pd_transition_args* a, void* stream) {
    constexpr int BM = PD_TRANSITION_BM;
    typedef TT<BM> T;
    auto k = transition_f16_kernel<3, BM>;
    if (!a) {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES) == hipSuccess
                   ? PD_OK : PD_ERR_LAUNCH;
    }
    if (!a->x || !a->shift || !a->scale1p || !a->gate || !a->W13 || !a->w13_inv || !a->W2 || !a->w2_inv || !a->y_amax || !a->h_amax)
        return PD_ERR_ARG;
    if (a->C != C_ || a->hidden != 3 * CH || a->M <= 0 || a->M % BM != 0) return PD_ERR_UNSUPPORTED;
    if (a->M / 128 < PD_TRANSITION_MIN128) return PD_ERR_UNSUPPORTED;   // (below 2 048 rows the launches are latency-bound either way)
    if (a->rows_per_group > 0 && a->gstride % 4 != 0) return PD_ERR_UNSUPPORTED;
    if (((uintptr_t)a->x | (uintptr_t)a->shift | (uintptr_t)a->scale1p | (uintptr_t)a->W13 | (uintptr_t)a->W2) & 15) return PD_ERR_UNSUPPORTED;
    const int ntiles = a->M / BM, grid = 256 * T::BLOCKS_PER_CU;
    hipLaunchKernelGGL(k, dim3(ntiles < grid ? ntiles : grid), dim3(T::NT), T::LDS_BYTES, (hipStream_t)stream, *a);
    return pd_check_launch();
}
