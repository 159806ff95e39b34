// Tail of the trunk's TriangleUpdate in ONE kernel (reference primitives/attentions.py:163,170-171; C = 128 pair channels, 32
// einsum channels):
//        z[m,:] += sigmoid(W_g RMSNorm_in(z[m,:]) + b_g) * (W_z RMSNorm_out(o[:,m]) + b_z)
// where o [32][M] is the channel-major output of the triangle einsum (attentions.py:164).  As separate launches this was the
// gate projection (reads z, writes a 33 MB gate tensor), a column-statistics pass over o, and a K = 32 projection with gate +
// residual on the generic fp32 kernel (reads o, the gate tensor and z, writes z): 77 us per triangle update at T = 256, sixty
// of them per trunk pass.  Here a block owns 64 pair rows: it normalises the z rows itself (four threads per row), normalises the
// 32 einsum channels of its rows (one wave per 8 channels, coalesced along m), keeps both as two-part fp16 operands in LDS,
// runs the two contractions (K = 128 and K = 32) with weight fragments straight from global memory, and finishes with one
// read-modify-write of z.  HBM sees z in, o in, z out.  Operand format and accuracy: gemm_f16.hip (bounds: a normalised row
// times a static gain is bounded by sqrt(K) max|w|).
//
// MODE 1 is the tail of the trunk's TriangleAttention (attentions.py:204,212-213) on the same structure:
//        z[m,:] += (W_g RMSNorm(z[m,:]) + b_g) * (W_o o[m,:] + b_o)
// with the attention output o [M][128] (row-major, bounded by max|v|) as the second operand, K = 128 for both contractions and a
// RAW gate: the q|k|v|g projection shrinks to q|k|v (a quarter of its 134 MB output gone) and the gate tensor never exists.
#include "gemm_tile_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int C_ = 128;                 // pair channels
constexpr int CO = 32;                  // einsum channels
constexpr int BM = 64;                  // rows per block tile
constexpr int LP = 136;                 // LDS row pitch of the z operand in fp16 (272 bytes)
template <int MODE> struct TM {
    static constexpr int K2 = MODE == 0 ? CO : C_;           // second contraction's K
    static constexpr int OP = MODE == 0 ? 40 : LP;           // LDS row pitch of the second operand (80 / 272 bytes)
    static constexpr int PART_O = BM * OP;
    static constexpr int NKS2 = K2 / 16;
    static constexpr int WZPART = (C_ / 32) * NKS2 * 1024;   // bytes per part of the second weight matrix
    static constexpr int LDS_BYTES = (2 * BM * LP + 2 * PART_O) * 2 + 4 * BM * 4;
};
constexpr int PART_A = BM * LP;
constexpr int NKS1 = C_ / 16;
constexpr int WGPART = (C_ / 32) * NKS1 * 1024;          // bytes per part of W_g (1 KB fragment blocks)

__device__ __forceinline__ void block_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// lab ablations (timing only, wrong results): 1 no re-read of z in the epilogue, 2 no store of z, 4 no MFMAs, 8 rows of o not loaded,
// 16 rows of z not loaded
#ifdef PD_TRI_TAIL_ABL
constexpr int TABL = PD_TRI_TAIL_ABL;
#else
constexpr int TABL = 0;
#endif
#ifndef PD_TRI_TAIL_GRID0
#define PD_TRI_TAIL_GRID0 3     // blocks per CU of the launch, MODE 0 (47 KB of LDS)
#endif
#ifndef PD_TRI_TAIL_GRID1
#define PD_TRI_TAIL_GRID1 2     // MODE 1 (71 KB)
#endif
__device__ __forceinline__ f32x16 tmma(f16x8 a, f16x8 b, f32x16 c, int, int, int) {
    if constexpr (TABL & 4) { c[0] += (float)a[0] + (float)b[0]; return c; }
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

#ifndef PD_TRI_TAIL_EPI
#define PD_TRI_TAIL_EPI 1       // lab: 0 = z re-read and stored in accumulator order
#endif
#ifndef PD_TRI_TAIL_PF
#define PD_TRI_TAIL_PF 0        // lab: 0 = a tile's rows requested at the top of its own iteration (the round-5 form)
#endif

template <int MODE>
__global__ __launch_bounds__(4 * BM) void tri_tail_kernel(const pd_tri_tail_args p) {
    constexpr int OP = TM<MODE>::OP, PART_O = TM<MODE>::PART_O, NKS2 = TM<MODE>::NKS2, WZPART = TM<MODE>::WZPART;
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    _Float16* sA = lds;                          // [2][BM][LP]   RMSNorm_in(z) w_in, scaled and split
    _Float16* sO = lds + 2 * PART_A;             // [2][BM][OP]   RMSNorm_out(o) w_out, scaled and split
    float* red = reinterpret_cast<float*>(lds + 2 * PART_A + 2 * PART_O);      // [4][BM] partial sums of squares of o
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const auto rsg = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Wg), 0, 2 * WGPART, 0x00020000);
    const auto rsz = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Wz), 0, 2 * WZPART, 0x00020000);
    const int loff = lane * 16;
    auto wfrag_g = [&](int block, int part) {
        return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsg, loff, block * 1024 + part * WGPART, 0));
    };
    auto wfrag_z = [&](int block, int part) {
        return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsz, loff, block * 1024 + part * WZPART, 0));
    };
    const float z_s = pd_pow2_scale(*p.zn_amax), o_s = pd_pow2_scale(*p.on_amax);
    const float inv_z_s = 1.0f / z_s, inv_o_s = 1.0f / o_s;
    const int ntiles = (p.M + BM - 1) / BM;
    const int n = 32 * wave + l31;               // this lane's output column
    const float cg = p.wg_inv[n] * inv_z_s, cz = p.wz_inv[n] * inv_o_s, bg = p.bg ? p.bg[n] : 0.f, bz = p.bz ? p.bz[n] : 0.f;

    // The rows of a tile are requested ONE TILE AHEAD (round 6): a block's life was a chain of round trips - rows of z and o from HBM,
    // barrier, weight fragments from L2, the read-modify-write of z - with two waves per SIMD to hide them (39.7 / 30.7 us per launch for
    // 100 / 75 MB: 2.5 TB/s).  Now the requests of tile t + 1 are issued right behind the barrier that ends tile t's staging and travel
    // under its contractions and its epilogue; same arithmetic, bit-identical results.
    const int pr = tid >> 2, pq = tid & 3;                  // staging: four threads per row, 16-byte chunks interleaved
    f32x4 zv[8];                                            // the thread's chunks of its z row
    f32x4 orv[MODE == 0 ? 1 : 8];                           // MODE 1: its chunks of the attention-output row
    float ov[MODE == 0 ? 8 : 1];                            // MODE 0: eight einsum channels of row `lane`
    auto fetch = [&](int tile) {
        const long long row0 = (long long)tile * BM;
        const bool live = row0 + pr < p.M;
        const float* zr = p.z + (row0 + (live ? pr : 0)) * C_;
#pragma unroll
        for (int i = 0; i < 8; ++i) zv[i] = (TABL & 16) ? f32x4{1.f, 2.f, 3.f, (float)i} : *reinterpret_cast<const f32x4*>(zr + 4 * (pq + 4 * i));
        if constexpr (MODE == 0) {
            const bool livel = row0 + lane < p.M;
            const float* orow = p.o + row0 + (livel ? lane : 0);
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = (TABL & 8) ? (float)e : orow[(long long)(8 * wave + e) * p.M];
        } else {
            const float* orow = p.o + (row0 + (live ? pr : 0)) * C_;
#pragma unroll
            for (int i = 0; i < 8; ++i) orv[i] = (TABL & 8) ? f32x4{1.f, 2.f, 3.f, (float)i} : *reinterpret_cast<const f32x4*>(orow + 4 * (pq + 4 * i));
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long row0 = (long long)tile * BM;
        if (!PD_TRI_TAIL_PF && tile != (int)blockIdx.x) fetch(tile);
        // ---- phase 0: RMSNorm of the tile's z rows (four threads per row) -> scale -> split -> sA
        {
            const int r = pr, q = pq;
            const bool live = row0 + r < p.M;
            f32x4 gw[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) gw[i] = *reinterpret_cast<const f32x4*>(p.w_in + 4 * (q + 4 * i));
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (!live) zv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) sq += zv[i][e] * zv[i][e];
            }
            sq += __shfl_xor(sq, 1);
            sq += __shfl_xor(sq, 2);
            const float rstd = rsqrtf(sq * (1.0f / C_) + p.eps) * z_s;      // the operand scale rides on rstd
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = 4 * (q + 4 * i);
                float t[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = zv[i][e] * rstd * gw[i][e];
                const pd_parts2 p0 = pd_split2h(t[0], t[1]), p1 = pd_split2h(t[2], t[3]);
                *reinterpret_cast<u32x2*>(sA + r * LP + c) = u32x2{p0.h, p1.h};
                *reinterpret_cast<u32x2*>(sA + PART_A + r * LP + c) = u32x2{p0.l, p1.l};
            }
        }
        if constexpr (MODE == 0) {
            // ---- phase 0b: the 32 einsum channels of the tile's rows: wave = 8 channels, lane = row (coalesced along m)
            {
                const bool live = row0 + lane < p.M;
                float ss = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (!live) ov[e] = 0.f;
                    ss += ov[e] * ov[e];
                }
                red[wave * BM + lane] = ss;
            }
            block_barrier();
            {
                const float ss = red[lane] + red[BM + lane] + red[2 * BM + lane] + red[3 * BM + lane];
                const float rstd = rsqrtf(ss * (1.0f / CO) + p.eps) * o_s;
                float t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = ov[e] * rstd * p.w_out[8 * wave + e];
                const pd_parts2 p0 = pd_split2h(t[0], t[1]), p1 = pd_split2h(t[2], t[3]), p2 = pd_split2h(t[4], t[5]), p3 = pd_split2h(t[6], t[7]);
                *reinterpret_cast<u32x4*>(sO + lane * OP + 8 * wave) = u32x4{p0.h, p1.h, p2.h, p3.h};
                *reinterpret_cast<u32x4*>(sO + PART_O + lane * OP + 8 * wave) = u32x4{p0.l, p1.l, p2.l, p3.l};
            }
        } else {
            // ---- phase 0b: the attention output rows (row-major, no norm): scale -> split -> sO
            const int r = pr, q = pq;
            const bool live = row0 + r < p.M;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = 4 * (q + 4 * i);
                f32x4 v = orv[i];
                if (!live) v = f32x4{0.f, 0.f, 0.f, 0.f};
                const pd_parts2 p0 = pd_split2h(v[0] * o_s, v[1] * o_s), p1 = pd_split2h(v[2] * o_s, v[3] * o_s);
                *reinterpret_cast<u32x2*>(sO + r * OP + c) = u32x2{p0.h, p1.h};
                *reinterpret_cast<u32x2*>(sO + PART_O + r * OP + c) = u32x2{p0.l, p1.l};
            }
        }
        block_barrier();
        if (PD_TRI_TAIL_PF && tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);      // the next tile's rows travel from here on
        __builtin_amdgcn_sched_barrier(0);

        // ---- phase 1: gate logits = sA . W_g^T (K = 128), phase 2: update = sO . W_z^T (K = 32); this wave: all 64 rows x columns
        // [32 wave, +32)
        f32x16 accg[2], accz[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accg[i][r] = 0.f; accz[i][r] = 0.f; }
        {
            constexpr int PF = 3;
            f16x8 wf[PF + 1][2];
            auto wload = [&](int buf, int ks) {
                wf[buf][0] = wfrag_g(wave * NKS1 + ks, 0);
                wf[buf][1] = wfrag_g(wave * NKS1 + ks, 1);
            };
#pragma unroll
            for (int ks = 0; ks < PF; ++ks) wload(ks, ks);
            const _Float16* abase = sA + l31 * LP + 8 * hh;
#pragma unroll
            for (int ks = 0; ks < NKS1; ++ks) {
                if (ks + PF < NKS1) wload((ks + PF) % (PF + 1), ks + PF);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const f16x8 a0 = *reinterpret_cast<const f16x8*>(abase + 32 * i * LP + 16 * ks);
                    const f16x8 a1 = *reinterpret_cast<const f16x8*>(abase + PART_A + 32 * i * LP + 16 * ks);
                    f32x16 t = accg[i];
                    t = tmma(a0, wf[ks % (PF + 1)][1], t, 0, 0, 0);
                    t = tmma(a1, wf[ks % (PF + 1)][0], t, 0, 0, 0);
                    t = tmma(a0, wf[ks % (PF + 1)][0], t, 0, 0, 0);
                    accg[i] = t;
                }
            }
            // second contraction: the same ring on the second weight matrix (NKS2 = 2 or 8 steps)
            constexpr int PF2 = NKS2 < PF ? NKS2 : PF;
            auto zload = [&](int buf, int ks) {
                wf[buf][0] = wfrag_z(wave * NKS2 + ks, 0);
                wf[buf][1] = wfrag_z(wave * NKS2 + ks, 1);
            };
#pragma unroll
            for (int ks = 0; ks < PF2; ++ks) zload(ks, ks);
            const _Float16* obase = sO + l31 * OP + 8 * hh;
#pragma unroll
            for (int ks = 0; ks < NKS2; ++ks) {
                if (ks + PF2 < NKS2) zload((ks + PF2) % (PF + 1), ks + PF2);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const f16x8 a0 = *reinterpret_cast<const f16x8*>(obase + 32 * i * OP + 16 * ks);
                    const f16x8 a1 = *reinterpret_cast<const f16x8*>(obase + PART_O + 32 * i * OP + 16 * ks);
                    f32x16 t = accz[i];
                    t = tmma(a0, wf[ks % (PF + 1)][1], t, 0, 0, 0);
                    t = tmma(a1, wf[ks % (PF + 1)][0], t, 0, 0, 0);
                    t = tmma(a0, wf[ks % (PF + 1)][0], t, 0, 0, 0);
                    accz[i] = t;
                }
            }
        }
        if constexpr (PD_TRI_TAIL_EPI == 0) {
        // ---- epilogue, round-5 form (lab): z re-read and stored in accumulator order (lane = column: 4-byte accesses, 128-byte row segments)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long rb = row0 + 32 * i + 4 * hh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long row = rb + (r & 3) + 8 * (r >> 2);
                if (row < p.M) {
                    float* zp = p.z + row * C_ + n;
                    const float gl = accg[i][r] * cg + bg;
                    const float zo = (TABL & 1) ? cg : *zp;
                    const float zn = zo + (MODE == 0 ? pd_sigmoid(gl) : gl) * (accz[i][r] * cz + bz);
                    if (!(TABL & 2) || zn == 12345.f) *zp = zn;
                }
            }
        }
        } else {
        // ---- epilogue: z += sigmoid(gate logits) * update with z read ONCE.  The staging threads still hold their chunks of the z rows
        // (zv: four threads per row); the accumulators hold lane = column.  Re-reading z in accumulator order cost 7.3 of 38 us and
        // storing it 2.8 (ablations, profiles/r06_tri_tail_ablations.txt).  Here the rows pass through LDS - an fp32 tile in the space
        // of the z operand, dead once every wave has left the contractions - : rows in (16-byte pieces), update in place in accumulator
        // order (the same expression on the same values: bit-identical), rows out and to HBM in the 16-byte pieces they were loaded in.
        float* const D = reinterpret_cast<float*>(sA);           // [BM][LP] floats = the bytes of sA's two parts
        block_barrier();                                          // X: the operand tiles are dead
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(D + pr * LP + 4 * (pq + 4 * i)) = zv[i];
        block_barrier();                                          // Y: the z rows are in D
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* dp = D + (32 * i + 4 * hh + (r & 3) + 8 * (r >> 2)) * LP + n;
                const float gl = accg[i][r] * cg + bg;
                const float zo = *dp;
                const float zn = zo + (MODE == 0 ? pd_sigmoid(gl) : gl) * (accz[i][r] * cz + bz);
                *dp = zn;
            }
        }
        block_barrier();                                          // Z: the updated rows are in D
        if (row0 + pr < p.M && (!(TABL & 2) || cg == 12345.f)) {
            float* zr = p.z + (row0 + pr) * C_;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<f32x4*>(zr + 4 * (pq + 4 * i)) = *reinterpret_cast<const f32x4*>(D + pr * LP + 4 * (pq + 4 * i));
        }
        }
        block_barrier();                          // the LDS tiles are free for the next tile
    }
}

}  // namespace

// see include/physdock_hip.h pd_tri_tail_args.  args == nullptr: one-time set-up (dynamic LDS limits), called by pd_init.
PD_EXPORT int pd_tri_tail(const pd_tri_tail_args* a, void* stream) {
    if (!a)
        return hipFuncSetAttribute(reinterpret_cast<const void*>(tri_tail_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, TM<0>::LDS_BYTES) == hipSuccess &&
                       hipFuncSetAttribute(reinterpret_cast<const void*>(tri_tail_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, TM<1>::LDS_BYTES) == hipSuccess
                   ? PD_OK : PD_ERR_LAUNCH;
    if (!a->z || !a->o || !a->w_in || !a->Wg || !a->wg_inv || !a->Wz || !a->wz_inv || !a->zn_amax || !a->on_amax) return PD_ERR_ARG;
    if (a->mode == 0 && !a->w_out) return PD_ERR_ARG;
    if (a->mode != 0 && a->mode != 1) return PD_ERR_ARG;
    if (a->C != C_ || a->Co != (a->mode == 0 ? CO : C_) || a->M <= 0) return PD_ERR_UNSUPPORTED;
    if (((uintptr_t)a->z | (uintptr_t)a->w_in | (uintptr_t)a->Wg | (uintptr_t)a->Wz | (a->mode ? (uintptr_t)a->o : 0)) & 15) return PD_ERR_UNSUPPORTED;
    const int ntiles = (a->M + BM - 1) / BM;
    if (a->mode == 0) {
        const int grid = 256 * PD_TRI_TAIL_GRID0;
        hipLaunchKernelGGL(tri_tail_kernel<0>, dim3(ntiles < grid ? ntiles : grid), dim3(4 * BM), TM<0>::LDS_BYTES, (hipStream_t)stream, *a);
    } else {
        const int grid = 256 * PD_TRI_TAIL_GRID1;
        hipLaunchKernelGGL(tri_tail_kernel<1>, dim3(ntiles < grid ? ntiles : grid), dim3(4 * BM), TM<1>::LDS_BYTES, (hipStream_t)stream, *a);
    }
    return pd_check_launch();
}
