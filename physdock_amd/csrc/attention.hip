// Biased multi-head attention, head width 32, exact fp32 on v_mfma_f32_32x32x2_f32.
//
// Replaces F.scaled_dot_product_attention in every attention of the hot path
// (reference primitives/attentions.py:48,92,130,211,259): DiT atom/token attention,
// trunk atom attention, MSA row/column attention, triangle attention (row/col) and the
// pair-biased single attention.
//
// Structure (flash style, one block = 4 waves = 128 queries of one (batch, head)):
//   * K/V tiles of 64 keys are staged through LDS (double-buffered, register prefetch) and
//     shared by the 4 waves; Q (32 rows per wave) lives in registers, pre-scaled by
//     scale*log2(e).
//   * Both MFMAs run "swapped": S^T = K.Q^T and O^T = V^T.P^T, so every lane owns ONE
//     query (lane&31) and 16 of its keys / output channels.  The softmax row max/sum is then
//     15 in-lane ops + one cross-half shuffle, P feeds the second MFMA straight from the
//     accumulator registers (no LDS round trip), and the running rescale of O is lane-local.
//   * The pair bias arrives in "fragment layout" [H][q-tile][k-tile][4][64][4] (written by
//     pd_gemm PD_OUT_BIASFRAG, already multiplied by log2 e): each wave fetches its 32x32
//     bias tile with four fully-coalesced 1 KiB loads.  The bias does not depend on the
//     batch index; batch is the fastest grid dimension so the blocks that share a bias tile
//     run together and the tile is served from L2 / Infinity Cache (the step- and
//     sample-invariant biases of the DiT are hoisted out of the loop entirely).
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "physdock_hip.h"

namespace {

constexpr int KT = 64;      // keys per LDS tile
constexpr int LDKS = 36;    // K row stride (floats): conflict-free ds_read_b128
constexpr int LDVS = 32;    // V row stride

#ifdef PD_LAB
__device__ unsigned long long* g_attn_trace = nullptr;
#endif

// NW = waves (32-query tiles) per block: 4, or 8 for long query ranges - the K/V tiles staged through LDS are then shared
// by twice as many MFMAs (half the global-load / LDS-write traffic per flop), at the same 4 waves per SIMD.
// SPLIT: the key range is cut into p.nsplit chunks handled by different blocks (blockIdx.y = query block * nsplit + chunk);
// each writes its un-normalised output and (running max, sum) to the workspace, attn_combine_kernel merges them.  For
// launches that cannot fill the chip (few samples): every wave's key loop is serial, so only shorter loops cut latency.
template <int NW, bool SPLIT>
__global__ __launch_bounds__(64 * NW, 4) void attn_kernel(const pd_attn_args p) {
    __shared__ __attribute__((aligned(16))) float sK[2][KT * LDKS];
    __shared__ __attribute__((aligned(16))) float sV[2][KT * LDVS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    // Batch is the fastest grid dimension, so the blocks that share a bias tile and the query blocks that share the K/V
    // panels of one (batch, head) are resident together.  Two XCD-aware remappings of the block index (whole (query
    // block, head) groups per XCD, interleaved or contiguous) were measured: both raise the fetch traffic of the DiT atom
    // attention from 433 to 637 MB (FETCH_SIZE) at unchanged run time, so the plain order stays.
    const int b = blockIdx.x, h = blockIdx.z;
    const int qb = SPLIT ? blockIdx.y / p.nsplit : blockIdx.y, chunk = SPLIT ? blockIdx.y % p.nsplit : 0;
    const int q0 = qb * (32 * NW) + wave * 32;
    const int query = q0 + l31;
    const bool wave_active = q0 < p.nq;

    const float* Kb = p.K + (long long)b * p.k_bs + h * 32;
    const float* Vb = p.V + (long long)b * p.v_bs + h * 32;

    // Q fragment: lane (query, hh) holds dims 8t+4hh+e
    f32x4 qf[4];
    {
        const float qs = p.scale * PD_LOG2E;
        const float* qp = p.Q + (long long)b * p.q_bs + (long long)query * p.q_ss + h * 32 + 4 * hh;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (query < p.nq) v = *reinterpret_cast<const f32x4*>(qp + 8 * t);
            qf[t] = v * qs;
        }
    }

    // tile pitch of the bias buffer = the (padded) key / query counts its writer used (pd_gemm PD_OUT_BIASFRAG T2 / T1),
    // which exceed the reduction bound nk when the boundary padded the sequence with masked entries
    const int nkt32 = ((p.bias_nk > 0 ? p.bias_nk : p.nk) + 31) >> 5;
    const int nqt32 = (p.nq + 31) >> 5;
    const float* bias_wave = nullptr;
    if (p.bias && wave_active)
        bias_wave = p.bias + (((long long)h * nqt32 + (q0 >> 5)) * nkt32) * 1024 + lane * 4;

    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // staging: thread -> key row (tid>>3)+RPP*i, 16-byte chunk tid&7
    constexpr int RPP = 8 * NW;            // key rows covered per pass of the block
    constexpr int NST = KT / RPP;          // passes per 64-key tile (2 for 4 waves, 1 for 8)
    const int srow = tid >> 3, schunk = (tid & 7) * 4;
    f32x4 rk[NST], rv[NST];
    auto gload = [&](int key0) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int key = key0 + srow + RPP * i;
            rk[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            rv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (key < p.nk) {
                rk[i] = *reinterpret_cast<const f32x4*>(Kb + (long long)key * p.k_ss + schunk);
                rv[i] = *reinterpret_cast<const f32x4*>(Vb + (long long)key * p.v_ss + schunk);
            }
        }
    };
    auto sstore = [&](int st) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            *reinterpret_cast<f32x4*>(&sK[st][(srow + RPP * i) * LDKS + schunk]) = rk[i];
            *reinterpret_cast<f32x4*>(&sV[st][(srow + RPP * i) * LDVS + schunk]) = rv[i];
        }
    };

    const int nit_all = (p.nk + KT - 1) / KT;
    const int it_lo = SPLIT ? (int)((long long)nit_all * chunk / p.nsplit) : 0;
    const int nit = SPLIT ? (int)((long long)nit_all * (chunk + 1) / p.nsplit) : nit_all;
    gload(it_lo * KT);
    sstore(it_lo & 1);
    __syncthreads();

    // one 32-key sub-tile: S^T = K.Q^T (+bias), online softmax, O^T += V^T.P^T.  RAGGED is only instantiated for the
    // single partial tile at the end of the key range, so the steady-state loop carries no masking code at all.
    auto subtile = [&](auto ragged_tag, int cur, int sub, int kt32) {
        constexpr bool RAGGED = decltype(ragged_tag)::value;
        f32x4 bf[4];
        if (bias_wave) {   // bias tile first: its latency hides under the 16 QK^T MFMAs
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bf[g] = *reinterpret_cast<const f32x4*>(bias_wave + (long long)kt32 * 1024 + g * 256);
        }
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float* kbase = &sK[cur][(sub * 32 + l31) * LDKS + 4 * hh];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(kbase + 8 * t);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[t][e], s, 0, 0, 0);
        }
        if (bias_wave) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] += bf[r >> 2][r & 3];
        }
        if constexpr (RAGGED) {   // padded keys drop out
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
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
            psum += s[r];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= alpha;
        const float* vbase = &sV[cur][(sub * 32 + 4 * hh) * LDVS + l31];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float vf = vbase[((r & 3) + 8 * (r >> 2)) * LDVS];
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, s[r], o, 0, 0, 0);
        }
    };
    const int nfull32 = p.nk >> 5;                 // sub-tiles with all 32 keys in range
#ifdef PD_LAB                                      // lab build only (tools/attn_trace.py): in-kernel phase trace
    unsigned long long* dbg = nullptr;
    if (g_attn_trace && lane == 0 && blockIdx.x < 8 && blockIdx.y < 2 && blockIdx.z == 0 && wave < 4)
        dbg = g_attn_trace + (((long long)blockIdx.y * 8 + blockIdx.x) * 4 + wave) * (4 * 64);
#define PD_STAMP(slot) if (dbg && it < 64) dbg[it * 4 + slot] = __builtin_amdgcn_s_memtime()
#else
#define PD_STAMP(slot)
#endif

    for (int it = it_lo; it < nit; ++it) {
        const int cur = it & 1;
        PD_STAMP(0);
        if (it + 1 < nit) gload((it + 1) * KT);
        PD_STAMP(1);
        if (wave_active) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int kt32 = it * 2 + sub;
                if (kt32 < nfull32) subtile(std::false_type{}, cur, sub, kt32);
                else if (kt32 * 32 < p.nk) subtile(std::true_type{}, cur, sub, kt32);
            }
        }
        PD_STAMP(2);
        if (it + 1 < nit) sstore(cur ^ 1);
        __syncthreads();
        PD_STAMP(3);
    }
#undef PD_STAMP

    if constexpr (SPLIT) {
        if (query < p.nq) {      // partial result of this key chunk: [chunk][b][query][h*32 + dim], (m, l) per (chunk, b, h, query)
            const float l = pd_xhalf_sum(l_run);
            const int C = p.nheads * 32;
            float* wo = p.ws + (((long long)chunk * p.nbatch + b) * p.nq + query) * C + h * 32 + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
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
        const float inv = 1.0f / l;
        float* op = p.O + (long long)b * p.o_bs + (long long)query * p.o_ss + h * 32 + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v = {o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv};
            *reinterpret_cast<f32x4*>(op + 8 * g) = v;
        }
    }
}

// merge the key chunks of a split launch: O = sum_s o_s 2^(m_s - M) / sum_s l_s 2^(m_s - M); 8 threads per (b, q, h)
__global__ __launch_bounds__(256) void attn_combine_kernel(const pd_attn_args p) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int C = p.nheads * 32;
    const long long total = (long long)p.nbatch * p.nq * p.nheads * 8;
    if (t >= total) return;
    const int d4 = (int)(t & 7);
    const int h = (int)((t >> 3) % p.nheads);
    const long long bq = (t >> 3) / p.nheads;
    const int q = (int)(bq % p.nq), b = (int)(bq / p.nq);
    const float* ml0 = p.ws + (long long)p.nsplit * p.nbatch * p.nq * C;
    float M = -INFINITY;
    for (int s = 0; s < p.nsplit; ++s) M = fmaxf(M, ml0[((((long long)s * p.nbatch + b) * p.nheads + h) * p.nq + q) * 2]);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float L = 0.f;
    for (int s = 0; s < p.nsplit; ++s) {
        const float* ml = ml0 + ((((long long)s * p.nbatch + b) * p.nheads + h) * p.nq + q) * 2;
        const float w = __builtin_amdgcn_exp2f(ml[0] - M);
        L += ml[1] * w;
        acc += *reinterpret_cast<const f32x4*>(p.ws + (((long long)s * p.nbatch + b) * p.nq + q) * C + h * 32 + d4 * 4) * w;
    }
    const f32x4 o = acc * (1.0f / L);
    if (p.O2) {
        // the split output of the fp16-format kernels (pd_gemm_args.A2): o times the power of two of the v bound, two fp16 parts
        const float sv = pd_pow2_scale(p.f16_amax ? p.f16_amax[2] : p.f16_v_amax);
        const long long rows = p.o2_rows > 0 ? p.o2_rows : (long long)p.nbatch * p.nq;
        unsigned short* op = reinterpret_cast<unsigned short*>(p.O2) + ((long long)b * p.nq + q) * C + h * 32 + d4 * 4;
        const pd_parts2 p0 = pd_split2h(o[0] * sv, o[1] * sv), p1 = pd_split2h(o[2] * sv, o[3] * sv);
        typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2_*>(op) = u32x2_{p0.h, p1.h};
        *reinterpret_cast<u32x2_*>(op + rows * C) = u32x2_{p0.l, p1.l};
        return;
    }
    *reinterpret_cast<f32x4*>(p.O + (long long)b * p.o_bs + (long long)q * p.o_ss + h * 32 + d4 * 4) = o;
}

}  // namespace

// number of key chunks pd_attention would use (1 = no split): only when the launch leaves most of the chip idle, the
// key range is long enough and the caller supplied a workspace of nsplit * nbatch * nq * (32 + 2) * nheads floats
#ifndef PD_ATTN_MIN_WAVES
#define PD_ATTN_MIN_WAVES 128      // (round 5: 1024 -> 128; per call at 1 / 2 / 4 / 7 samples 87.8 / 94.2 / 103.9 / 122.5 -> 85.0 / 91.6 / 101.2 / 119.6 ms: a 256-key launch of 32 blocks is latency-bound on either pipe and the fp16-format kernels have the shorter chain)
#endif
#ifndef PD_ATTN_NOSPLIT_BLOCKS
#define PD_ATTN_NOSPLIT_BLOCKS 320     // 128-query blocks from which a launch is not key-split (5 samples x 4 heads x 2048 atoms: 109.6 -> 106.2 ms per call, 7 samples 127 -> 122; at 256 blocks the split still wins)
#endif
static int attn_nsplit(const pd_attn_args* a) {
#ifdef PD_LAB
    static const int on = [] { const char* e = getenv("PD_ATTN_SPLIT"); return e ? atoi(e) : 1; }();
    if (!on) return 1;
#endif
    if (!a->ws) return 1;
    const long long blocks = (long long)a->nbatch * a->nheads * ((a->nq + 127) / 128);
    const int nit = (a->nk + KT - 1) / KT;
    if (blocks >= PD_ATTN_NOSPLIT_BLOCKS || nit < 8) return 1;
    long long s = 1024 / blocks;
    s = s < nit / 4 ? s : nit / 4;
    s = s < 8 ? s : 8;
    while (s > 1 && a->ws_bytes < 4ll * s * a->nbatch * a->nq * a->nheads * 34) --s;
    return s < 2 ? 1 : (int)s;
}

// attn_split.hip: the same kernel on the bf16 matrix pipe (3 x bf16 split operands)
extern "C" int pd_attention_split_try(const pd_attn_args* a, void* stream, int init_only);
// attn_f16.hip: key-split launch of the fp16-parts kernel (partials in the combine kernel's format)
extern "C" int pd_attention_f16_split(const pd_attn_args* a, void* stream, int init_only);
// attn_pipe.hip: the software-pipelined form of the fp16-parts kernel (bias as the accumulator's initial value)
extern "C" int pd_attention_pipe_ok(const pd_attn_args* a);
extern "C" int pd_attention_pipe_try(const pd_attn_args* a, void* stream, int init_only);

// waves per block pd_attention uses for these arguments (= template argument of attn_kernel; for profiling)
PD_EXPORT int pd_attention_variant(const pd_attn_args* a) {
    if (!a) return PD_ERR_ARG;
    // the bf16 split-operand kernel pays off when the launch fills the chip (>= one 32-query wave per SIMD); smaller launches
    // are latency-bound and stay on the fp32-MFMA kernel (with its key-split option)
    if (!a->fp32_mfma && attn_nsplit(a) <= 1 && (long long)a->nbatch * a->nheads * ((a->nq + 31) / 32) >= PD_ATTN_MIN_WAVES)
        return (a->f16x3 ? (pd_attention_pipe_ok(a) ? 3000 : 2000) : 1000) + (a->nq > 128 ? 8 : 4);      // 2000 +: two-part fp16 operands (attn_f16.hip); 3000 +: pipelined (attn_pipe.hip)
#ifdef PD_LAB
    static const int wide = [] { const char* e = getenv("PD_ATTN_WIDE"); return e ? atoi(e) : 1; }();
#else
    constexpr int wide = 1;
#endif
    // 8-wave blocks pay off (+2 %) when they still fill the chip twice over; short query ranges / few batches keep 4 waves
    const int ns = attn_nsplit(a);
    if (ns > 1)                                                        // split launch: 4-wave blocks, ns key chunks;
        return 4 + 100 * ns + ((a->f16x3 && !a->fp32_mfma && !a->O2 && !a->K2) ? 2000 : 0);      // 2000 +: on the fp16-parts kernel
    return (wide && a->nq >= 512 && (long long)a->nbatch * a->nheads * ((a->nq + 255) / 256) >= 1024) ? 8 : 4;
}

// Tail round of a pipelined launch (see the header): samples of a last round of 256-query blocks that would fill at most half of the
// 512 block slots (two blocks per CU).  Their attention runs key-split on attn_parts_kernel<4, 2, ., true> + attn_combine_kernel.
#ifndef PD_ATTN_TAIL
#define PD_ATTN_TAIL 0      // measured: no gain (round 5, NOTES.md: a quarter-full last round costs ~0.15 of a round, not one) - lab knob
#endif
static int attn_tail(const pd_attn_args* a, int* nsplit) {
    if (!PD_ATTN_TAIL || !a->ws || a->nq <= 128) return 0;
    const int bps = a->nheads * ((a->nq + 255) / 256);            // blocks per sample
    if (bps > 256 || 512 % bps) return 0;
    const int per = 512 / bps;                                    // samples per full round
    const int bt = a->nbatch % per;
    if (a->nbatch < per || bt == 0 || bt * bps > 256) return 0;
    const int nit = (a->nk + KT - 1) / KT;
    int s = 512 / (bt * bps);
    s = s < 4 ? s : 4;
    s = s < nit / 4 ? s : nit / 4;
    while (s > 1 && a->ws_bytes < 4ll * s * bt * a->nq * a->nheads * 34) --s;
    if (s < 2) return 0;
    *nsplit = s;
    return bt;
}

PD_EXPORT int pd_attention_tail(const pd_attn_args* a, int* nsplit) {
    int ns = 0;
    if (!a || pd_attention_variant(a) < 3000) return 0;
    const int bt = attn_tail(a, &ns);
    if (nsplit) *nsplit = bt ? ns : 0;
    return bt;
}

PD_EXPORT int pd_attention(const pd_attn_args* a, void* stream) {
    if (!a || !a->Q || ((!a->K || !a->V) && !(a->K2 && a->V2)) || (!a->O && !a->O2) || (!a->K2) != (!a->V2)) return PD_ERR_ARG;
    if (a->nq <= 0 || a->nk <= 0 || a->nbatch <= 0 || a->nheads <= 0) return PD_ERR_ARG;
    // 16-byte vector access on every row start
    const long long strides[] = {a->q_bs, a->q_ss, a->k_bs, a->k_ss, a->v_bs, a->v_ss, a->o_bs, a->o_ss};
    for (long long s : strides) if (s % 4) return PD_ERR_UNSUPPORTED;
    if (((uintptr_t)a->Q | (uintptr_t)a->K | (uintptr_t)a->V | (uintptr_t)a->O | (uintptr_t)a->bias) & 15)
        return PD_ERR_UNSUPPORTED;
    const int variant = pd_attention_variant(a);
    // A bias whose producer folded the operand-scale product into it (bias_prescale != 1) is only meaningful to the pipelined kernel:
    // any other kernel would consume it as an unscaled bias and return a wrong softmax silently - refuse instead.
    if (a->bias && a->bias_prescale > 0.f && a->bias_prescale != 1.f && variant < 3000) return PD_ERR_ARG;
    if ((a->O2 || a->K2) && (variant < 2000 || variant % 1000 > 100)) return PD_ERR_UNSUPPORTED;   // only the unsplit fp16-parts kernel writes the split output / reads pre-split K, V
    if (variant >= 2000 && variant % 1000 > 100) {       // key-split launch on the fp16-parts kernel + the shared combine kernel
        if (((uintptr_t)a->ws & 15) != 0) return PD_ERR_UNSUPPORTED;
        pd_attn_args s = *a;
        s.nsplit = (variant % 1000) / 100;
        const int r = pd_attention_f16_split(&s, stream, 0);
        if (r != PD_OK) return r;
        const long long total = (long long)a->nbatch * a->nq * a->nheads * 8;
        hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, s);
        return pd_check_launch();
    }
    if (variant >= 3000) {
        int ns = 0;
        const int bt = attn_tail(a, &ns);
        if (bt == 0) return pd_attention_pipe_try(a, stream, 0);
        if (((uintptr_t)a->ws & 15) != 0) return PD_ERR_UNSUPPORTED;
        // full rounds on the pipelined kernel, the samples of the last one key-split: same buffers, disjoint sample ranges
        const int n1 = a->nbatch - bt;
        const long long C = (long long)a->nheads * 32, rows = (long long)a->nbatch * a->nq;
        pd_attn_args m = *a;
        m.nbatch = n1;
        m.o2_rows = a->O2 ? rows : 0;
        int r = pd_attention_pipe_try(&m, stream, 0);
        if (r != PD_OK) return r;
        pd_attn_args t = *a;
        t.nbatch = bt; t.nsplit = ns; t.o2_rows = a->O2 ? rows : 0;
        t.Q = a->Q + n1 * a->q_bs;
        if (a->K) t.K = a->K + n1 * a->k_bs;
        if (a->V) t.V = a->V + n1 * a->v_bs;
        if (a->O) t.O = a->O + n1 * a->o_bs;
        if (a->O2) t.O2 = reinterpret_cast<unsigned short*>(a->O2) + n1 * a->nq * C;
        if (a->K2) {
            t.K2 = reinterpret_cast<const unsigned short*>(a->K2) + n1 * a->kv2_bs;
            t.V2 = reinterpret_cast<const unsigned short*>(a->V2) + n1 * a->kv2_bs;
        }
        r = pd_attention_f16_split(&t, stream, 0);
        if (r != PD_OK) return r;
        const long long total = (long long)bt * a->nq * a->nheads * 8;
        hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, t);
        return pd_check_launch();
    }
    if (variant >= 1000) return pd_attention_split_try(a, stream, 0);
    if (variant > 100) {
        if (((uintptr_t)a->ws & 15) != 0) return PD_ERR_UNSUPPORTED;
        pd_attn_args s = *a;
        s.nsplit = variant / 100;
        dim3 grid(a->nbatch, ((a->nq + 127) / 128) * s.nsplit, a->nheads);
        hipLaunchKernelGGL((attn_kernel<4, true>), grid, dim3(256), 0, (hipStream_t)stream, s);
        const long long total = (long long)a->nbatch * a->nq * a->nheads * 8;
        hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, s);
    } else if (variant == 8) {
        dim3 grid(a->nbatch, (a->nq + 255) / 256, a->nheads);
        hipLaunchKernelGGL((attn_kernel<8, false>), grid, dim3(512), 0, (hipStream_t)stream, *a);
    } else {
        dim3 grid(a->nbatch, (a->nq + 127) / 128, a->nheads);
        hipLaunchKernelGGL((attn_kernel<4, false>), grid, dim3(256), 0, (hipStream_t)stream, *a);
    }
    return pd_check_launch();
}

#ifdef PD_LAB
extern "C" __attribute__((visibility("default"))) int pd_lab_set_attn_trace(void* buf) {
    unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_trace), &p, sizeof(p)) == hipSuccess ? PD_OK : PD_ERR_LAUNCH;
}
#endif

// resident blocks per CU the runtime computes for the kernel (diagnostic, tools/attn_trace.py)
extern "C" int pd_attention_init(void) {
    const int r = pd_attention_pipe_try(nullptr, nullptr, 1);
    return r != PD_OK ? r : pd_attention_split_try(nullptr, nullptr, 1);
}

PD_EXPORT int pd_attention_occupancy(void) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (attn_kernel<4, false>), 256, 0) != hipSuccess) return PD_ERR_LAUNCH;
    return n;
}
