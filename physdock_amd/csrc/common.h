// Shared device helpers for the PhysDock sampler kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PD_EXPORT extern "C" __attribute__((visibility("default")))

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// error codes of the C ABI (include/physdock_hip.h)
enum { PD_OK = 0, PD_ERR_ARG = -1, PD_ERR_LAUNCH = -2, PD_ERR_UNSUPPORTED = -3 };

static inline int pd_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PD_OK : PD_ERR_LAUNCH;
}

#define PD_LOG2E 1.4426950408889634f

__device__ __forceinline__ float pd_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float pd_silu(float x) { return x / (1.0f + __expf(-x)); }
// The same on the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division, which hipcc expands into ten VALU instructions
// (v_div_scale x 2, v_rcp, four fma, v_div_fmas, v_div_fixup).  Used where it was measured to pay (round 5, tools/ab_silu.sh): the SwiGLU
// epilogue of the GEMM kernels (token up-projection 130.2 -> 126.9 us), the fused pool, the atom-pair FFN (921 -> 785 us).  NOT used in
// transition_f16.hip: there the reciprocal form compiles to 164 instead of 200 registers and a schedule that runs 11 % SLOWER (153.8 ->
// 170.7 us at 64 samples, 34 -> 52 us at one).  exp(-x) = inf (x < -88.7) gives rcp = 0 and silu = -0, as the division does.
__device__ __forceinline__ float pd_sigmoid_r(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float pd_silu_r(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// activation ids shared by prologues/epilogues
enum { PD_ACT_NONE = 0, PD_ACT_SILU = 1, PD_ACT_SIGMOID = 2, PD_ACT_RELU = 3 };

__device__ __forceinline__ float pd_act(float x, int act) {
    switch (act) {
        case PD_ACT_SILU: return pd_silu(x);
        case PD_ACT_SIGMOID: return pd_sigmoid(x);
        case PD_ACT_RELU: return fmaxf(x, 0.0f);
        default: return x;
    }
}

// activation on a float4 with ONE uniform branch (keeps unrolled callers compact)
__device__ __forceinline__ void pd_act4(f32x4& v, int act) {
    if (act == PD_ACT_NONE) return;
    if (act == PD_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
    } else if (act == PD_ACT_SILU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = pd_silu(v[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = pd_sigmoid(v[e]);
    }
}

// row index inside a 32x32 MFMA C fragment: lane half hh (=lane>>5), register r (0..15)
// value held by lane ^ 32 combined with the own value, without the LDS round trip of ds_bpermute: gfx950's
// v_permlane32_swap exchanges the upper half of one register with the lower half of another
__device__ __forceinline__ float pd_xhalf_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float pd_xhalf_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Sum over the 32 lanes of each wave half, result in every lane, on the VALU only: four v_add_f32_dpp (quad_perm x2,
// row_half_mirror, row_mirror) + one v_permlane16_swap.  __shfl_xor lowers to ds_bpermute (LDS crossbar + index math).
template <int CTRL>
__device__ __forceinline__ float pd_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float pd_half_sum32(float v) {
    float s = v + pd_dpp<0xB1>(v);          // quad_perm [1,0,3,2]
    s += pd_dpp<0x4E>(s);                   // quad_perm [2,3,0,1]
    s += pd_dpp<0x141>(s);                  // row_half_mirror: the other quad of an 8-lane group
    s += pd_dpp<0x140>(s);                  // row_mirror: the other 8-lane group of a 16-lane row
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);      // the other 16-lane row of the half
}

// Error-free 3-way bf16 split of TWO floats at a time: a = h + m + l per value, each part's pair packed in one dword
// (low half = first value) - the form v_cvt_pk_bf16_f32 produces and the bf16 MFMA fragments consume.  Working on pairs
// keeps it at 11 VALU operations per two values (3 packed conversions, 4 re-expansions, 4 subtractions).
typedef float pd_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pd_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pd_cvt_pk_bf16(float a, float b) {
    const pd_f32x2 v = {a, b};
    const pd_bf16x2 r = __builtin_convertvector(v, pd_bf16x2);
    return __builtin_bit_cast(unsigned, r);
}
struct pd_parts { unsigned h, m, l; };
__device__ __forceinline__ pd_parts pd_split2(float a, float b) {
    pd_parts r;
    r.h = pd_cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(r.h << 16), rb = b - __uint_as_float(r.h & 0xffff0000u);
    r.m = pd_cvt_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(r.m << 16), sb = rb - __uint_as_float(r.m & 0xffff0000u);
    r.l = pd_cvt_pk_bf16(sa, sb);
    return r;
}

// Two-way fp16 split of TWO floats at a time: a = h + l with h = fp16(a) (round to nearest even) and l = fp16(a - h); the
// residual a - h is exact in fp32, so h + l carries 22 significand bits of a as long as |a| < 65504 and l is a normal fp16
// number (|a| >= 2^-3; below that the error is bounded by 2^-25 absolute).  3 VALU operations per two values (one packed
// conversion + two mixed-precision fma) against 11 for the three-way bf16 split.  Callers scale by a power of
// two first (pd_pow2_scale) so that the operand's largest magnitude sits just below 2^15.
typedef _Float16 pd_f16x2 __attribute__((ext_vector_type(2)));
struct pd_parts2 { unsigned h, l; };
__device__ __forceinline__ pd_parts2 pd_split2h(float a, float b) {
    pd_parts2 r;
    const pd_f32x2 v = {a, b};
    const pd_f16x2 h = __builtin_convertvector(v, pd_f16x2);          // v_cvt_pk_f16_f32
    r.h = __builtin_bit_cast(unsigned, h);
#ifndef PD_SPLIT2H_PLAIN
    // residual and its conversion in ONE instruction per value: v_fma_mix{lo,hi}_f16 reads the fp16 half as an fp32 operand,
    // computes fma(h, -1, a) = a - h (exact in fp32) and rounds it to fp16 into the low / high half of the destination -
    // three instructions per pair instead of six (two re-expansions, a subtraction pair, a packed conversion).  hipcc has no
    // pattern that forms these from C (it re-expands and subtracts), hence the asm; plain VALU, no memory, no extra hazards.
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(r.l) : "v"(r.h), "v"(a), "v"(b));
#else
    const pd_f32x2 res = {a - (float)h[0], b - (float)h[1]};
    const pd_f16x2 l = __builtin_convertvector(res, pd_f16x2);
    r.l = __builtin_bit_cast(unsigned, l);
#endif
    return r;
}
// power of two s such that amax * s lies in [2^14, 2^15) (amax = f 2^(e-127), f in [1, 2)  ->  s = 2^(14 - (e - 127))): the
// largest power-of-two scaling that keeps a value bounded by amax below fp16's 65504; scales are capped to [2^-59, 2^54] so
// that zero / tiny / huge bounds stay finite
__device__ __forceinline__ float pd_pow2_scale(float amax) {
    int e = (__float_as_int(amax) >> 23) & 0xff;
    e = e < 87 ? 87 : (e > 200 ? 200 : e);
    return __int_as_float((268 - e) << 23);
}

__device__ __forceinline__ int pd_frag_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
