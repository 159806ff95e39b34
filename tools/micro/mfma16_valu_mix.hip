// Micro-benchmark (round 4): do v_mfma_f32_32x32x16_f16 instructions and fp32 VALU instructions of the SAME SIMD overlap?
// Every wave runs, per trip, 12 MFMAs (two dependent chains of 6, as a 32-key attention sub-tile does) and NV independent VALU
// instructions (v_fma_f32 on private registers), interleaved in program order NV / 12 after each MFMA.  MODE 0: MFMAs only,
// 1: VALU only, 2: both.  W waves per SIMD (1, 2, 4).  If the two pipes overlapped, T(both) ~ max(T(mfma), T(valu)); the
// attention kernel of this repo behaves like T(mfma) + T(valu) (profiles/r04_attn_pipe_ablations.txt).  The same with the fp32
// MFMA (32x32x2, 64 cycles) for comparison.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NV, bool F32>
__global__ void mix(float* out, int iters, float a, float b) {
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    f16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(a + threadIdx.x * 1e-3f); fb[i] = (_Float16)b; }
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = a + i + threadIdx.x;
    constexpr int PER = NV / 12;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            if (MODE != 1) {
                if (F32) {
                    if (u < 6) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                    else acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
                } else {
                    if (u < 6) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0);
                    else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc1, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE != 0) {
#pragma unroll
                for (int v = 0; v < PER; ++v) x[(u * PER + v) & 15] = __builtin_fmaf(x[(u * PER + v) & 15], 1.0001f, b);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + x[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NV, bool F32>
double run(int w) {
    float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 4000;
    hipLaunchKernelGGL((mix<MODE, NV, F32>), dim3(256), dim3(256 * w), 0, 0, out, 10, 1.f, 1.f);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((mix<MODE, NV, F32>), dim3(256), dim3(256 * w), 0, 0, out, iters, 1.f, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(out);
    return ms * 1e3 / (iters * (double)w);          // us per (wave trip) per SIMD ... normalised: time per trip of one wave slot
}

template <int NV, bool F32>
void row(int w) {
    const double m = run<0, NV, F32>(w), v = run<1, NV, F32>(w), both = run<2, NV, F32>(w);
    printf("%s waves/SIMD=%d  12 MFMA + %3d VALU per trip:  mfma only %7.4f us  valu only %7.4f us  both %7.4f us   (max %.4f, sum %.4f)  both/sum %.2f\n",
           F32 ? "f32 32x32x2 " : "f16 32x32x16", w, NV, m, v, both, m > v ? m : v, m + v, both / (m + v));
}

int main() {
    for (int w = 1; w <= 4; w *= 2) {
        row<48, false>(w); row<96, false>(w); row<120, false>(w); row<192, false>(w);
        row<96, true>(w); row<192, true>(w);
    }
    return 0;
}
