// Micro-benchmark: does VALU work of a SECOND wave on the same SIMD slow down a wave that issues back-to-back MFMAs?
// Block = 8 waves: waves 0-3 (one per SIMD) run 32x32x2 f32 MFMAs, waves 4-7 run `valu_per_mfma` dependent-free VALU
// instructions per MFMA of the partner (0 = idle partner).  Reports cycles per MFMA of the MFMA waves.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VALU, int KIND>
__global__ __launch_bounds__(512) void mix(float* out, long long* clk, int iters, float a, float b) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        const long long t1 = __builtin_amdgcn_s_memtime();
        out[blockIdx.x * 512 + threadIdx.x] = s;
        if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
    } else if (VALU > 0) {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = a + i + threadIdx.x;
        unsigned long long y = threadIdx.x + 12345ull;
        for (int it = 0; it < iters * 64; ++it) {       // one trip per partner MFMA
#pragma unroll
            for (int v = 0; v < VALU; ++v) {
                if (KIND == 0) x[v & 7] = x[v & 7] * 1.0001f + b;           // v_fma_f32 (full rate)
                else if (KIND == 1) y = y * 3ull + (unsigned long long)(v + it);   // 64-bit integer (mul_lo/hi, addc)
                else x[v & 7] = __builtin_amdgcn_exp2f(x[v & 7]);             // transcendental
            }
        }
        float s = (float)y;
        for (int i = 0; i < 8; ++i) s += x[i];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

template <int VALU, int KIND>
void run(const char* what) {
    float* out; long long* clk;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&clk, 256 * 8);
    const int iters = 400;
    hipLaunchKernelGGL((mix<VALU, KIND>), dim3(256), dim3(512), 0, 0, out, clk, 10, 1.f, 1.f);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((mix<VALU, KIND>), dim3(256), dim3(512), 0, 0, out, clk, iters, 1.f, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
    printf("%-34s VALU/MFMA=%2d: kernel %7.3f ms, MFMA waves %.1f memtime ticks per MFMA (x24 = %.1f clk at 100 MHz ref / 2.4 GHz)\n",
           what, VALU, ms, avg / (iters * 64.0), avg / (iters * 64.0) * 24.0);
    (void)hipFree(out); (void)hipFree(clk);
}

int main() {
    run<0, 0>("partner idle");
    run<2, 0>("partner v_fma_f32"); run<4, 0>("partner v_fma_f32"); run<8, 0>("partner v_fma_f32"); run<16, 0>("partner v_fma_f32");
    run<2, 1>("partner 64-bit int"); run<4, 1>("partner 64-bit int"); run<8, 1>("partner 64-bit int");
    run<2, 2>("partner v_exp_f32"); run<4, 2>("partner v_exp_f32");
    return 0;
}
