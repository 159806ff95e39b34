// Micro-benchmark: sustained issue rate of v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 from W waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate ; run: /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k32(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k16(float* out, int iters, float a, float b) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, int waves_per_simd, int nacc, double flop_per_mfma) {
    float* out; hipMalloc(&out, 256 * 1024 * 4 * 8);
    const int iters = 2000;
    const int blocks = 256, threads = 256 * waves_per_simd;     // one block per CU, waves_per_simd waves on each SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 10, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma_per_wave = (double)iters * 16 * nacc;
    const double total = n_mfma_per_wave * blocks * threads / 64;
    const double tf = total * flop_per_mfma / (ms * 1e-3) / 1e12;
    // cycles per MFMA per SIMD at 2.4 GHz
    const double clk = ms * 1e-3 * 2.4e9 / (n_mfma_per_wave * waves_per_simd);
    printf("%-28s waves/SIMD=%d acc=%d: %8.3f ms  %7.1f TF  %6.1f clk/MFMA/SIMD (at 2.4 GHz)\n", name, waves_per_simd, nacc, ms, tf, clk);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 4; w *= 2) {
        run("mfma_f32_32x32x2 ", k32<4>, w, 4, 4096.0);
        run("mfma_f32_32x32x2 ", k32<2>, w, 2, 4096.0);
        run("mfma_f32_32x32x2 ", k32<1>, w, 1, 4096.0);
        run("mfma_f32_16x16x4 ", k16<4>, w, 4, 2048.0);
        run("mfma_f32_16x16x4 ", k16<8>, w, 8, 2048.0);
    }
    return 0;
}
