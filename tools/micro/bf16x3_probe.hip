// Accuracy / rate probe: fp32 GEMM emulated on the bf16 matrix pipe by error-free 3-way splitting
// (a = a_hi + a_mid + a_lo, each bf16, exact for normal fp32) and 6 or 9 partial products accumulated in fp32,
// against v_mfma_f32_32x32x2_f32 and an fp64 reference.   hipcc --offload-arch=gfx950 -O3 bf16x3_probe.hip -o bf16x3_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline void split3(float a, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)a;
    const float r1 = a - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

// one wave per 32x32 output tile, operands straight from global memory (accuracy probe, not a fast kernel)
template <int NPROD>
__global__ void gemm_split(const float* A, const float* B, float* C, int M, int N, int K) {
    const int lane = threadIdx.x, l31 = lane & 31, hh = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        bf16x8 a[3], b[3];
        for (int e = 0; e < 8; ++e) {
            __bf16 h, m, l;
            split3(A[(size_t)(m0 + l31) * K + k0 + 8 * hh + e], h, m, l); a[0][e] = h; a[1][e] = m; a[2][e] = l;
            split3(B[(size_t)(n0 + l31) * K + k0 + 8 * hh + e], h, m, l); b[0][e] = h; b[1][e] = m; b[2][e] = l;
        }
        // smallest terms first
        if (NPROD == 9) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[1], acc, 0, 0, 0);
        }
        if (NPROD >= 6) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
        }
        if (NPROD >= 3) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) C[(size_t)(m0 + (r & 3) + 8 * (r >> 2) + 4 * hh) * N + n0 + l31] = acc[r];
}

__global__ void gemm_f32(const float* A, const float* B, float* C, int M, int N, int K) {
    const int lane = threadIdx.x, l31 = lane & 31, hh = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(size_t)(m0 + l31) * K + k0 + hh], B[(size_t)(n0 + l31) * K + k0 + hh], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[(size_t)(m0 + (r & 3) + 8 * (r >> 2) + 4 * hh) * N + n0 + l31] = acc[r];
}

// rate: 4 waves per block, back-to-back MFMAs on register operands
template <int WHICH>
__global__ void rate(float* out, int iters) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(1.0f + e); }
    for (int i = 0; i < iters; ++i)
        for (int j = 0; j < 4; ++j) {
            if (WHICH == 0) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
            else acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a[0], (float)b[0], acc[j], 0, 0, 0);
        }
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const int M = 256, N = 256, K = 512;
    std::vector<float> A(M * K), B(N * K);
    srand(1);
    auto rnd = [] { float u = 0; for (int i = 0; i < 12; ++i) u += rand() / (float)RAND_MAX; return u - 6.f; };
    for (auto& v : A) v = rnd() * expf(2.f * rnd());       // wide dynamic range
    for (auto& v : B) v = rnd();
    std::vector<double> ref((size_t)M * N), mag((size_t)M * N);
    for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) {
        double s = 0, a = 0;
        for (int k = 0; k < K; ++k) { const double p = (double)A[i * K + k] * B[j * K + k]; s += p; a += fabs(p); }
        ref[(size_t)i * N + j] = s; mag[(size_t)i * N + j] = a;
    }
    // fp32 sequential fma chain on the host (what the f32 MFMA is bitwise equal to, per the microarch guide)
    double e_chain = 0;
    for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) {
        float s = 0; for (int k = 0; k < K; ++k) s = fmaf(A[i * K + k], B[j * K + k], s);
        e_chain = fmax(e_chain, fabs(s - ref[(size_t)i * N + j]) / mag[(size_t)i * N + j]);
    }
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> C((size_t)M * N);
    auto report = [&](const char* name) {
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        double mx = 0, rms = 0;
        for (size_t i = 0; i < C.size(); ++i) { const double e = fabs(C[i] - ref[i]) / mag[i]; mx = fmax(mx, e); rms += e * e; }
        printf("%-28s max err / sum|a b| = %.3e   rms = %.3e   (2^-24 = 5.96e-8)\n", name, mx, sqrt(rms / C.size()));
    };
    printf("host fp32 fmaf chain         max err / sum|a b| = %.3e\n", e_chain);
    dim3 grid(M / 32, N / 32);
    gemm_f32<<<grid, 64>>>(dA, dB, dC, M, N, K); report("v_mfma_f32_32x32x2_f32");
    gemm_split<1><<<grid, 64>>>(dA, dB, dC, M, N, K); report("bf16 x1 (plain bf16)");
    gemm_split<3><<<grid, 64>>>(dA, dB, dC, M, N, K); report("bf16 x3 (hh, hm, mh)");
    gemm_split<6><<<grid, 64>>>(dA, dB, dC, M, N, K); report("bf16 x6");
    gemm_split<9><<<grid, 64>>>(dA, dB, dC, M, N, K); report("bf16 x9");
    // rates
    float* dout; hipMalloc(&dout, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        const int iters = 2000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (which == 0) rate<0><<<1024, 256>>>(dout, iters); else rate<1><<<1024, 256>>>(dout, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 1024.0 * 4 * iters * 4 * (which == 0 ? 32768.0 : 4096.0);
        printf("%s back-to-back: %.1f TFLOP/s\n", which == 0 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_32x32x2_f32 ", flops / ms / 1e9);
    }
    return 0;
}
