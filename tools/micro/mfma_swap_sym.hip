// Micro-test (round 6): is v_mfma_f32_32x32x16_f16 symmetric under an exchange of its operands, bit for bit?
// D = A B + C (lane = column of D, registers = rows) against D' = B^T A^T + C^T (the same fragments with their roles exchanged: lane = row
// of D, registers = columns).  If the pipe adds the sixteen products and C in an order that does not depend on the operand role, D'^T == D
// exactly - a kernel can then choose the accumulator orientation that suits its epilogue without changing a bit of its results.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ int frag_row(int r, int hh) { return 8 * (r >> 2) + 4 * hh + (r & 3); }

__global__ void k(const _Float16* A, const _Float16* B, const float* C, float* D, float* Dt, int nsteps) {
    // A [32][16 nsteps] row-major (row i, k), B [32][16 nsteps] (column j, k), C [32][32]
    const int lane = threadIdx.x, l31 = lane & 31, hh = lane >> 5;
    f32x16 acc, acct;
    for (int r = 0; r < 16; ++r) { acc[r] = C[frag_row(r, hh) * 32 + l31]; acct[r] = C[l31 * 32 + frag_row(r, hh)]; }
    for (int s = 0; s < nsteps; ++s) {
        f16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = A[l31 * 16 * nsteps + 16 * s + 8 * hh + e]; b[e] = B[l31 * 16 * nsteps + 16 * s + 8 * hh + e]; }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);       // acc[r] = D[row frag_row(r, hh)][col l31]
        acct = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acct, 0, 0, 0);     // acct[r] = D[row l31][col frag_row(r, hh)]
    }
    for (int r = 0; r < 16; ++r) { D[frag_row(r, hh) * 32 + l31] = acc[r]; Dt[l31 * 32 + frag_row(r, hh)] = acct[r]; }
}

int main() {
    const int nsteps = 32, K = 16 * nsteps;
    _Float16 *hA = (_Float16*)malloc(32 * K * 2), *hB = (_Float16*)malloc(32 * K * 2);
    float hC[1024], hD[1024], hDt[1024];
    srand(1);
    int worst = 0;
    for (int trial = 0; trial < 50; ++trial) {
        for (int i = 0; i < 32 * K; ++i) {
            hA[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * (1 << (rand() % 12)));
            hB[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * (1 << (rand() % 12)));
        }
        for (int i = 0; i < 1024; ++i) hC[i] = (rand() / (float)RAND_MAX - 0.5f) * 1000.f;
        _Float16 *dA, *dB; float *dC, *dD, *dDt;
        hipMalloc(&dA, 32 * K * 2); hipMalloc(&dB, 32 * K * 2); hipMalloc(&dC, 4096); hipMalloc(&dD, 4096); hipMalloc(&dDt, 4096);
        hipMemcpy(dA, hA, 32 * K * 2, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 32 * K * 2, hipMemcpyHostToDevice); hipMemcpy(dC, hC, 4096, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, dDt, nsteps);
        hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost); hipMemcpy(hDt, dDt, 4096, hipMemcpyDeviceToHost);
        int diff = 0;
        for (int i = 0; i < 1024; ++i) diff += memcmp(&hD[i], &hDt[i], 4) != 0;
        if (diff > worst) worst = diff;
        hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD); hipFree(dDt);
    }
    printf("v_mfma_f32_32x32x16_f16, K = %d, 50 random operand sets: elements of D that differ between A.B and (B^T.A^T)^T: worst %d of 1024\n", K, worst);
    return 0;
}
