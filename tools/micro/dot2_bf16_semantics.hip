#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* a, const unsigned* b, const float* c, float* o, int n) {
    int i = threadIdx.x; if (i >= n) return;
    o[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a[i]), __builtin_bit_cast(bf16x2_t, b[i]), c[i], false);
}
int main() {
    // bf16: 2.0 = 0x4000, 3.0 = 0x4040, 1.0 = 0x3f80, -1.0 = 0xbf80
    unsigned A[5] = {0x40404000u, 0x40404000u, 0x40404000u, 0xb91c39b4u, 0xb91c39b4u};   // low = 2.0, high = 3.0; then the mids of (0.840188, 0.394383)
    unsigned B[5] = {0x00003f80u, 0x3f800000u, 0x0000bf80u, 0xbf800000u, 0x0000bf80u};
    float Cc[5] = {10, 10, 10, -0.00014825165f, 0.00034803152f}, O[5];
    unsigned *da, *db; float *dc, *dd;
    hipMalloc(&da, 20); hipMalloc(&db, 20); hipMalloc(&dc, 20); hipMalloc(&dd, 20);
    hipMemcpy(da, A, 20, hipMemcpyHostToDevice); hipMemcpy(db, B, 20, hipMemcpyHostToDevice); hipMemcpy(dc, Cc, 20, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dc, dd, 5);
    hipMemcpy(O, dd, 20, hipMemcpyDeviceToHost);
    for (int i = 0; i < 5; ++i) printf("A=%08x B=%08x C=%.9g -> %.9g\n", A[i], B[i], Cc[i], O[i]);
}
