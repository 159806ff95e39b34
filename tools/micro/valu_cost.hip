// Micro-benchmark (round 4): issue cost of the VALU instructions the attention softmax is made of, four waves per SIMD, independent
// operands: cycles per instruction per SIMD = kernel time x clock / (instructions per wave x waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k(float* out, int iters, float a, float b) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = a + 0.01f * i + 1e-3f * threadIdx.x;
    unsigned u[16];
    for (int i = 0; i < 16; ++i) u[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int v = 0; v < 64; ++v) {
            float& r = x[v & 15];
            if (KIND == 0) r = __builtin_fmaf(r, 1.0001f, b);
            else if (KIND == 1) r = __builtin_amdgcn_exp2f(r) * 1e-30f + 0.f * b, r = r;      // exp + mul (mul cost subtracted by KIND 7)
            else if (KIND == 2) { f32x2 t = {r, x[(v + 1) & 15]}; f16x2 h = __builtin_convertvector(t, f16x2); u[v & 15] ^= __builtin_bit_cast(unsigned, h); }
            else if (KIND == 3) { unsigned d = u[v & 15]; asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(d) : "v"(u[(v + 3) & 15]), "v"(r)); u[v & 15] = d; }
            else if (KIND == 4) r = __builtin_fmaxf(__builtin_fmaxf(r, x[(v + 1) & 15]), x[(v + 2) & 15]);
            else if (KIND == 5) r = r * b;
            else if (KIND == 6) r = r + x[(v + 5) & 15];
            else if (KIND == 7) r = r * 1e-30f;
            else if (KIND == 9) {            // packed add: two row-sum additions in one instruction
                f32x2 p = {x[v & 14], x[(v & 14) + 1]}; const f32x2 q = {x[(v + 4) & 14], x[((v + 4) & 14) + 1]};
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(q));
                x[v & 14] = p[0]; x[(v & 14) + 1] = p[1];
            } else if (KIND == 10) {         // packed fma, second operand an SGPR pair, third a VGPR pair
                f32x2 p = {x[v & 14], x[(v & 14) + 1]}; const f32x2 q = {x[(v + 4) & 14], x[((v + 4) & 14) + 1]};
                const f32x2 c = {a, a};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "s"(c), "v"(q));
                x[v & 14] = p[0]; x[(v & 14) + 1] = p[1];
            } else if (KIND == 11) {         // packed fma, third operand ONE VGPR broadcast to both halves (op_sel_hi = 0 on src2)
                f32x2 p = {x[v & 14], x[(v & 14) + 1]}; const f32x2 c = {a, a};
                f32x2 q = {x[(v + 4) & 14], x[((v + 4) & 14) + 1]};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,1,0]" : "+v"(p) : "s"(c), "v"(q));
                x[v & 14] = p[0]; x[(v & 14) + 1] = p[1];
            } else if (KIND == 12) {         // packed mul by an SGPR pair
                f32x2 p = {x[v & 14], x[(v & 14) + 1]}; const f32x2 c = {b, b};
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "s"(c));
                x[v & 14] = p[0]; x[(v & 14) + 1] = p[1];
            }
            else if (KIND == 8) { const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(r), __float_as_uint(r), false, false); r = __uint_as_float(s[0]); }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += x[i] + (float)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name) {
    float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 4000, w = 4;
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(256 * w), 0, 0, out, 10, 0.5f, 0.25f);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(256 * w), 0, 0, out, iters, 0.5f, 0.25f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.3f ms  = %.2f ns per instruction per SIMD (x clock in GHz = cycles; 2.4 GHz: %.2f)\n", name, ms,
           ms * 1e6 / (iters * 64.0 * w), ms * 1e6 / (iters * 64.0 * w) * 2.4);
    (void)hipFree(out);
}

int main() {
    run<0>("v_fma_f32");
    run<5>("v_mul_f32");
    run<6>("v_add_f32");
    run<7>("v_mul_f32 (const)");
    run<1>("v_exp_f32 + v_mul_f32");
    run<2>("v_cvt_pk_f16_f32 + v_xor");
    run<3>("v_fma_mixlo_f16");
    run<4>("v_max3_f32");
    run<8>("v_permlane32_swap");
    run<9>("v_pk_add_f32 (2 adds)");
    run<10>("v_pk_fma_f32 v,s,v (2 fma)");
    run<11>("v_pk_fma_f32 v,s,v bcast");
    run<12>("v_pk_mul_f32 v,s (2 mul)");
    return 0;
}
