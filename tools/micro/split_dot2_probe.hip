// Is the residual of the 3-way bf16 split computable with v_dot2_f32_bf16 (r = a - bf16(a) as (-1, 0) . (hi_a, hi_b) + a)
// bit for bit like the shift / mask / subtract sequence?  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -I physdock_amd/csrc -I include tools/micro/split_dot2_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "common.h"

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

// hipcc (ROCm 7.2) folds a CONSTANT selector pair such as 0x0000bf80 into the inline operand "-1.0" of v_dot2c_f32_bf16, which
// the hardware reads as the fp32 pattern 0xbf800000 = (0, -1): the wrong half (tools/micro/dot2_bf16_semantics.hip).  The
// selectors are therefore made opaque (SGPRs).
__device__ __forceinline__ pd_parts split2_dot(float a, float b) {
    unsigned lo = 0x0000bf80u, hi = 0xbf800000u;      // -1 times the low / the high half
    asm volatile("" : "+s"(lo), "+s"(hi));
    const bf16x2_t lo_sel = __builtin_bit_cast(bf16x2_t, lo), hi_sel = __builtin_bit_cast(bf16x2_t, hi);
    pd_parts r;
    r.h = pd_cvt_pk_bf16(a, b);
    const bf16x2_t h = __builtin_bit_cast(bf16x2_t, r.h);
    const float ra = __builtin_amdgcn_fdot2_f32_bf16(h, lo_sel, a, false), rb = __builtin_amdgcn_fdot2_f32_bf16(h, hi_sel, b, false);
    r.m = pd_cvt_pk_bf16(ra, rb);
    const bf16x2_t m = __builtin_bit_cast(bf16x2_t, r.m);
    const float sa = __builtin_amdgcn_fdot2_f32_bf16(m, lo_sel, ra, false), sb = __builtin_amdgcn_fdot2_f32_bf16(m, hi_sel, rb, false);
    r.l = pd_cvt_pk_bf16(sa, sb);
    return r;
}

__global__ void probe(const float* x, unsigned* ref, unsigned* dot, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const pd_parts p = pd_split2(x[2 * i], x[2 * i + 1]), q = split2_dot(x[2 * i], x[2 * i + 1]);
    ref[3 * i] = p.h; ref[3 * i + 1] = p.m; ref[3 * i + 2] = p.l;
    dot[3 * i] = q.h; dot[3 * i + 1] = q.m; dot[3 * i + 2] = q.l;
}

int main() {
    const int n = 1 << 22;
    std::vector<float> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        const int kind = i % 8;
        const float u = (float)rand() / RAND_MAX;
        if (kind < 3) h[i] = u;                                              // probabilities
        else if (kind < 5) h[i] = (u - 0.5f) * 60.f;                         // activations
        else if (kind == 5) h[i] = ldexpf(u + 0.5f, -(rand() % 140));        // tiny (down into the fp32 denormals)
        else if (kind == 6) h[i] = -ldexpf(u + 0.5f, (rand() % 40) - 20);
        else { unsigned b = ((unsigned)rand() << 16) ^ (unsigned)rand(); b &= 0xbfffffffu; memcpy(&h[i], &b, 4); if (!std::isfinite(h[i])) h[i] = 1.f; }
    }
    float* dx; unsigned *dr, *dd;
    hipMalloc(&dx, n * 4); hipMalloc(&dr, n / 2 * 12); hipMalloc(&dd, n / 2 * 12);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    probe<<<n / 2 / 256, 256>>>(dx, dr, dd, n);
    std::vector<unsigned> r(n / 2 * 3), d(n / 2 * 3);
    hipMemcpy(r.data(), dr, r.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost);
    long long bad = 0, bad_normal = 0;
    for (size_t i = 0; i < r.size(); ++i)
        if (r[i] != d[i]) {
            ++bad;
            const float a = h[2 * (i / 3)], b = h[2 * (i / 3) + 1];
            if (fabsf(a) > 1e-30f && fabsf(b) > 1e-30f) { if (bad_normal < 5) printf("mismatch part %zu of (%g, %g): %08x vs %08x\n", i % 3, a, b, r[i], d[i]); ++bad_normal; }
        }
    printf("pairs %d: mismatching part words %lld (of which with both values above 1e-30: %lld)\n", n / 2, bad, bad_normal);
    return 0;
}
