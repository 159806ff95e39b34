// Which XCD does workgroup (x, y, z) land on?  Reads HW_REG_XCC_ID (hwreg 20 on gfx94x/gfx950) per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    if (threadIdx.x == 0) {
        const unsigned v = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);     // XCC_ID[3:0]
        out[blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)] = (int)v;
    }
    __builtin_amdgcn_s_sleep(64);
}
int main() {
    const dim3 grids[] = {dim3(64, 8, 4), dim3(24, 1, 1), dim3(5, 3, 2), dim3(512, 1, 1)};
    for (const dim3& g : grids) {
        const int n = g.x * g.y * g.z;
        int* d; (void)hipMalloc(&d, n * 4);
        hipLaunchKernelGGL(k, g, dim3(256), 0, 0, d);
        int* h = new int[n]; (void)hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
        printf("grid (%d,%d,%d): first 40 linear ids -> XCC:", g.x, g.y, g.z);
        for (int i = 0; i < (n < 40 ? n : 40); ++i) printf(" %d", h[i]);
        int agree = 0; for (int i = 0; i < n; ++i) agree += (h[i] == i % 8);
        printf("\n   lin %% 8 rule holds for %d of %d workgroups\n", agree, n);
        (void)hipFree(d); delete[] h;
    }
    return 0;
}
