// Element-wise pieces of the ConfidenceModule (reference models/layers/confidence_module.py:56-88).  The
// Pairformer and AtomTransformer stacks in the middle are the trunk's own kernels (engine.py); these three passes are the
// module's entry and exit, all HBM-bound (one read + one write of a [T,T,C] / [A,A,c_ap] tensor each).
#include "common.h"
#include "physdock_hip.h"

namespace {

// z_out[i,j,:] = z[i,j,:] + si[i,:] + sj[j,:] + Wd[:, bin(|xc_i - xc_j|)]          (confidence_module.py:68-72)
// xc = x_pred[0, token_id_to_centre_atom_id]; bin = first argmin_k |d - v_k|, v = linspace(3.375, 24.375, 13) (exact in fp32:
// 3.375 + 1.75 k), i.e. one_hot_with_nearest_bin (utils/tensor_utils.py:673-686) followed by the bias-free 13 -> c_z Linear,
// which for a one-hot input is a row gather of W^T [13][C].  One thread per four channels of one pair.
__global__ __launch_bounds__(256) void confidence_pair_init_kernel(const float* __restrict__ z, const float* __restrict__ si,
                                                                  const float* __restrict__ sj, const float* __restrict__ WdT,
                                                                  const float* __restrict__ x, const long long* __restrict__ centre,
                                                                  float* __restrict__ out, int T, int C) {
    const int c4 = C >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)T * T * c4) return;
    const long long pair = idx / c4;
    const int c = (int)(idx - pair * c4) * 4;
    const int i = (int)(pair / T), j = (int)(pair - (long long)i * T);
    const long long ai = centre[i], aj = centre[j];
    const float dx = x[3 * ai] - x[3 * aj], dy = x[3 * ai + 1] - x[3 * aj + 1], dz = x[3 * ai + 2] - x[3 * aj + 2];
    const float d = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    int bin = 0;
    float best = fabsf(d - 3.375f);
#pragma unroll
    for (int k = 1; k < 13; ++k) {
        const float e = fabsf(d - (3.375f + 1.75f * (float)k));
        if (e < best) { best = e; bin = k; }
    }
    const f32x4 a = *reinterpret_cast<const f32x4*>(z + pair * C + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(si + (long long)i * C + c);
    const f32x4 e = *reinterpret_cast<const f32x4*>(sj + (long long)j * C + c);
    const f32x4 w = *reinterpret_cast<const f32x4*>(WdT + bin * C + c);
    f32x4 o;
#pragma unroll
    for (int t = 0; t < 4; ++t) o[t] = ((a[t] + b[t]) + e[t]) + w[t];      // the reference's order of the three additions
    *reinterpret_cast<f32x4*>(out + pair * C + c) = o;
}

// out[i,j,:] = z[i,j,:] + z[j,i,:]                                                  (confidence_module.py:75)
__global__ __launch_bounds__(256) void pair_symmetrize_kernel(const float* __restrict__ z, float* __restrict__ out, int T, int C) {
    const int c4 = C >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)T * T * c4) return;
    const long long pair = idx / c4;
    const int c = (int)(idx - pair * c4) * 4;
    const int i = (int)(pair / T), j = (int)(pair - (long long)i * T);
    const f32x4 a = *reinterpret_cast<const f32x4*>(z + pair * C + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(z + ((long long)j * T + i) * C + c);
    *reinterpret_cast<f32x4*>(out + pair * C + c) = a + b;
}

// ap[i,j,:] = |x_i - x_j| w + b                                                      (confidence_module.py:80, Linear(1, c_ap))
__global__ __launch_bounds__(256) void atom_dist_embed_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, float* __restrict__ ap, int A, int C) {
    const int c4 = C >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)A * A * c4) return;
    const long long pair = idx / c4;
    const int c = (int)(idx - pair * c4) * 4;
    const int i = (int)(pair / A), j = (int)(pair - (long long)i * A);
    const float dx = x[3 * j] - x[3 * i], dy = x[3 * j + 1] - x[3 * i + 1], dz = x[3 * j + 2] - x[3 * i + 2];
    const float d = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    f32x4 o;
#pragma unroll
    for (int t = 0; t < 4; ++t) o[t] = __fadd_rn(__fmul_rn(d, w[c + t]), b ? b[c + t] : 0.f);
    *reinterpret_cast<f32x4*>(ap + pair * C + c) = o;
}

inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

PD_EXPORT int pd_confidence_pair_init(const float* z, const float* si, const float* sj, const float* WdT, const float* x,
                                      const long long* centre, float* out, int T, int C, void* stream) {
    if (!z || !si || !sj || !WdT || !x || !centre || !out || T <= 0 || C <= 0) return PD_ERR_ARG;
    if (C % 4) return PD_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(confidence_pair_init_kernel, dim3(blocks_for((long long)T * T * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, z, si, sj, WdT, x, centre, out, T, C);
    return pd_check_launch();
}

PD_EXPORT int pd_pair_symmetrize(const float* z, float* out, int T, int C, void* stream) {
    if (!z || !out || z == out || T <= 0 || C <= 0) return PD_ERR_ARG;
    if (C % 4) return PD_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pair_symmetrize_kernel, dim3(blocks_for((long long)T * T * (C / 4))), dim3(256), 0, (hipStream_t)stream, z,
                       out, T, C);
    return pd_check_launch();
}

PD_EXPORT int pd_atom_dist_embed(const float* x, const float* w, const float* b, float* ap, int A, int C, void* stream) {
    if (!x || !w || !ap || A <= 0 || C <= 0) return PD_ERR_ARG;
    if (C % 4) return PD_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(atom_dist_embed_kernel, dim3(blocks_for((long long)A * A * (C / 4))), dim3(256), 0, (hipStream_t)stream, x,
                       w, b, ap, A, C);
    return pd_check_launch();
}
