// Pair-representation and pooling kernels of the conditioning trunk / DiT glue.
// All HBM-bound streaming or gather kernels: coalesced float4 traffic, no MFMA.
#include "common.h"
#include "physdock_hip.h"

namespace {

// ap[l,m,:] = cl[l,:] + cm[m,:] + v*(Wp.d + Wd/(1+|d|) + Wv),  d = pos_l - pos_m, v = [uid_l == uid_m]
// reference: layers/diffusion_conditioning.py:116-124 (before the pair FFN).
template <int CAP>
__global__ __launch_bounds__(256) void atom_pair_init_kernel(const float* __restrict__ pos, const long long* __restrict__ uid,
                                                            const float* __restrict__ cl, const float* __restrict__ cm,
                                                            const float* __restrict__ Wp, const float* __restrict__ Wd,
                                                            const float* __restrict__ Wv, float* __restrict__ ap, int A) {
    __shared__ float sWp[CAP * 3], sWd[CAP], sWv[CAP];
    for (int i = threadIdx.x; i < CAP * 3; i += 256) sWp[i] = Wp[i];
    for (int i = threadIdx.x; i < CAP; i += 256) { sWd[i] = Wd[i]; sWv[i] = Wv[i]; }
    __syncthreads();
    const int l = blockIdx.y;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= A) return;
    const float dx = pos[3 * l] - pos[3 * m], dy = pos[3 * l + 1] - pos[3 * m + 1], dz = pos[3 * l + 2] - pos[3 * m + 2];
    const float v = uid[l] == uid[m] ? 1.f : 0.f;
    const float inv = 1.f / (1.f + sqrtf(dx * dx + dy * dy + dz * dz));
    float* out = ap + ((long long)l * A + m) * CAP;
#pragma unroll
    for (int c4 = 0; c4 < CAP / 4; ++c4) {
        f32x4 o;
        const f32x4 a = *reinterpret_cast<const f32x4*>(cl + (long long)l * CAP + c4 * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(cm + (long long)m * CAP + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = c4 * 4 + e;
            float p = (sWp[c * 3] * dx + sWp[c * 3 + 1] * dy + sWp[c * 3 + 2] * dz) * v;
            p = p + (sWd[c] * inv) * v;
            p = p + (sWv[c] * v) * v;
            o[e] = (a[e] + b[e]) + p;
        }
        *reinterpret_cast<f32x4*>(out + c4 * 4) = o;
    }
}

// ap[l,m,:] += zt[a2t[l], a2t[m], :]       (diffusion_conditioning.py:237)
template <int CAP>
__global__ __launch_bounds__(256) void pair_gather_add_kernel(float* __restrict__ ap, const float* __restrict__ zt,
                                                             const long long* __restrict__ a2t, int A, int T) {
    const int l = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;          // (m, chunk)
    const int m = idx / (CAP / 4), c4 = idx % (CAP / 4);
    if (m >= A) return;
    const long long src = ((long long)a2t[l] * T + a2t[m]) * CAP + c4 * 4;
    float* dst = ap + ((long long)l * A + m) * CAP + c4 * 4;
    f32x4 v = *reinterpret_cast<f32x4*>(dst);
    v += *reinterpret_cast<const f32x4*>(zt + src);
    *reinterpret_cast<f32x4*>(dst) = v;
}

// z[i,j,:] = si[i,:] + sj[j,:] + RelPos(i,j) + bonds[i,j]*wb      (diffusion_conditioning.py:65-94,187-189)
// RelPos = W[:, d_res] + W[:, 66:108] . rel_tok_feat[i,j,:] + W[:,108]*same_entity + W[:, 109 + d_chain]
// WT is the RelPos weight transposed to [115][CZ].  One block per row i, threads over channels.
__global__ __launch_bounds__(256) void pair_init_z_kernel(const float* __restrict__ si, const float* __restrict__ sj,
                                                         const float* __restrict__ WT, const float* __restrict__ wb,
                                                         const int* __restrict__ asym, const int* __restrict__ sym,
                                                         const int* __restrict__ ent, const long long* __restrict__ res,
                                                         const float* __restrict__ rtf, const float* __restrict__ bonds,
                                                         float* __restrict__ z, int T, int CZ) {
    extern __shared__ float sW[];                            // [42][CZ] slice of WT
    for (int i = threadIdx.x; i < 42 * CZ; i += blockDim.x) sW[i] = WT[66 * CZ + i];
    __syncthreads();
    const int i = blockIdx.x;
    const int jchunk = (T + gridDim.y - 1) / gridDim.y;      // the keys of a row are split over blockIdx.y (more blocks than CUs)
    const int jend = (blockIdx.y + 1) * jchunk < T ? (blockIdx.y + 1) * jchunk : T;
    const int ai = asym[i], si_ = sym[i], ei = ent[i];
    const long long ri = res[i];
    for (int c = threadIdx.x; c < CZ; c += blockDim.x) {
        const float base = si[(long long)i * CZ + c];
        const float wbc = wb[c];
        for (int j = blockIdx.y * jchunk; j < jend; ++j) {
            const bool chain_same = ai == asym[j], ent_same = ei == ent[j];
            long long dr = ri - res[j] + 32;
            dr = dr < 0 ? 0 : (dr > 64 ? 64 : dr);
            const int d_res = chain_same ? (int)dr : 65;
            int dc = si_ - sym[j] + 2;
            dc = dc < 0 ? 0 : (dc > 4 ? 4 : dc);
            const int d_chain = (chain_same || !ent_same) ? 5 : dc;
            float acc = WT[d_res * CZ + c];
            const float* f = rtf + ((long long)i * T + j) * 42;
#pragma unroll 6
            for (int k = 0; k < 42; ++k) acc += f[k] * sW[k * CZ + c];
            acc += ent_same ? WT[108 * CZ + c] : 0.f;
            acc += WT[(109 + d_chain) * CZ + c];
            float v = (base + sj[(long long)j * CZ + c]) + acc;
            v += bonds[(long long)i * T + j] * wbc;
            z[((long long)i * T + j) * CZ + c] = v;
        }
    }
}

// out[b,t,:] = sum_{atoms of t} u[b,atom,:] / (n_t + 1e-3) [+ add[t,:]]     (transformers.py:205-212)
// one thread per (b, t, 16-byte channel chunk); loads are issued 8 atoms at a time (clamped row, zero
// weight beyond the token) so every lane keeps 8 independent 16-byte loads in flight.
__global__ __launch_bounds__(256) void segment_pool_kernel(const float* __restrict__ u, const int* __restrict__ tok_start,
                                                          const float* __restrict__ add, float* __restrict__ out,
                                                          int A, int T, int C, long long n4) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n4) return;
    const int c4 = idx % (C / 4);
    const long long bt = idx / (C / 4);
    const int t = bt % T;
    const long long b = bt / T;
    const int s = tok_start[t], e = tok_start[t + 1];
    const float* base = u + (b * A) * C + c4 * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int a0 = s; a0 < e; a0 += 8) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int a = a0 + k < e ? a0 + k : e - 1;
            v[k] = *reinterpret_cast<const f32x4*>(base + (long long)a * C);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) if (a0 + k < e) acc += v[k];
    }
    acc *= 1.f / ((float)(e - s) + 1e-3f);
    if (add) acc += *reinterpret_cast<const f32x4*>(add + (long long)t * C + c4 * 4);
    *reinterpret_cast<f32x4*>(out + bt * C + c4 * 4) = acc;
}

// ba[b,l,:] += us[b, a2t[l], :]                                            (transformers.py:214-216)
// I: the index type - unsigned 32-bit whenever the element count allows (as pd_precond: 64-bit divisions cost ~300 instructions per 16 bytes)
template <typename I>
__global__ __launch_bounds__(256) void unpool_add_kernel(float* __restrict__ ba, const float* __restrict__ us,
                                                        const long long* __restrict__ a2t, int A, int T, int C, long long n4) {
    const I idx = (I)blockIdx.x * 256 + threadIdx.x;
    if ((long long)idx >= n4) return;
    const I nc4 = (I)(C / 4), row = idx / nc4;               // b*A + l
    const int c4 = (int)(idx - row * nc4);
    const I b = row / (I)A;
    const int l = (int)(row - b * (I)A);
    f32x4 v = *reinterpret_cast<f32x4*>(ba + (long long)row * C + c4 * 4);
    v += *reinterpret_cast<const f32x4*>(us + ((long long)b * T + a2t[l]) * C + c4 * 4);
    *reinterpret_cast<f32x4*>(ba + (long long)row * C + c4 * 4) = v;
}

// y[r,:] += x[idx[r],:]  /  y = a + b helpers
__global__ __launch_bounds__(256) void gather_rows_add_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                             const long long* __restrict__ idx, int R, int C) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)R * (C / 4)) return;
    const int c4 = i % (C / 4);
    const long long r = i / (C / 4);
    f32x4 v = *reinterpret_cast<f32x4*>(y + r * C + c4 * 4);
    v += *reinterpret_cast<const f32x4*>(x + idx[r] * C + c4 * 4);
    *reinterpret_cast<f32x4*>(y + r * C + c4 * 4) = v;
}

// out = a*sa [+ b*sb]  (elementwise, float4) - used for z += template embedding * t_mask etc.
__global__ __launch_bounds__(256) void axpby_kernel(float* __restrict__ out, const float* __restrict__ a, float sa,
                                                   const float* __restrict__ b, const float* __restrict__ sb_ptr, float sb,
                                                   long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const float sbv = sb_ptr ? sb_ptr[0] * sb : sb;
    if (4 * i + 3 < n) {
        f32x4 v = reinterpret_cast<const f32x4*>(a)[i] * sa;
        if (b) v += reinterpret_cast<const f32x4*>(b)[i] * sbv;
        reinterpret_cast<f32x4*>(out)[i] = v;
    } else {
        for (long long k = 4 * i; k < n; ++k) out[k] = a[k] * sa + (b ? b[k] * sbv : 0.f);      // tail of an n that is not a multiple of 4
    }
}

// mask2d[i,j] = z_mask[i,j] * templ_feat[i,j,39] * [asym_i == asym_j]     (diffusion_conditioning.py:41-42)
__global__ __launch_bounds__(256) void template_mask_kernel(const float* __restrict__ z_mask, const float* __restrict__ templ,
                                                           const int* __restrict__ asym, float* __restrict__ out, int T, int D) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)T * T) return;
    const int i = idx / T, j = idx % T;
    out[idx] = z_mask[idx] * templ[idx * D + (D - 1)] * (asym[i] == asym[j] ? 1.f : 0.f);
}

// templ_feat[i,j,:] = [ 39-bin distogram of the pseudo-beta distance | mask ], mask = z_mask * protein_i * protein_j
// (feature_loader.py:944-968 get_template_feat, inference branch; utils/tensor_utils.py:689-703 dgram_from_positions:
// bin b is set when lower[b] < d^2 < upper[b], both strict, lower = linspace(3.25, 50.75, 39)^2 supplied by the host so that
// the boundaries are the reference's own fp32 values; d^2 = (dx^2 + dy^2) + dz^2 without fused multiply-adds)
__global__ __launch_bounds__(256) void template_feat_kernel(const float* __restrict__ x, const long long* __restrict__ pb,
                                                           const float* __restrict__ z_mask, const float* __restrict__ is_prot,
                                                           const float* __restrict__ lower, float* __restrict__ out, int T,
                                                           int NB, float inf) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)T * T) return;
    const int i = (int)(idx / T), j = (int)(idx - (long long)i * T);
    const long long ai = pb[i], aj = pb[j];
    const float dx = x[3 * ai] - x[3 * aj], dy = x[3 * ai + 1] - x[3 * aj + 1], dz = x[3 * ai + 2] - x[3 * aj + 2];
    const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    const float m = z_mask[idx] * (is_prot[i] * is_prot[j]);
    float* o = out + idx * (NB + 1);
    for (int b = 0; b < NB; ++b) {
        const float up = b + 1 < NB ? lower[b + 1] : inf;
        const float bit = (d2 > lower[b] && d2 < up) ? 1.f : 0.f;
        o[b] = bit * m;
    }
    o[NB] = m;
}

}  // namespace

PD_EXPORT int pd_template_feat(const float* x, const long long* pseudo_beta_atom, const float* z_mask, const float* is_protein,
                               const float* lower, float* out, int T, int no_bins, void* stream) {
    if (!x || !pseudo_beta_atom || !z_mask || !is_protein || !lower || !out || T <= 0 || no_bins <= 0) return PD_ERR_ARG;
    const long long n = (long long)T * T;
    hipLaunchKernelGGL(template_feat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       pseudo_beta_atom, z_mask, is_protein, lower, out, T, no_bins, 1e8f);
    return pd_check_launch();
}

PD_EXPORT int pd_atom_pair_init(const float* pos, const long long* uid, const float* cl, const float* cm,
                                const float* Wp, const float* Wd, const float* Wv, float* ap, int A, int c_ap, void* stream) {
    if (!pos || !uid || !cl || !cm || !ap || A <= 0) return PD_ERR_ARG;
    dim3 grid((A + 255) / 256, A);
    hipStream_t s = (hipStream_t)stream;
    if (c_ap == 16) hipLaunchKernelGGL(atom_pair_init_kernel<16>, grid, dim3(256), 0, s, pos, uid, cl, cm, Wp, Wd, Wv, ap, A);
    else if (c_ap == 8) hipLaunchKernelGGL(atom_pair_init_kernel<8>, grid, dim3(256), 0, s, pos, uid, cl, cm, Wp, Wd, Wv, ap, A);
    else if (c_ap == 32) hipLaunchKernelGGL(atom_pair_init_kernel<32>, grid, dim3(256), 0, s, pos, uid, cl, cm, Wp, Wd, Wv, ap, A);
    else return PD_ERR_UNSUPPORTED;
    return pd_check_launch();
}

PD_EXPORT int pd_pair_gather_add(float* ap, const float* zt, const long long* a2t, int A, int T, int c_ap, void* stream) {
    if (!ap || !zt || !a2t) return PD_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((A * (c_ap / 4) + 255) / 256, A);
    if (c_ap == 16) hipLaunchKernelGGL(pair_gather_add_kernel<16>, grid, dim3(256), 0, s, ap, zt, a2t, A, T);
    else if (c_ap == 8) hipLaunchKernelGGL(pair_gather_add_kernel<8>, grid, dim3(256), 0, s, ap, zt, a2t, A, T);
    else if (c_ap == 32) hipLaunchKernelGGL(pair_gather_add_kernel<32>, grid, dim3(256), 0, s, ap, zt, a2t, A, T);
    else return PD_ERR_UNSUPPORTED;
    return pd_check_launch();
}

PD_EXPORT int pd_pair_init_z(const float* si, const float* sj, const float* WT, const float* wb, const int* asym,
                             const int* sym, const int* ent, const long long* res, const float* rel_tok_feat,
                             const float* bonds, float* z, int T, int CZ, void* stream) {
    if (!si || !sj || !WT || !z || T <= 0) return PD_ERR_ARG;
    const int threads = CZ < 256 ? ((CZ + 63) / 64) * 64 : 256;
    hipLaunchKernelGGL(pair_init_z_kernel, dim3(T, T >= 64 ? 16 : 1), dim3(threads), 42 * CZ * sizeof(float), (hipStream_t)stream, si, sj,
                       WT, wb, asym, sym, ent, res, rel_tok_feat, bonds, z, T, CZ);
    return pd_check_launch();
}

PD_EXPORT int pd_segment_pool(const float* u, const int* tok_start, const float* add, float* out, int B, int A, int T,
                              int C, void* stream) {
    if (!u || !tok_start || !out || C % 4) return PD_ERR_ARG;
    const long long n4 = (long long)B * T * (C / 4);
    hipLaunchKernelGGL(segment_pool_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u, tok_start,
                       add, out, A, T, C, n4);
    return pd_check_launch();
}

PD_EXPORT int pd_unpool_add(float* ba, const float* us, const long long* a2t, int B, int A, int T, int C, void* stream) {
    if (!ba || !us || !a2t || C % 4) return PD_ERR_ARG;
    const long long n4 = (long long)B * A * (C / 4);
    if (n4 + 256 < 0x7fffffffll)
        hipLaunchKernelGGL(unpool_add_kernel<unsigned>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ba, us, a2t,
                           A, T, C, n4);
    else
        hipLaunchKernelGGL(unpool_add_kernel<long long>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ba, us, a2t,
                           A, T, C, n4);
    return pd_check_launch();
}

PD_EXPORT int pd_gather_rows_add(float* y, const float* x, const long long* idx, int R, int C, void* stream) {
    if (!y || !x || !idx || C % 4) return PD_ERR_ARG;
    const long long n = (long long)R * (C / 4);
    hipLaunchKernelGGL(gather_rows_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, x,
                       idx, R, C);
    return pd_check_launch();
}

PD_EXPORT int pd_axpby(float* out, const float* a, float sa, const float* b, const float* sb_ptr, float sb, long long n,
                       void* stream) {
    if (!out || !a || n <= 0) return PD_ERR_ARG;
    hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)(((n + 3) / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, a, sa, b,
                       sb_ptr, sb, n);
    return pd_check_launch();
}

PD_EXPORT int pd_template_mask(const float* z_mask, const float* templ_feat, const int* asym, float* out, int T, int D,
                               void* stream) {
    if (!z_mask || !templ_feat || !asym || !out) return PD_ERR_ARG;
    hipLaunchKernelGGL(template_mask_kernel, dim3((unsigned)(((long long)T * T + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, z_mask, templ_feat, asym, out, T, D);
    return pd_check_launch();
}
