// Feature tensorisation and the PDB writer on the device: the steps either side of the sampler.
//   FeatureLoader.transform   reference PhysDock/data/feature_loader.py:970-998  (make_feats :803-851, _make_token_bonds :853-911,
//                             masks :982-983; the template block is pd_template_feat in pair.hip)
//   FeatureLoader.write_pdb_block                        feature_loader.py:1230-1283
// All of it is byte / index work bounded by HBM: every kernel reads its inputs once and writes its output once, coalesced.
#include "common.h"
#include "physdock_hip.h"

namespace {

// target_feat[t] = [one_hot(restype[t], NC) | profile[t, 0:NP] | deletion_mean[t]]                       (feature_loader.py:805-809)
__global__ __launch_bounds__(256) void target_feat_kernel(const long long* __restrict__ restype, const float* __restrict__ profile,
                                                         const float* __restrict__ deletion_mean, float* __restrict__ out, int T,
                                                         int NC, int NP) {
    const int W = NC + NP + 1;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)T * W) return;
    const int t = (int)(idx / W), c = (int)(idx - (long long)t * W);
    float v;
    if (c < NC) v = restype[t] == c ? 1.f : 0.f;
    else if (c < NC + NP) v = profile[(long long)t * NP + (c - NC)];
    else v = deletion_mean[t];
    out[idx] = v;
}

// msa_feat[r, t] = [one_hot(msa[inds[r], t], NC) | clamp(del, 0, 1) | atan(del / 3) * (2 / pi)],  del = deletion_matrix[inds[r], t]
// (feature_loader.py:813-826; the row gather `msa[inds]` of :815-816 is fused)
__global__ __launch_bounds__(256) void msa_feat_kernel(const long long* __restrict__ msa, const float* __restrict__ deletion,
                                                      const long long* __restrict__ inds, float two_over_pi,
                                                      float* __restrict__ out, int S2, int T, int NC) {
    const int W = NC + 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)S2 * T * W) return;
    const long long cell = idx / W;
    const int c = (int)(idx - cell * W);
    const int r = (int)(cell / T), t = (int)(cell - (long long)r * T);
    const long long src = inds[r] * T + t;
    float v;
    if (c < NC) v = msa[src] == c ? 1.f : 0.f;
    else {
        const float d = deletion[src];
        v = c == NC ? fminf(fmaxf(d, 0.f), 1.f) : __fmul_rn(atanf(__fdiv_rn(d, 3.f)), two_over_pi);
    }
    out[idx] = v;
}

// out[i, j] = m[i] * m[j]                                                                                (feature_loader.py:982-983)
__global__ __launch_bounds__(256) void outer_mask_kernel(const float* __restrict__ m, float* __restrict__ out, int N) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)N * N) return;
    const int i = (int)(idx / N), j = (int)(idx - (long long)i * N);
    out[idx] = m[i] * m[j];
}

// One workgroup per searched chain pair (ci, cj): the atom pair with the smallest  |x_a - x_b| + (1 - mask_a mask_b) 1000  (first
// in row-major (a, b) order on ties = torch.argmin of the flattened matrix); when that distance is below the threshold, the
// tokens of the two atoms are bonded, symmetrically, in `between` [T,T] (zeroed by the caller)        (feature_loader.py:882-900)
__global__ __launch_bounds__(256) void chain_contacts_kernel(const float* __restrict__ x, const float* __restrict__ a_mask,
                                                            const int* __restrict__ chain_start, const int* __restrict__ pairs,
                                                            const long long* __restrict__ a2t, float threshold,
                                                            float* __restrict__ between, int T, float* __restrict__ min_out,
                                                            long long* __restrict__ arg_out) {
    __shared__ float sv[256];
    __shared__ long long si[256];
    const int p = blockIdx.x;
    const int ci = pairs[2 * p], cj = pairs[2 * p + 1];
    const int i0 = chain_start[ci], ni = chain_start[ci + 1] - i0;
    const int j0 = chain_start[cj], nj = chain_start[cj + 1] - j0;
    const long long n = (long long)ni * nj;
    float best = INFINITY;
    long long arg = 0x7fffffffffffffffLL;
    for (long long k = threadIdx.x; k < n; k += 256) {
        const int a = (int)(k / nj), b = (int)(k - (long long)a * nj);
        const float* xa = x + 3 * (long long)(i0 + a);
        const float* xb = x + 3 * (long long)(j0 + b);
        const float dx = xa[0] - xb[0], dy = xa[1] - xb[1], dz = xa[2] - xb[2];
        const float d = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
        const float v = __fadd_rn(d, __fmul_rn(1.f - __fmul_rn(a_mask[i0 + a], a_mask[j0 + b]), 1000.f));
        if (v < best) { best = v; arg = k; }            // k ascending per thread: the first minimum is kept
    }
    sv[threadIdx.x] = best;
    si[threadIdx.x] = arg;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const float ov = sv[threadIdx.x + s];
            const long long oi = si[threadIdx.x + s];
            if (ov < sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (min_out) min_out[p] = sv[0];
        if (arg_out) arg_out[p] = n > 0 ? si[0] : -1;
        if (n > 0 && sv[0] < threshold) {
            const int a = (int)(si[0] / nj), b = (int)(si[0] - (long long)a * nj);
            const long long ti = a2t[i0 + a], tj = a2t[j0 + b];
            between[ti * T + tj] = 1.f;
            between[tj * T + ti] = 1.f;
        }
    }
}

// One thread per output byte of the ATOM / HETATM records: the template row (everything but the coordinates, built once per
// system on the host) is copied, columns 31-54 are the three coordinates as Python's f"{v:>8.3f}" of the fp32 value
// (feature_loader.py:1265-1267): v * 1000 is exact in fp64 (24 + 10 significant bits), rint = the round-half-even of the
// correctly rounded decimal conversion, a negative sign survives rounding to zero ("-0.000").  Values that do not fit the
// 8-character field (or are not finite) are counted in *overflow and printed as '*'.
__global__ __launch_bounds__(256) void pdb_format_kernel(const float* __restrict__ x, const unsigned char* __restrict__ tmpl,
                                                        const int* __restrict__ atom, unsigned char* __restrict__ out,
                                                        int* __restrict__ overflow, int B, int A, int N) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long per = (long long)N * 81;
    if (idx >= (long long)B * per) return;
    const int b = (int)(idx / per);
    const long long r = idx - (long long)b * per;
    const int n = (int)(r / 81), col = (int)(r - (long long)n * 81);
    unsigned char ch = tmpl[r];
    if (col >= 30 && col < 54) {
        const int f = (col - 30) >> 3, p = (col - 30) & 7;
        const float v = x[((long long)b * A + atom[n]) * 3 + f];
        const bool neg = __builtin_signbit(v);
        const double sc = fabs((double)v) * 1000.0;
        bool bad = !(sc < 1e13);                                     // inf / nan / absurd
        const long long m = bad ? 0 : (long long)rint(sc);
        const long long I = m / 1000;
        const int F = (int)(m - I * 1000);
        if (I >= 10000 || (neg && I >= 1000)) bad = true;
        if (bad) {
            ch = '*';
            if (p == 0) atomicAdd(overflow, 1);
        } else if (p == 7) ch = '0' + F % 10;
        else if (p == 6) ch = '0' + (F / 10) % 10;
        else if (p == 5) ch = '0' + F / 100;
        else if (p == 4) ch = '.';
        else {
            const int q = 3 - p;                                     // this column holds the 10^q digit, q = 0..3
            const int nd = I >= 1000 ? 4 : I >= 100 ? 3 : I >= 10 ? 2 : 1;
            long long pw = 1;
            for (int e = 0; e < q; ++e) pw *= 10;
            ch = q < nd ? '0' + (int)((I / pw) % 10) : (neg && q == nd) ? '-' : ' ';
        }
    }
    out[idx] = ch;
}

inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

PD_EXPORT int pd_target_feat(const long long* restype, const float* profile, const float* deletion_mean, float* out, int T,
                             int n_class, int n_profile, void* stream) {
    if (!restype || !profile || !deletion_mean || !out || T <= 0 || n_class <= 0 || n_profile < 0) return PD_ERR_ARG;
    hipLaunchKernelGGL(target_feat_kernel, dim3(blocks_for((long long)T * (n_class + n_profile + 1))), dim3(256), 0,
                       (hipStream_t)stream, restype, profile, deletion_mean, out, T, n_class, n_profile);
    return pd_check_launch();
}

PD_EXPORT int pd_msa_feat(const long long* msa, const float* deletion_matrix, const long long* inds, float two_over_pi, float* out,
                          int n_rows_out, int T, int n_class, void* stream) {
    if (!msa || !deletion_matrix || !inds || !out || n_rows_out <= 0 || T <= 0 || n_class <= 0) return PD_ERR_ARG;
    hipLaunchKernelGGL(msa_feat_kernel, dim3(blocks_for((long long)n_rows_out * T * (n_class + 2))), dim3(256), 0,
                       (hipStream_t)stream, msa, deletion_matrix, inds, two_over_pi, out, n_rows_out, T, n_class);
    return pd_check_launch();
}

PD_EXPORT int pd_outer_mask(const float* m, float* out, int N, void* stream) {
    if (!m || !out || N <= 0) return PD_ERR_ARG;
    hipLaunchKernelGGL(outer_mask_kernel, dim3(blocks_for((long long)N * N)), dim3(256), 0, (hipStream_t)stream, m, out, N);
    return pd_check_launch();
}

PD_EXPORT int pd_chain_contacts(const float* x, const float* a_mask, const int* chain_start, const int* pairs, int n_pairs,
                                const long long* a2t, float threshold, float* between, int T, float* min_out, long long* arg_out,
                                void* stream) {
    if (!x || !a_mask || !chain_start || !pairs || !a2t || !between || n_pairs < 0 || T <= 0) return PD_ERR_ARG;
    if (n_pairs == 0) return PD_OK;
    hipLaunchKernelGGL(chain_contacts_kernel, dim3(n_pairs), dim3(256), 0, (hipStream_t)stream, x, a_mask, chain_start, pairs, a2t,
                       threshold, between, T, min_out, arg_out);
    return pd_check_launch();
}

PD_EXPORT int pd_pdb_format(const float* x, const unsigned char* tmpl, const int* atom, unsigned char* out, int* overflow, int B,
                            int A, int N, void* stream) {
    if (!x || !tmpl || !atom || !out || !overflow || B <= 0 || A <= 0 || N < 0) return PD_ERR_ARG;
    if (N == 0) return PD_OK;
    hipLaunchKernelGGL(pdb_format_kernel, dim3(blocks_for((long long)B * N * 81)), dim3(256), 0, (hipStream_t)stream, x, tmpl, atom,
                       out, overflow, B, A, N);
    return pd_check_launch();
}
