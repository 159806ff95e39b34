// MMFF94 ligand relaxation on the device: energy, analytic gradient and the BFGS iterations the reference runs through
// RDKit on the host (reference models/model.py:26-52: for every sample `AllChem.MMFFOptimizeMolecule(ref_mol,
// mmffVariant="MMFF94", maxIters=mmff_iters, ignoreInterfragInteractions=True)`, called once per reverse-diffusion step
// while t_cur <= gamma_min * mmff_gamma_0_factor, model.py:252-255).
//
// RDKit (pip rdkit==2024.3.3) is a third-party dependency that is neither vendored by the reference nor installed in
// this image, so the arithmetic follows the PUBLISHED algorithm (parity with RDKit itself: unpinned):
//   * MMFF94 functional forms of Halgren, J. Comput. Chem. 17 (1996) 490, with the constants of RDKit's MMFF/Params.h;
//   * the BFGS of Numerical Recipes (dfpmin / lnsrch) as RDKit's Numerics/Optimizer/BFGSOpt.h codes it, driven by
//     ForceField::minimize(maxIts, forceTol = 1e-4) with the gradient scaling of ForceFieldsHelper::calcGradient.
// The parameter tables (pd_mmff_terms) come from the host: physdock_amd/mmff.py reads them from RDKit's own per-term
// getters when RDKit is present and verifies this kernel's energy / gradient against RDKit's force field.
//
// Mapping: one workgroup (256 threads) per diffusion sample, everything in fp64 (as RDKit).  One thread owns one ligand
// atom and computes ITS gradient: it walks the atom's incident bonded terms (CSR list built on the host) and the dense
// non-bonded row of the atom - no atomics, so the summation order and therefore the result is run-to-run reproducible.
// Positions live in LDS; the BFGS vectors and the dense inverse Hessian (dim^2 doubles, dim = 3 L <= a few hundred)
// live in a global workspace that stays L2-resident.  The whole relaxation is one launch and part of the step-loop graph.
#include "common.h"
#include "physdock_hip.h"

namespace {

constexpr double MDYNE_A = 143.9325;
constexpr double PI_D = 3.14159265358979323846;
constexpr double DEG2RAD = PI_D / 180.0;
constexpr double RAD2DEG = 180.0 / PI_D;
constexpr double C2 = MDYNE_A * DEG2RAD * DEG2RAD;
constexpr double C5 = MDYNE_A * DEG2RAD;
constexpr double CS = -2.0;
constexpr double CS3 = 7.0 / 12.0;
constexpr double CB = -0.006981317;
constexpr double ELE_K = 332.0716;
constexpr double ELE_BUF = 0.05;
constexpr int NT = 256;

struct V3 { double x, y, z; };
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ double norm(V3 a) { return sqrt(dot(a, a)); }
__device__ __forceinline__ V3 ld(const double* p, int a) { return {p[3 * a], p[3 * a + 1], p[3 * a + 2]}; }
__device__ __forceinline__ double clip1(double c) { return c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c); }

// deterministic block reductions (fixed tree) of doubles; the result is returned to every thread
__device__ __forceinline__ double block_sum_d(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ double block_max_d(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// cos(theta) at j and its gradient with respect to the three points
struct Ang { double c, la, lb; V3 gi, gj, gk; };
__device__ __forceinline__ Ang angle_terms(V3 pi, V3 pj, V3 pk) {
    Ang r;
    const V3 a = pi - pj, b = pk - pj;
    r.la = norm(a); r.lb = norm(b);
    r.c = clip1(dot(a, b) / (r.la * r.lb));
    r.gi = (1.0 / r.la) * ((1.0 / r.lb) * b - (r.c / r.la) * a);
    r.gk = (1.0 / r.lb) * ((1.0 / r.la) * a - (r.c / r.lb) * b);
    r.gj = -1.0 * (r.gi + r.gk);
    return r;
}
__device__ __forceinline__ V3 pick3(int slot, V3 a, V3 b, V3 c) { return slot == 0 ? a : (slot == 1 ? b : c); }
__device__ __forceinline__ V3 pick4(int slot, V3 a, V3 b, V3 c, V3 d) { return slot == 0 ? a : (slot == 1 ? b : (slot == 2 ? c : d)); }

// energy of one bonded term (returned through e) and its gradient with respect to the atom in `slot`
__device__ V3 bonded_term(const pd_mmff_terms& T, int kind, int t, int slot, const double* p, double& e) {
    if (kind == 0) {                                   // bond stretching
        const int i = T.bond_idx[2 * t], j = T.bond_idx[2 * t + 1];
        const double kb = T.bond_par[2 * t], r0 = T.bond_par[2 * t + 1];
        const V3 d = ld(p, i) - ld(p, j);
        const double r = norm(d), x = r - r0;
        e = 0.5 * MDYNE_A * kb * x * x * (1.0 + CS * x + CS3 * CS * CS * x * x);
        const double dE = MDYNE_A * kb * x * (1.0 + 1.5 * CS * x + 2.0 * CS3 * CS * CS * x * x);
        return ((slot == 0 ? dE : -dE) / r) * d;
    }
    if (kind == 1) {                                   // angle bending
        const int i = T.angle_idx[3 * t], j = T.angle_idx[3 * t + 1], k = T.angle_idx[3 * t + 2];
        const double ka = T.angle_par[3 * t], th0 = T.angle_par[3 * t + 1], lin = T.angle_par[3 * t + 2];
        const Ang A = angle_terms(ld(p, i), ld(p, j), ld(p, k));
        double dEdc;
        if (lin > 0.5) {
            e = MDYNE_A * ka * (1.0 + A.c);
            dEdc = MDYNE_A * ka;
        } else {
            const double x = RAD2DEG * acos(A.c) - th0;
            e = 0.5 * C2 * ka * x * x * (1.0 + CB * x);
            const double s = fmax(sqrt(fmax(1.0 - A.c * A.c, 0.0)), 1e-8);
            dEdc = C2 * ka * x * (1.0 + 1.5 * CB * x) * RAD2DEG * (-1.0 / s);
        }
        return dEdc * pick3(slot, A.gi, A.gj, A.gk);
    }
    if (kind == 2) {                                   // stretch-bend
        const int i = T.strbnd_idx[3 * t], j = T.strbnd_idx[3 * t + 1], k = T.strbnd_idx[3 * t + 2];
        const double* q = T.strbnd_par + 5 * t;        // kbaIJK, kbaKJI, r0_ij, r0_kj, theta0
        const V3 pi = ld(p, i), pj = ld(p, j), pk = ld(p, k);
        const Ang A = angle_terms(pi, pj, pk);
        const double dth = RAD2DEG * acos(A.c) - q[4];
        const double d1 = A.la - q[2], d2 = A.lb - q[3];
        e = C5 * dth * (q[0] * d1 + q[1] * d2);
        const double s = fmax(sqrt(fmax(1.0 - A.c * A.c, 0.0)), 1e-8);
        const double w = C5 * (q[0] * d1 + q[1] * d2) * RAD2DEG * (-1.0 / s);
        const V3 ua = (1.0 / A.la) * (pi - pj), ub = (1.0 / A.lb) * (pk - pj);
        const V3 gi = w * A.gi + (C5 * dth * q[0]) * ua;
        const V3 gk = w * A.gk + (C5 * dth * q[1]) * ub;
        const V3 gj = w * A.gj - (C5 * dth) * (q[0] * ua + q[1] * ub);
        return pick3(slot, gi, gj, gk);
    }
    if (kind == 3) {                                   // out-of-plane bending: Wilson angle of j-l against the plane i-j-k
        const int i = T.oop_idx[4 * t], j = T.oop_idx[4 * t + 1], k = T.oop_idx[4 * t + 2], l = T.oop_idx[4 * t + 3];
        const double koop = T.oop_par[t];
        const V3 pj = ld(p, j);
        const V3 a = ld(p, i) - pj, b = ld(p, k) - pj, c = ld(p, l) - pj;
        const V3 n = cross(a, b);
        const double N = norm(n), Cn = norm(c);
        const double s = clip1(dot(n, c) / (N * Cn));
        const double chi = RAD2DEG * asin(s);
        e = 0.5 * C2 * koop * chi * chi;
        const double dEds = C2 * koop * chi * RAD2DEG / fmax(sqrt(fmax(1.0 - s * s, 0.0)), 1e-8);
        const V3 ds_dc = (1.0 / (N * Cn)) * n - (s / (Cn * Cn)) * c;
        const V3 gn = (1.0 / (N * Cn)) * c - (s / (N * N)) * n;
        const V3 ds_da = cross(b, gn), ds_db = cross(gn, a);
        const V3 gj = -1.0 * (ds_da + ds_db + ds_dc);
        return dEds * pick4(slot, ds_da, gj, ds_db, ds_dc);
    }
    // torsion
    const int i = T.tors_idx[4 * t], j = T.tors_idx[4 * t + 1], k = T.tors_idx[4 * t + 2], l = T.tors_idx[4 * t + 3];
    const double v1 = T.tors_par[3 * t], v2 = T.tors_par[3 * t + 1], v3 = T.tors_par[3 * t + 2];
    const V3 pi = ld(p, i), pj = ld(p, j), pk = ld(p, k), pl = ld(p, l);
    const V3 r1 = pi - pj, r2 = pk - pj, r3 = pj - pk, r4 = pl - pk;
    const V3 t1 = cross(r1, r2), t2 = cross(r3, r4);
    const double d1 = norm(t1), d2 = norm(t2);
    e = 0.0;
    if (d1 < 1e-12 || d2 < 1e-12) return {0.0, 0.0, 0.0};
    const double c = clip1(dot(t1, t2) / (d1 * d2));
    e = 0.5 * (v1 * (1.0 + c) + v2 * (1.0 - (2.0 * c * c - 1.0)) + v3 * (1.0 + (4.0 * c * c * c - 3.0 * c)));
    const double dEdc = 0.5 * (v1 - 4.0 * v2 * c + 3.0 * v3 * (4.0 * c * c - 1.0));
    const V3 g1 = (1.0 / d1) * ((1.0 / d2) * t2 - (c / d1) * t1);
    const V3 g2 = (1.0 / d2) * ((1.0 / d1) * t1 - (c / d2) * t2);
    const V3 dr1 = cross(r2, g1), dr2 = cross(g1, r1), dr3 = cross(r4, g2), dr4 = cross(g2, r3);
    const V3 gj = dr3 - dr1 - dr2;
    const V3 gk = dr2 - dr3 - dr4;
    return dEdc * pick4(slot, dr1, gj, gk, dr4);
}

// threads that share one atom's terms: the largest power of two <= NT / L, at most 8 (lanes of one wave, adjacent)
__device__ __forceinline__ int threads_per_atom(int L) {
    int t = 8;
    while (t > 1 && t * L > NT) t >>= 1;
    return t;
}

// MMFF94 energy of the conformation p (LDS, [L][3]) - returned to every thread - and, if g != nullptr, its gradient
// g[3 L] (any address space reachable by generic pointers).  Must be called by all NT threads.
// An atom's incident bonded terms and its non-bonded row are dealt round-robin to `tpa` adjacent lanes, whose partial
// gradients / energies are then added in a fixed xor-tree: no atomics, bit-reproducible, and a 32-atom ligand keeps all 256
// threads busy instead of 32 (one evaluation 100 -> ~20 us; the relaxation is a chain of ~15 of them per sampler step).
__device__ double energy_grad(const pd_mmff_terms& T, const double* p, double* g, double* red) {
    const int L = T.n_atoms;
    const int tpa = threads_per_atom(L), sub = threadIdx.x & (tpa - 1);
    double e_own = 0.0;
    const int a_step = NT / tpa;
    const int a_end = (L + a_step - 1) / a_step * a_step;          // whole passes: every lane takes part in the shuffles below
    for (int a = threadIdx.x / tpa; a < a_end; a += a_step) {
        V3 ga = {0.0, 0.0, 0.0};
        double e_nb = 0.0, e_b = 0.0;
        if (a < L) {
            for (int q = T.inc_ptr[a] + sub; q < T.inc_ptr[a + 1]; q += tpa) {
                const int code = T.inc[q];
                const int kind = (code >> 28) & 7, slot = (code >> 24) & 15, t = code & 0xFFFFFF;
                double e;
                ga = ga + bonded_term(T, kind, t, slot, p, e);
                if (slot == 0) e_b += e;
            }
            const V3 pa = ld(p, a);
            const double* Rr = T.vdw_R + (long long)a * L;
            const double* Er = T.vdw_eps + (long long)a * L;
            const double* Qr = T.ele_qq + (long long)a * L;
            for (int j = sub; j < L; j += tpa) {
                const double eps = Er[j], qq = Qr[j];
                if (j == a || (eps == 0.0 && qq == 0.0)) continue;
                const V3 d = pa - ld(p, j);
                const double r = norm(d);
                double dE = 0.0;
                if (eps != 0.0) {                            // buffered 14-7
                    const double Rs = Rr[j];
                    const double R2 = Rs * Rs, R7 = R2 * R2 * R2 * Rs;
                    const double r2 = r * r, r6 = r2 * r2 * r2, r7 = r6 * r;
                    const double den = r + 0.07 * Rs;
                    const double a1 = 1.07 * Rs / den;
                    const double a2 = a1 * a1, a7 = a2 * a2 * a2 * a1;
                    const double bden = r7 + 0.12 * R7;
                    const double bt = 1.12 * R7 / bden - 2.0;
                    e_nb += eps * a7 * bt;
                    dE += eps * ((-7.0 * a7 / den) * bt + a7 * (-1.12 * R7 * 7.0 * r6 / (bden * bden)));
                }
                if (qq != 0.0) {                             // buffered Coulomb, constant dielectric
                    const double rb = r + ELE_BUF;
                    e_nb += ELE_K * qq / rb;
                    dE += -ELE_K * qq / (rb * rb);
                }
                ga = ga + (dE / r) * d;
            }
        }
        double e_a = e_b + 0.5 * e_nb;                       // every pair is visited from both of its atoms
        for (int o = 1; o < tpa; o <<= 1) {                  // fixed tree over the atom's lanes
            ga.x += __shfl_xor(ga.x, o); ga.y += __shfl_xor(ga.y, o); ga.z += __shfl_xor(ga.z, o);
            e_a += __shfl_xor(e_a, o);
        }
        if (sub == 0 && a < L) {
            e_own += e_a;
            if (g) { g[3 * a] = ga.x; g[3 * a + 1] = ga.y; g[3 * a + 2] = ga.z; }
        }
    }
    return block_sum_d(e_own, red);
}

// ForceFieldsHelper::calcGradient: gradient x 0.1, then halved while its (signed) maximum stays above 10.  Returns the
// scale the optimiser's convergence test uses.  p: LDS positions; g: output (global).
__device__ double scaled_gradient(const pd_mmff_terms& T, const double* p, double* g, int dim, double* red) {
    energy_grad(T, p, g, red);
    __syncthreads();
    double mx = -1e8;
    for (int i = threadIdx.x; i < dim; i += NT) { g[i] *= 0.1; mx = fmax(mx, g[i]); }
    mx = block_max_d(mx, red);
    double scale = 0.1;
    if (mx > 10.0) {
        for (int guard = 0; guard < 1100 && mx * scale > 10.0; ++guard) scale *= 0.5;     // (bounded: mx may be inf)
        for (int i = threadIdx.x; i < dim; i += NT) g[i] *= scale;
    }
    __syncthreads();
    return scale;
}

__global__ __launch_bounds__(NT) void mmff_energy_grad_kernel(const pd_mmff_terms T, const double* __restrict__ pos,
                                                             double* __restrict__ energy, double* __restrict__ grad) {
    extern __shared__ double sm[];
    double* sp = sm;                 // [3 L]
    double* red = sm + 3 * T.n_atoms;
    const int b = blockIdx.x, dim = 3 * T.n_atoms;
    for (int i = threadIdx.x; i < dim; i += NT) sp[i] = pos[(long long)b * dim + i];
    __syncthreads();
    const double e = energy_grad(T, sp, grad ? grad + (long long)b * dim : nullptr, red);
    if (threadIdx.x == 0 && energy) energy[b] = e;
}

constexpr double FUNCTOL = 1e-4, MOVETOL = 1e-7, EPS_ = 3e-8, TOLX = 4.0 * EPS_, MAXSTEP = 100.0, FORCE_TOL = 1e-4;

// x_ref = x with ligand rows relaxed: BFGSOpt.h::minimize on the MMFF94 energy, maxIts = max_iters
// out[i] = sign * sum_j H[j][i] v[j] for the symmetric dim x dim matrix H (column reads are coalesced).  With dim <= NT / 2 the
// j range is cut into NT / dim parts summed in a fixed order (partials through `part`, NT doubles of LDS): 96 outputs of a
// 32-atom ligand use 192 threads instead of 96, each with half the dependent chain.  Must be called by all NT threads.
__device__ void matvec_sym(const double* __restrict__ H, const double* __restrict__ v, double* __restrict__ out, double sign,
                           int dim, double* part) {
    const int tid = threadIdx.x;
    const int parts = dim <= NT ? NT / dim : 1;
    if (parts == 1) {
        for (int i = tid; i < dim; i += NT) {
            double h = 0.0;
#pragma unroll 4
            for (int j = 0; j < dim; ++j) h += H[(long long)j * dim + i] * v[j];
            out[i] = sign * h;
        }
        __syncthreads();
        return;
    }
    const int jc = (dim + parts - 1) / parts;
    double acc = 0.0;
    if (tid < parts * dim) {
        const int i = tid % dim, pt = tid / dim;
        const int j0 = pt * jc, j1 = (j0 + jc < dim) ? j0 + jc : dim;
#pragma unroll 4
        for (int j = j0; j < j1; ++j) acc += H[(long long)j * dim + i] * v[j];
    }
    part[tid] = acc;
    __syncthreads();
    if (tid < dim) {
        double h = 0.0;
        for (int pt = 0; pt < parts; ++pt) h += part[pt * dim + tid];
        out[tid] = sign * h;
    }
    __syncthreads();
}

__global__ __launch_bounds__(NT) void mmff_relax_kernel(const pd_mmff_terms T, const float* __restrict__ x,
                                                       const int* __restrict__ lig_idx, float* __restrict__ x_ref,
                                                       double* __restrict__ ws, int A, int max_iters) {
    extern __shared__ double sm[];
    const int L = T.n_atoms, dim = 3 * L, b = blockIdx.x, tid = threadIdx.x;
    double* sp = sm;                 // positions under evaluation [dim]
    double* red = sm + dim;          // [4]
    double* part = red + 4;          // [NT] partial sums of matvec_sym
    double* base = ws + (long long)b * ((long long)dim * dim + 8ll * dim);
    double* H = base;                                  // inverse Hessian [dim][dim], kept exactly symmetric
    double* pos = H + (long long)dim * dim;
    double* grad = pos + dim;
    double* dgrad = grad + dim;
    double* hdg = dgrad + dim;
    double* npos = hdg + dim;
    double* xi = npos + dim;

    const float* xb = x + (long long)b * A * 3;
    float* ob = x_ref + (long long)b * A * 3;
    for (int i = tid; i < A * 3; i += NT) ob[i] = xb[i];                  // `x_ref = deepcopy(x_denoised)` (model.py:253)
    for (int i = tid; i < dim; i += NT) {
        const double v = (double)xb[3 * lig_idx[i / 3] + i % 3];
        pos[i] = v; sp[i] = v;
    }
    __syncthreads();
    double fp = energy_grad(T, sp, nullptr, red);
    double gscale = scaled_gradient(T, sp, grad, dim, red);
    double ssum = 0.0;
    for (int i = tid; i < dim; i += NT) { xi[i] = -grad[i]; ssum += pos[i] * pos[i]; }
    for (long long e = tid; e < (long long)dim * dim; e += NT) H[e] = (e / dim == e % dim) ? 1.0 : 0.0;
    ssum = block_sum_d(ssum, red);
    const double max_step = MAXSTEP * fmax(sqrt(ssum), (double)dim);
    __syncthreads();

    for (int iter = 1; iter <= max_iters; ++iter) {
        // ---------------- linearSearch (Numerical Recipes lnsrch)
        double s = 0.0;
        for (int i = tid; i < dim; i += NT) s += xi[i] * xi[i];
        s = sqrt(block_sum_d(s, red));
        if (s > max_step) {
            for (int i = tid; i < dim; i += NT) xi[i] *= max_step / s;
        }
        double slope = 0.0, test = 0.0;
        for (int i = tid; i < dim; i += NT) {
            slope += xi[i] * grad[i];
            test = fmax(test, fabs(xi[i]) / fmax(fabs(pos[i]), 1.0));
        }
        slope = block_sum_d(slope, red);
        test = block_max_d(test, red);
        if (!(slope < 0.0)) break;                      // RDKit: "bad direction in linearSearch"; positions stay as they are
        const double lam_min = MOVETOL / test;
        double lam = 1.0, lam2 = 0.0, val2 = 0.0, fnew = fp;
        bool moved = false;
        for (int it = 0; it < 1000; ++it) {
            if (lam < lam_min) break;
            for (int i = tid; i < dim; i += NT) { const double v = pos[i] + lam * xi[i]; npos[i] = v; sp[i] = v; }
            __syncthreads();
            fnew = energy_grad(T, sp, nullptr, red);
            if (fnew - fp <= FUNCTOL * lam * slope) { moved = true; break; }
            double tmp;
            if (it == 0) tmp = -slope / (2.0 * (fnew - fp - slope));
            else {
                const double rhs1 = fnew - fp - lam * slope, rhs2 = val2 - fp - lam2 * slope;
                const double a = (rhs1 / (lam * lam) - rhs2 / (lam2 * lam2)) / (lam - lam2);
                const double bb = (-lam2 * rhs1 / (lam * lam) + lam * rhs2 / (lam2 * lam2)) / (lam - lam2);
                if (a == 0.0) tmp = -slope / (2.0 * bb);
                else {
                    const double disc = bb * bb - 3.0 * a * slope;
                    if (disc < 0.0) tmp = 0.5 * lam;
                    else if (bb <= 0.0) tmp = (-bb + sqrt(disc)) / (3.0 * a);
                    else tmp = -slope / (bb + sqrt(disc));
                }
                if (tmp > 0.5 * lam) tmp = 0.5 * lam;
            }
            lam2 = lam; val2 = fnew;
            lam = fmax(tmp, 0.1 * lam);
            __syncthreads();
        }
        if (!moved) {                                   // "nothing was done": the new point is the old one
            __syncthreads();
            for (int i = tid; i < dim; i += NT) npos[i] = pos[i];
        }
        __syncthreads();
        fp = fnew;
        // ---------------- step bookkeeping and convergence tests
        test = 0.0;
        for (int i = tid; i < dim; i += NT) {
            const double d = npos[i] - pos[i];
            xi[i] = d; pos[i] = npos[i]; sp[i] = npos[i];
            test = fmax(test, fabs(d) / fmax(fabs(npos[i]), 1.0));
            dgrad[i] = grad[i];
        }
        test = block_max_d(test, red);
        if (test < TOLX) break;
        gscale = scaled_gradient(T, sp, grad, dim, red);
        test = 0.0;
        for (int i = tid; i < dim; i += NT) {
            test = fmax(test, fabs(grad[i]) * fmax(fabs(pos[i]), 1.0));
            dgrad[i] = grad[i] - dgrad[i];
        }
        test = block_max_d(test, red) / fmax(fnew * gscale, 1.0);
        if (test < FORCE_TOL) break;
        // ---------------- BFGS update of the inverse Hessian (H is symmetric: column reads are coalesced)
        double fac = 0.0, fae = 0.0, sdg = 0.0, sxi = 0.0;
        __syncthreads();                                // dgrad is complete
        matvec_sym(H, dgrad, hdg, 1.0, dim, part);
        for (int i = tid; i < dim; i += NT) {
            fac += dgrad[i] * xi[i]; fae += dgrad[i] * hdg[i]; sdg += dgrad[i] * dgrad[i]; sxi += xi[i] * xi[i];
        }
        fac = block_sum_d(fac, red); fae = block_sum_d(fae, red); sdg = block_sum_d(sdg, red); sxi = block_sum_d(sxi, red);
        if (fac > sqrt(EPS_ * sdg * sxi)) {
            fac = 1.0 / fac;
            const double fad = 1.0 / fae;
            for (int i = tid; i < dim; i += NT) dgrad[i] = fac * xi[i] - fad * hdg[i];
            __syncthreads();
            const unsigned udim = (unsigned)dim, n2 = udim * udim;       // dim <= 3072: 32-bit index arithmetic
            for (unsigned e = tid; e < n2; e += NT) {
                const unsigned r = e / udim, c = e - r * udim;
                const unsigned i = r < c ? r : c, j = r < c ? c : r;      // the (i <= j) element RDKit computes and mirrors
                H[e] += (fac * xi[i]) * xi[j] - (fad * hdg[i]) * hdg[j] + (fae * dgrad[i]) * dgrad[j];
            }
        }
        __syncthreads();
        matvec_sym(H, grad, xi, -1.0, dim, part);
    }
    __syncthreads();
    for (int i = tid; i < dim; i += NT) ob[3 * lig_idx[i / 3] + i % 3] = (float)pos[i];
}

bool terms_ok(const pd_mmff_terms* t) {
    if (!t || t->n_atoms <= 0 || t->n_atoms > 1024) return false;
    if (!t->vdw_R || !t->vdw_eps || !t->ele_qq || !t->inc_ptr || !t->inc) return false;
    if ((t->n_bond && (!t->bond_idx || !t->bond_par)) || (t->n_angle && (!t->angle_idx || !t->angle_par)) ||
        (t->n_strbnd && (!t->strbnd_idx || !t->strbnd_par)) || (t->n_oop && (!t->oop_idx || !t->oop_par)) ||
        (t->n_tors && (!t->tors_idx || !t->tors_par)))
        return false;
    return true;
}

}  // namespace

PD_EXPORT int pd_mmff_energy_grad(const pd_mmff_terms* terms, const double* pos, double* energy, double* grad, int B,
                                  void* stream) {
    if (!terms_ok(terms) || !pos || (!energy && !grad) || B <= 0) return PD_ERR_ARG;
    const size_t lds = (3 * (size_t)terms->n_atoms + 4) * sizeof(double);
    hipLaunchKernelGGL(mmff_energy_grad_kernel, dim3(B), dim3(NT), lds, (hipStream_t)stream, *terms, pos, energy, grad);
    return pd_check_launch();
}

PD_EXPORT int pd_mmff_relax(const pd_mmff_terms* terms, const float* x, const int* lig_idx, float* x_ref, double* ws,
                            long long ws_doubles, int B, int A, int max_iters, void* stream) {
    if (!terms_ok(terms) || !x || !lig_idx || !x_ref || !ws || B <= 0 || A < terms->n_atoms || max_iters < 0) return PD_ERR_ARG;
    const long long dim = 3ll * terms->n_atoms;
    if (ws_doubles < (long long)B * (dim * dim + 8 * dim)) return PD_ERR_ARG;
    const size_t lds = ((size_t)dim + 4 + NT) * sizeof(double);
    hipLaunchKernelGGL(mmff_relax_kernel, dim3(B), dim3(NT), lds, (hipStream_t)stream, *terms, x, lig_idx, x_ref, ws, A, max_iters);
    return pd_check_launch();
}
