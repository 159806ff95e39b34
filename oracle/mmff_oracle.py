"""CPU restatement (numpy, float64) of the ligand relaxation the reference delegates to RDKit
(reference PhysDock/models/model.py:26-52: `AllChem.MMFFOptimizeMolecule(ref_mol, mmffVariant="MMFF94",
maxIters=mmff_iters, ignoreInterfragInteractions=True)`).

TEST INFRASTRUCTURE ONLY - imported by tests/, never by physdock_amd/.

**Parity unpinned.**  The arithmetic lives in the third-party dependency `rdkit==2024.3.3` (enviroment.yaml:33; C++
`Code/ForceField/MMFF/*` and `Code/Numerics/Optimizer/BFGSOpt.h`), which is not vendored in the reference and not
installed in this container; the reference holds no test or golden vector for this sub-step.  What is restated here
is the *published* algorithm:

* energy terms: T. A. Halgren, J. Comput. Chem. 17 (1996) 490-519 (MMFF94 functional forms, the constants RDKit's
  `MMFF/Params.h` uses: 143.9325 kcal/(mol A^2 mdyn), cubic-stretch -2 /A, cubic-bend -0.006981317 /deg (= -0.4 /rad),
  buffered 14-7 with 1.07 / 0.07 / 1.12 / 0.12, 332.0716 kcal A /(mol e^2) with 0.05 A buffer, 0.75 for 1-4 pairs);
* optimiser: the BFGS of Numerical Recipes (`dfpmin` + `lnsrch`) as RDKit's `BFGSOpt.h` codes it (FUNCTOL 1e-4,
  MOVETOL 1e-7, EPS 3e-8, TOLX 4 EPS, MAXSTEP 100, gradient scaled by 0.1 and halved while its maximum exceeds 10 -
  `ForceField.cpp` calcGradient), `forceTol` 1e-4, `maxIts = mmff_iters`.

The parameter assignment (atom typing, MMFFBOND.PAR ... tables) is NOT restated: term tables are inputs
(`physdock_amd.mmff.MMFFTerms`: built from RDKit's own per-term getters when RDKit is present, synthetic in the tests).
Pinned here instead: analytic gradients against central differences, hand-computed known answers for every term
kind, monotone energy decrease, and (on the GPU) the HIP kernel `pd_mmff_relax` against this restatement.
"""
from __future__ import annotations

import numpy as np

MDYNE_A = 143.9325
DEG2RAD = np.pi / 180.0
RAD2DEG = 180.0 / np.pi
C2 = MDYNE_A * DEG2RAD * DEG2RAD          # 0.0438449...
C5 = MDYNE_A * DEG2RAD                     # 2.51210...
CS = -2.0
CB = -0.006981317
ELE_K = 332.0716
ELE_BUF = 0.05


def _unit(v):
    n = np.linalg.norm(v)
    return v / n, n


def _angle_terms(pi, pj, pk):
    """cos(theta) at j and its gradient with respect to the three points"""
    a, b = pi - pj, pk - pj
    la, lb = np.linalg.norm(a), np.linalg.norm(b)
    c = float(np.clip(a @ b / (la * lb), -1.0, 1.0))
    dca = (b / lb - c * a / la) / la
    dcb = (a / la - c * b / lb) / lb
    return c, dca, -(dca + dcb), dcb, la, lb


def energy_and_grad(pos, terms, want_grad=True):
    """MMFF94 energy (kcal/mol) and gradient of one conformation.  pos [L,3] float64; terms: dict of numpy arrays
    (see physdock_amd.mmff.MMFFTerms.as_numpy)."""
    pos = np.asarray(pos, dtype=np.float64)
    g = np.zeros_like(pos)
    E = 0.0
    # ---- bond stretching
    for (i, j), (kb, r0) in zip(terms["bond_idx"], terms["bond_par"]):
        d = pos[i] - pos[j]
        r = np.linalg.norm(d)
        x = r - r0
        E += 0.5 * MDYNE_A * kb * x * x * (1.0 + CS * x + 7.0 / 12.0 * CS * CS * x * x)
        dE = MDYNE_A * kb * x * (1.0 + 1.5 * CS * x + 2.0 * (7.0 / 12.0) * CS * CS * x * x)
        g[i] += dE * d / r
        g[j] -= dE * d / r
    # ---- angle bending
    for (i, j, k), (ka, th0, lin) in zip(terms["angle_idx"], terms["angle_par"]):
        c, gi, gj, gk, _, _ = _angle_terms(pos[i], pos[j], pos[k])
        if lin > 0.5:
            E += MDYNE_A * ka * (1.0 + c)
            dEdc = MDYNE_A * ka
        else:
            th = RAD2DEG * np.arccos(c)
            x = th - th0
            E += 0.5 * C2 * ka * x * x * (1.0 + CB * x)
            dEdth = C2 * ka * x * (1.0 + 1.5 * CB * x)                # per degree
            s = max(np.sqrt(max(1.0 - c * c, 0.0)), 1e-8)
            dEdc = dEdth * RAD2DEG * (-1.0 / s)
        g[i] += dEdc * gi; g[j] += dEdc * gj; g[k] += dEdc * gk
    # ---- stretch-bend
    for (i, j, k), (kijk, kkji, r0ij, r0kj, th0) in zip(terms["strbnd_idx"], terms["strbnd_par"]):
        c, gi, gj, gk, la, lb = _angle_terms(pos[i], pos[j], pos[k])
        th = RAD2DEG * np.arccos(c)
        dth = th - th0
        d1, d2 = la - r0ij, lb - r0kj
        E += C5 * dth * (kijk * d1 + kkji * d2)
        s = max(np.sqrt(max(1.0 - c * c, 0.0)), 1e-8)
        dth_dc = RAD2DEG * (-1.0 / s)
        w = C5 * (kijk * d1 + kkji * d2) * dth_dc
        ua, ub = (pos[i] - pos[j]) / la, (pos[k] - pos[j]) / lb
        g[i] += w * gi + C5 * dth * kijk * ua
        g[k] += w * gk + C5 * dth * kkji * ub
        g[j] += w * gj - C5 * dth * (kijk * ua + kkji * ub)
    # ---- out-of-plane bending (Wilson angle of j-l against the plane i-j-k, j central)
    for (i, j, k, l), koop in zip(terms["oop_idx"], terms["oop_par"]):
        a, b, cc = pos[i] - pos[j], pos[k] - pos[j], pos[l] - pos[j]
        n = np.cross(a, b)
        N, Cn = np.linalg.norm(n), np.linalg.norm(cc)
        s = float(np.clip(n @ cc / (N * Cn), -1.0, 1.0))
        chi = RAD2DEG * np.arcsin(s)
        E += 0.5 * C2 * koop * chi * chi
        dEds = C2 * koop * chi * RAD2DEG / max(np.sqrt(max(1.0 - s * s, 0.0)), 1e-8)
        ds_dc = n / (N * Cn) - s * cc / (Cn * Cn)
        gn = cc / (N * Cn) - s * n / (N * N)
        ds_da, ds_db = np.cross(b, gn), np.cross(gn, a)
        g[i] += dEds * ds_da; g[k] += dEds * ds_db; g[l] += dEds * ds_dc
        g[j] -= dEds * (ds_da + ds_db + ds_dc)
    # ---- torsions
    for (i, j, k, l), (v1, v2, v3) in zip(terms["tors_idx"], terms["tors_par"]):
        r1, r2, r3, r4 = pos[i] - pos[j], pos[k] - pos[j], pos[j] - pos[k], pos[l] - pos[k]
        t1, t2 = np.cross(r1, r2), np.cross(r3, r4)
        d1, d2 = np.linalg.norm(t1), np.linalg.norm(t2)
        if d1 < 1e-12 or d2 < 1e-12:
            continue
        c = float(np.clip(t1 @ t2 / (d1 * d2), -1.0, 1.0))
        E += 0.5 * (v1 * (1.0 + c) + v2 * (1.0 - (2.0 * c * c - 1.0)) + v3 * (1.0 + (4.0 * c * c * c - 3.0 * c)))
        dEdc = 0.5 * (v1 - 4.0 * v2 * c + 3.0 * v3 * (4.0 * c * c - 1.0))
        g1 = (t2 / d2 - c * t1 / d1) / d1          # dc/dt1
        g2 = (t1 / d1 - c * t2 / d2) / d2          # dc/dt2
        # t1 = r1 x r2, t2 = r3 x r4
        dr1, dr2 = np.cross(r2, g1), np.cross(g1, r1)
        dr3, dr4 = np.cross(r4, g2), np.cross(g2, r3)
        g[i] += dEdc * dr1
        g[j] += dEdc * (-dr1 - dr2 + dr3)
        g[k] += dEdc * (dr2 - dr3 - dr4)
        g[l] += dEdc * dr4
    # ---- non-bonded: buffered 14-7 van der Waals and buffered Coulomb on the dense pair tables
    R, eps, qq = terms["vdw_R"], terms["vdw_eps"], terms["ele_qq"]
    L = pos.shape[0]
    for i in range(L):
        for j in range(i + 1, L):
            if eps[i, j] == 0.0 and qq[i, j] == 0.0:
                continue
            d = pos[i] - pos[j]
            r = np.linalg.norm(d)
            dE = 0.0
            if eps[i, j] != 0.0:
                Rs = R[i, j]
                R7 = Rs ** 7
                q = r / Rs
                a = 1.07 * Rs / (r + 0.07 * Rs)
                a7 = a ** 7
                r7 = r ** 7
                bt = 1.12 * R7 / (r7 + 0.12 * R7) - 2.0
                E += eps[i, j] * a7 * bt
                da7 = -7.0 * a7 / (r + 0.07 * Rs)
                dbt = -1.12 * R7 * 7.0 * r ** 6 / (r7 + 0.12 * R7) ** 2
                dE += eps[i, j] * (da7 * bt + a7 * dbt)
                del q
            if qq[i, j] != 0.0:
                E += ELE_K * qq[i, j] / (r + ELE_BUF)
                dE += -ELE_K * qq[i, j] / (r + ELE_BUF) ** 2
            g[i] += dE * d / r
            g[j] -= dE * d / r
    return (E, g) if want_grad else E


def scaled_gradient(pos, terms):
    """ForceFieldsHelper::calcGradient: gradient x 0.1, then halved while its (signed) maximum stays above 10; returns
    (scaled gradient, the scale the optimiser's convergence test uses)"""
    _, g = energy_and_grad(pos, terms)
    g = g.reshape(-1) * 0.1
    scale = 0.1
    mx = float(g.max())
    if mx > 10.0:
        while mx * scale > 10.0:
            scale *= 0.5
        g = g * scale
    return g, scale


FUNCTOL, MOVETOL, EPS, MAXSTEP = 1e-4, 1e-7, 3e-8, 100.0
TOLX = 4.0 * EPS


def _line_search(x_old, f_old, grad, direction, func, max_step):
    """Numerical Recipes lnsrch as coded in BFGSOpt.h::linearSearch -> (new point, new value, result code)"""
    d = direction
    s = np.sqrt(np.sum(d * d))
    if s > max_step:
        d *= max_step / s
    slope = float(np.sum(d * grad))
    if slope >= 0.0:
        return x_old.copy(), f_old, -1
    test = float(np.max(np.abs(d) / np.maximum(np.abs(x_old), 1.0)))
    lam_min = MOVETOL / test
    lam, lam2, val2, f_new = 1.0, 0.0, 0.0, f_old
    for it in range(1000):
        if lam < lam_min:
            return x_old.copy(), f_new, 1
        x_new = x_old + lam * d
        f_new = func(x_new)
        if f_new - f_old <= FUNCTOL * lam * slope:
            return x_new, f_new, 0
        if it == 0:
            tmp = -slope / (2.0 * (f_new - f_old - slope))
        else:
            rhs1 = f_new - f_old - lam * slope
            rhs2 = val2 - f_old - lam2 * slope
            a = (rhs1 / (lam * lam) - rhs2 / (lam2 * lam2)) / (lam - lam2)
            b = (-lam2 * rhs1 / (lam * lam) + lam * rhs2 / (lam2 * lam2)) / (lam - lam2)
            if a == 0.0:
                tmp = -slope / (2.0 * b)
            else:
                disc = b * b - 3.0 * a * slope
                if disc < 0.0:
                    tmp = 0.5 * lam
                elif b <= 0.0:
                    tmp = (-b + np.sqrt(disc)) / (3.0 * a)
                else:
                    tmp = -slope / (b + np.sqrt(disc))
            if tmp > 0.5 * lam:
                tmp = 0.5 * lam
        lam2, val2 = lam, f_new
        lam = max(tmp, 0.1 * lam)
    return x_old.copy(), f_new, 1


def minimize(pos0, terms, max_iters=5, force_tol=1e-4, return_energies=False):
    """BFGSOpt.h::minimize on the MMFF94 energy.  pos0 [L,3] -> relaxed [L,3] (float64)."""
    L = pos0.shape[0]
    dim = 3 * L
    x = np.asarray(pos0, dtype=np.float64).reshape(-1).copy()

    def func(p):
        return energy_and_grad(p.reshape(L, 3), terms, want_grad=False)

    fp = func(x)
    grad, _ = scaled_gradient(x.reshape(L, 3), terms)
    H = np.eye(dim)
    xi = -grad.copy()
    max_step = MAXSTEP * max(np.sqrt(np.sum(x * x)), float(dim))
    energies = [fp]
    for _ in range(max_iters):
        x_new, f_new, code = _line_search(x, fp, grad, xi, func, max_step)
        if code < 0:
            break                                  # RDKit raises ("bad direction in linearSearch"); positions unchanged
        fp = f_new
        xi = x_new - x
        x = x_new
        energies.append(func(x))
        if float(np.max(np.abs(xi) / np.maximum(np.abs(x), 1.0))) < TOLX:
            break
        dgrad = grad.copy()
        grad, gscale = scaled_gradient(x.reshape(L, 3), terms)
        term = max(f_new * gscale, 1.0)
        if float(np.max(np.abs(grad) * np.maximum(np.abs(x), 1.0))) / term < force_tol:
            break
        dgrad = grad - dgrad
        hdg = H @ dgrad
        fac, fae = float(dgrad @ xi), float(dgrad @ hdg)
        sum_dg, sum_xi = float(dgrad @ dgrad), float(xi @ xi)
        if fac > np.sqrt(EPS * sum_dg * sum_xi):
            fac = 1.0 / fac
            fad = 1.0 / fae
            u = fac * xi - fad * hdg
            H += fac * np.outer(xi, xi) - fad * np.outer(hdg, hdg) + fae * np.outer(u, u)
        xi = -(H @ grad)
    out = x.reshape(L, 3)
    return (out, energies) if return_energies else out
