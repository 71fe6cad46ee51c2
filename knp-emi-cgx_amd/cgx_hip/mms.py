"""Manufactured-solution verification layer of the native path (``MMS_test`` configs).

Mirrors reference src/CGx/utils/setup_mms.py (``ExactSolutionsKNPEMI``: exact solutions, symbolic source terms)
and the MMS branches of src/CGx/KNPEMI/KNPEMIx_problem.py (:109-134 Dirichlet data, :363-431 initial data,
:616-651 extra terms of L, :746-805 unit parameters, :845-907 error norms).

This is verification tooling, not the timed hot path: the analytic source integrals are evaluated on the host
with NumPy once per step and added to the right-hand side that the HIP kernels assembled; the Dirichlet rows
themselves are applied by the library (``knp_set_dirichlet``).  SymPy replaces UFL's symbolic differentiation.
"""
from __future__ import annotations

import math

import numpy as np
import sympy as sy


class ExactSolutionsKNPEMI:
    """Same public surface as the reference class: ``get_exact_solutions()``, ``get_mms_terms()``."""

    valence = {"Na": 1, "K": 1, "Cl": -1}

    def __init__(self, mesh, t):
        self.mesh = mesh
        self.dim = mesh.geometry.dim
        if self.dim not in (2, 3):
            raise ValueError("Mesh geometry dimension must be 2 or 3.")
        self.t = t                                         # Constant: current time
        self.X = sy.symbols("x y z", real=True)[: self.dim]
        self.ts = sy.Symbol("t", real=True)
        self.nsym = sy.symbols("n_x n_y n_z", real=True)[: self.dim]
        self._fn = {}

    # ---- symbolic part (setup_mms.py:29-74)
    def get_exact_solutions(self):
        bump = sy.exp(-self.ts)
        wave = sy.Integer(1)
        for c in self.X:
            bump = bump * sy.sin(2 * sy.pi * c)
            wave = wave * sy.cos(2 * sy.pi * c)
        return {"Na_i": 0.7 + 0.3 * bump, "K_i": 0.3 + 0.3 * bump, "Cl_i": 1.0 + 0.6 * bump,
                "phi_i": wave * (1 + sy.exp(-self.ts)),
                "Na_e": 1.0 + 0.6 * bump, "K_e": 1.0 + 0.2 * bump, "Cl_e": 2.0 + 0.8 * bump, "phi_e": wave,
                "phi_i_init": wave, "phi_e_init": wave}

    # ---- symbolic part (setup_mms.py:76-156)
    def get_mms_terms(self):
        ex = self.get_exact_solutions()
        X, t, n = self.X, self.ts, self.nsym
        d = self.dim

        def flux(k, phi, z):
            return [-sy.diff(k, X[a]) - z * k * sy.diff(phi, X[a]) for a in range(d)]

        def divergence(v):
            return sum(sy.diff(v[a], X[a]) for a in range(d))
        J = {f"{ion}_{r}": flux(ex[f"{ion}_{r}"], ex[f"phi_{r}"], z) for ion, z in self.valence.items() for r in "ie"}
        src = {}
        for ion in self.valence:
            for r in "ie":
                src[f"f_{ion}_{r}"] = sy.diff(ex[f"{ion}_{r}"], t) + divergence(J[f"{ion}_{r}"])
        for r in "ie":
            src[f"f_phi_{r}"] = -sum(z * divergence(J[f"{ion}_{r}"]) for ion, z in self.valence.items())
        total_i = [sum(z * J[f"{ion}_i"][a] for ion, z in self.valence.items()) for a in range(d)]
        total_e = [-sum(z * J[f"{ion}_e"][a] for ion, z in self.valence.items()) for a in range(d)]
        Im_i = sum(total_i[a] * n[a] for a in range(d))
        Im_e = sum(total_e[a] * n[a] for a in range(d))
        phi_m = ex["phi_i"] - ex["phi_e"]
        for ion in self.valence:                              # I_ch^k = phi_m (passive)
            src[f"f_phi_{ion}"] = sy.diff(phi_m, t) + phi_m - Im_i
        src["f_phi_m"] = sy.diff(phi_m, t) + 3 * phi_m - Im_i
        src["f_gamma"] = Im_i + Im_e
        for ion in self.valence:
            src[f"J_{ion}_e"] = J[f"{ion}_e"]
        return ex, src

    # ---- numeric evaluation
    def evaluate(self, expr, x, t, normal=None):
        """expr at points x (..., dim) and time t; normal (..., dim) for the membrane terms."""
        key = id(expr)
        if key not in self._fn:
            self._fn[key] = (sy.lambdify(list(self.X) + [self.ts] + list(self.nsym), expr, "numpy"), expr)
        fn = self._fn[key][0]
        nn = [normal[..., a] for a in range(self.dim)] if normal is not None else [0.0] * self.dim
        val = fn(*[x[..., a] for a in range(self.dim)], t, *nn)
        return np.broadcast_to(np.asarray(val, dtype=np.float64), x.shape[:-1])


def simplex_quadrature(dim, m=5):
    """Collapsed Gauss-Jacobi rule on the reference triangle / tetrahedron (exact to degree 2m-1):
    barycentric points (n_q, dim+1) and weights summing to one."""
    from scipy.special import roots_jacobi
    g, wg = np.polynomial.legendre.leggauss(m)
    rules = [((g + 1) / 2, wg / 2)]
    for a in range(1, dim):
        xj, wj = roots_jacobi(m, float(a), 0.0)
        rules.append(((xj + 1) / 2, wj / 2 ** (a + 1)))
    pts, wts = [], []
    if dim == 2:
        (t, wt), (u, wu) = rules
        for i in range(m):
            for j in range(m):
                l1, l2 = u[i], t[j] * (1 - u[i])
                pts.append((1 - l1 - l2, l1, l2)); wts.append(wu[i] * wt[j])
    else:
        (c, wc), (b, wb), (a, wa) = rules
        for i in range(m):
            for j in range(m):
                for k in range(m):
                    l1 = a[i]; l2 = b[j] * (1 - a[i]); l3 = c[k] * (1 - a[i]) * (1 - b[j])
                    pts.append((1 - l1 - l2 - l3, l1, l2, l3)); wts.append(wa[i] * wb[j] * wc[k])
    wts = np.array(wts)
    return np.array(pts), wts / wts.sum()


class MMSAssembler:
    """Host-side integrals of the MMS source terms and the error norms for one problem."""

    def __init__(self, problem):
        p = self.p = problem
        lm = p.local_mesh
        self.dim = d = p.mesh.geometry.dim
        self.x = lm.coords
        self.cells = lm.cells
        self.side = p.cell_side
        self.qp, self.qw = simplex_quadrature(d, 5)
        Xc = self.x[self.cells]
        E = Xc[:, 1:, :] - Xc[:, :1, :]
        self.vol = np.abs(np.linalg.det(E)) / math.factorial(d)
        self.xq_c = np.einsum("qa,cad->cqd", self.qp, Xc)
        # membrane facets: quadrature points and the normal pointing from the intra ('+') to the extra side
        fv = p._fv
        Xf = self.x[fv]
        opp = self.x[self.cells[lm.gamma[:, 0], lm.gamma[:, 1]]]
        if d == 2:
            tv = Xf[:, 1] - Xf[:, 0]
            nrm = np.stack([tv[:, 1], -tv[:, 0]], axis=1)
        else:
            nrm = np.cross(Xf[:, 1] - Xf[:, 0], Xf[:, 2] - Xf[:, 0])
        nrm = nrm / np.linalg.norm(nrm, axis=1)[:, None]
        nrm[np.einsum("fd,fd->f", nrm, Xf[:, 0] - opp) < 0] *= -1.0
        self.fv = fv
        self.xq_f = np.einsum("qa,fad->fqd", p.q_pts, Xf)
        self.nq_f = np.broadcast_to(nrm[:, None, :], self.xq_f.shape)
        self.fmeas = p._fmeas

    def rhs_vector(self, node_i, node_e, n_dof):
        """The extra terms of L (KNPEMIx_problem.py:618-626, 646-651) as a vector in the native DoF numbering.
        The exterior-boundary terms (:629-630) only reach Dirichlet rows and are omitted."""
        p = self.p
        M, src = p.M, p.src_terms
        t, dt = float(p.t.value), float(p.dt.value)
        F = float(p.F.value)
        b = np.zeros(n_dof)
        cnode = np.where(self.side[:, None] == 0, node_i[self.cells], node_e[self.cells])
        is_i = (self.side == 0)[:, None]

        def cell_vec(fq):
            return self.vol[:, None] * np.einsum("q,cq,qa->ca", self.qw, fq, self.qp)
        for j, ion in enumerate(p.ion_list):
            name = ion["name"]
            fq = np.where(is_i, M.evaluate(src[f"f_{name}_i"], self.xq_c, t), M.evaluate(src[f"f_{name}_e"], self.xq_c, t))
            np.add.at(b, 4 * cnode + j, dt * cell_vec(fq))
        fq = np.where(is_i, M.evaluate(src["f_phi_i"], self.xq_c, t), M.evaluate(src["f_phi_e"], self.xq_c, t))
        np.add.at(b, 4 * cnode + 3, -dt * cell_vec(fq))
        # membrane terms need alpha_i, alpha_e of the previous concentrations at the facet quadrature points
        lam = p.q_pts
        ki = [p.wh[0][j].numpy()[self.fv] @ lam.T for j in range(p.N_ions)]
        ke = [p.wh[1][j].numpy()[self.fv] @ lam.T for j in range(p.N_ions)]
        wi = [float(ion["Di"].value) * float(ion["z"].value) ** 2 for ion in p.ion_list]
        we = [float(ion["De"].value) * float(ion["z"].value) ** 2 for ion in p.ion_list]
        den_i = sum(w * k for w, k in zip(wi, ki))
        den_e = sum(w * k for w, k in zip(we, ke))

        def facet_vec(gq):
            return self.fmeas[:, None] * np.einsum("q,fq,qa->fa", p.q_w, gq, lam)
        fgam = M.evaluate(src["f_gamma"], self.xq_f, t, self.nq_f)
        fpm = M.evaluate(src["f_phi_m"], self.xq_f, t, self.nq_f)
        ni, ne = node_i[self.fv], node_e[self.fv]
        for j, ion in enumerate(p.ion_list):
            z = float(ion["z"].value)
            fim = M.evaluate(src[f"f_phi_{ion['name']}"], self.xq_f, t, self.nq_f)
            al_i, al_e = wi[j] * ki[j] / den_i, we[j] * ke[j] / den_e
            np.add.at(b, 4 * ni + j, facet_vec(dt / (F * z) * al_i * fim))                      # :622
            np.add.at(b, 4 * ne + j, facet_vec(-dt / (F * z) * al_e * (fim + fgam)))            # :623, :626
        np.add.at(b, 4 * ni + 3, facet_vec(dt * fpm))                                            # :650
        np.add.at(b, 4 * ne + 3, facet_vec(-dt * (fpm + fgam)))                                  # :650-651
        return b

    def l2_errors(self):
        """[Na_i, Na_e, K_i, K_e, Cl_i, Cl_e, phi_i, phi_e] (order of KNPEMIx_problem.py:907)."""
        p = self.p
        t = float(p.t.value)
        out = []
        names = [ion["name"] for ion in p.ion_list] + ["phi"]
        for k, nm in enumerate(names):
            for r, sidx in (("i", 0), ("e", 1)):
                sel = self.side == sidx
                uh = p.wh[sidx][k].numpy()
                uq = np.einsum("qa,ca->cq", self.qp, uh[self.cells[sel]])
                ex = p.M.evaluate(p.exact_sols[f"{nm}_{r}"], self.xq_c[sel], t)
                out.append(math.sqrt(float((self.vol[sel][:, None] * self.qw[None, :] * (uq - ex) ** 2).sum())))
        return out
