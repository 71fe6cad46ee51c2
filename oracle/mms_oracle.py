"""
MMS (method of manufactured solutions) restatement for the CPU oracle  --  TEST INFRASTRUCTURE ONLY.

Follows the reference's verification set-up:
  exact solutions / source terms   src/CGx/utils/setup_mms.py:29-156
  extra terms of the linear form   src/CGx/KNPEMI/KNPEMIx_problem.py:616-651
  Dirichlet data                   src/CGx/KNPEMI/KNPEMIx_problem.py:106-134 (extracellular ions and phi_e on
                                   the whole exterior boundary, interpolated from the exact solution at t = 0)
  initial data                     src/CGx/KNPEMI/KNPEMIx_problem.py:363-431, 746-805 (unit constants, D = 1,
                                   z = (1, 1, -1), passive membrane: I_ch^k = phi_m, setup_mms.py:113-116)
  errors                           src/CGx/KNPEMI/KNPEMIx_problem.py:845-907
  recorded errors (5 levels)       src/CGx/utils/errors.py:8-28 (mesh sequence not recorded)

The exterior-boundary integrals of the form (KNPEMIx_problem.py:629-630) only touch test functions of boundary
vertices, whose rows are Dirichlet rows for the extracellular fields: they are not needed.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
import sympy as sy

import knpemi_oracle as K

# src/CGx/utils/errors.py:8-28, columns Na_i, Na_e, K_i, K_e, Cl_i, Cl_e, phi_i, phi_e ; 5 refinement levels
RECORDED_2D = np.array([
    [0.00901073402128234, 0.0023339935502871972, 0.0005882726326089442, 0.00014694428281715626, 3.6771513073070695e-05],
    [0.031204752571547766, 0.008082930868950062, 0.002037262650989265, 0.000508870841996577, 0.0001257084587271984],
    [0.00900205288154319, 0.0023321107352221207, 0.0005878311072466451, 0.00014683661121181747, 3.6275185147001105e-05],
    [0.010398988757595273, 0.002693769200174614, 0.0006789562744643979, 0.00016959104955482663, 4.1894787274231886e-05],
    [0.018012781610371623, 0.004666101255209068, 0.0011761022248566386, 0.0002937802938437501, 7.257652721500289e-05],
    [0.041603738561854245, 0.010776698866192116, 0.0027162184250172946, 0.0006784617055205049, 0.00016760319277215832],
    [0.0925829489389946, 0.02475728789294459, 0.0062963114451772976, 0.0015807071904532471, 0.0003954393706490666],
    [0.06130402545841192, 0.016651052901881527, 0.00425313689702738, 0.0010691441588083642, 0.0002677078673553347]]).T
RECORDED_3D = np.array([
    [0.006696696268423694, 0.0017887592187751772, 0.00045449546089368006, 0.00011388430384310379, 2.828137545450862e-05],
    [0.03545971349117606, 0.009469839026924316, 0.0024059416956314053, 0.0006028226133204843, 0.00014969304708653826],
    [0.0067008763994408655, 0.0017899729274002977, 0.00045480417916292175, 0.00011396162897019657, 2.8300585814705867e-05],
    [0.011820767425187617, 0.0031568340219538396, 0.0008020360727853242, 0.00020095475667407947, 4.9901129754905946e-05],
    [0.01339756986302902, 0.0035787297426306875, 0.0009092985387635402, 0.00022784551865369447, 5.658184352123676e-05],
    [0.047280479253960374, 0.012626672131834112, 0.003207977429677548, 0.0008037772536968105, 0.00019959414442763504],
    [0.06822405421827074, 0.01932293293267229, 0.0049947086831696645, 0.0012593417059151134, 0.00031550071245278326],
    [0.06740032756597, 0.01960198734939858, 0.005110742558010626, 0.0012916167127701565, 0.00032378829494624634]]).T
NAMES = ["Na_i", "Na_e", "K_i", "K_e", "Cl_i", "Cl_e", "phi_i", "phi_e"]


class MMSTerms:
    """Symbolic exact solutions and source terms (setup_mms.py) -> NumPy callables f(x, y[, z], t, n...)."""

    def __init__(self, dim):
        self.dim = dim
        X = sy.symbols("x y z")[:dim]
        t = sy.Symbol("t")
        N = sy.symbols("nx ny nz")[:dim]
        vt = sy.exp(-t)
        pt = 1
        for c in X:
            vt = vt * sy.sin(2 * sy.pi * c)
            pt = pt * sy.cos(2 * sy.pi * c)
        ex = {"Na_i": 0.7 + 0.3 * vt, "Na_e": 1.0 + 0.6 * vt, "K_i": 0.3 + 0.3 * vt, "K_e": 1.0 + 0.2 * vt,
              "Cl_i": 1.0 + 0.6 * vt, "Cl_e": 2.0 + 0.8 * vt, "phi_i": pt * (1 + sy.exp(-t)), "phi_e": pt}
        z = {"Na": 1, "K": 1, "Cl": -1}
        grad = lambda f: [sy.diff(f, c) for c in X]
        div = lambda v: sum(sy.diff(v[k], X[k]) for k in range(dim))
        J = {}
        for ion in z:
            for r in "ie":
                gk, gp = grad(ex[f"{ion}_{r}"]), grad(ex[f"phi_{r}"])
                J[f"{ion}_{r}"] = [-gk[k] - z[ion] * ex[f"{ion}_{r}"] * gp[k] for k in range(dim)]
        src = {}
        for ion in z:
            for r in "ie":
                src[f"f_{ion}_{r}"] = sy.diff(ex[f"{ion}_{r}"], t) + div(J[f"{ion}_{r}"])
        for r in "ie":
            src[f"f_phi_{r}"] = -sum(z[ion] * div(J[f"{ion}_{r}"]) for ion in z)
        tot_i = [sum(z[ion] * J[f"{ion}_i"][k] for ion in z) for k in range(dim)]
        tot_e = [-sum(z[ion] * J[f"{ion}_e"][k] for ion in z) for k in range(dim)]
        Im_i = sum(tot_i[k] * N[k] for k in range(dim))
        Im_e = sum(tot_e[k] * N[k] for k in range(dim))
        phim = ex["phi_i"] - ex["phi_e"]
        for ion in z:
            src[f"f_phi_{ion}"] = sy.diff(phim, t) + phim - Im_i          # Ich_k = phi_m
        src["f_phi_m"] = sy.diff(phim, t) + 3 * phim - Im_i
        src["f_gamma"] = Im_i + Im_e
        args = list(X) + [t] + list(N)
        self.exact = {k: sy.lambdify(list(X) + [t], v, "numpy") for k, v in ex.items()}
        self.src = {k: sy.lambdify(args, v, "numpy") for k, v in src.items()}
        self.sym = (ex, src, X, t, N)

    def ex(self, name, x, t):
        return np.broadcast_to(self.exact[name](*[x[..., k] for k in range(self.dim)], t), x.shape[:-1]).astype(float)

    def f(self, name, x, t, n=None):
        nn = [n[..., k] for k in range(self.dim)] if n is not None else [0.0] * self.dim
        return np.broadcast_to(self.src[name](*[x[..., k] for k in range(self.dim)], t, *nn), x.shape[:-1]).astype(float)


def cell_quadrature(dim, m=5):
    """Collapsed Gauss-Jacobi rule on the reference simplex (exact to degree 2m-1); barycentric points, weights sum 1."""
    from scipy.special import roots_jacobi
    if dim == 2:
        x0, w0 = np.polynomial.legendre.leggauss(m)
        x1, w1 = roots_jacobi(m, 1.0, 0.0)
        u, wu = 0.5 * (x1 + 1), w1 / 4
        t, wt = 0.5 * (x0 + 1), w0 / 2
        P, W = [], []
        for a in range(m):
            for b in range(m):
                l1, l2 = u[a], t[b] * (1 - u[a])
                P.append((1 - l1 - l2, l1, l2)); W.append(wu[a] * wt[b])
        W = np.array(W)
        return np.array(P), W / W.sum()
    x0, w0 = np.polynomial.legendre.leggauss(m)
    x1, w1 = roots_jacobi(m, 1.0, 0.0)
    x2, w2 = roots_jacobi(m, 2.0, 0.0)
    a_, wa = 0.5 * (x2 + 1), w2 / 8
    b_, wb = 0.5 * (x1 + 1), w1 / 4
    c_, wc = 0.5 * (x0 + 1), w0 / 2
    P, W = [], []
    for i in range(m):
        for j in range(m):
            for k in range(m):
                l1 = a_[i]
                l2 = b_[j] * (1 - a_[i])
                l3 = c_[k] * (1 - a_[i]) * (1 - b_[j])
                P.append((1 - l1 - l2 - l3, l1, l2, l3)); W.append(wa[i] * wb[j] * wc[k])
    W = np.array(W)
    return np.array(P), W / W.sum()


def run_mms(dim, N, dt=1e-5, steps=1, quad_m=5):
    """One MMS run on the unit square / cube with N^dim boxes.  Returns the 8 L2 errors (order of NAMES)."""
    mk = K.unit_square_mesh if dim == 2 else K.unit_cube_mesh
    coords, cells = mk(N)
    tags = K.mark_subdomains(coords, cells)
    params = K.Params(dt=dt, T=1.0, F=1.0, R=1.0, C_M=1.0, z=(1.0, 1.0, -1.0), D=(1.0, 1.0, 1.0))
    o = K.OracleKNPEMI(coords, cells, tags, params=params, models=[K.Model("passive", (4,))], mesh_conversion_factor=1.0)
    T = MMSTerms(dim)
    ions = ["Na", "K", "Cl"]
    x = o.coords
    # initial data: exact at t = 0; phi_m_prev = phi_i_init - phi_e_init = 0 (KNPEMIx_problem.py:756-758)
    for j, ion in enumerate(ions):
        o.k[0][j] = T.ex(f"{ion}_i", x, 0.0)
        o.k[1][j] = T.ex(f"{ion}_e", x, 0.0)
    o.phi_m = np.zeros(o.n_v)
    # Dirichlet dofs: extracellular fields on the exterior boundary (values: exact at t = 0)
    on_bdry = np.any((np.abs(x) < 1e-14) | (np.abs(x - 1.0) < 1e-14), axis=1)
    bv = np.nonzero(on_bdry)[0]
    bc_dofs, bc_vals = [], []
    for j, ion in enumerate(ions):
        bc_dofs.append(4 * o.lay.node_e[bv] + j); bc_vals.append(T.ex(f"{ion}_e", x[bv], 0.0))
    bc_dofs.append(4 * o.lay.node_e[bv] + 3); bc_vals.append(T.ex("phi_e", x[bv], 0.0))
    bc_dofs = np.concatenate(bc_dofs); bc_vals = np.concatenate(bc_vals)
    qp, qw = cell_quadrature(dim, quad_m)
    # facet normals '+' -> '-' (outward of the intracellular cell)
    Xf = o.coords[o.fv]
    opp = o.coords[o.cells[o.gamma[:, 0], o.gamma[:, 1]]]
    if dim == 2:
        tvec = Xf[:, 1] - Xf[:, 0]
        nrm = np.stack([tvec[:, 1], -tvec[:, 0]], axis=1)
    else:
        nrm = np.cross(Xf[:, 1] - Xf[:, 0], Xf[:, 2] - Xf[:, 0])
    nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    flip = np.einsum("fd,fd->f", nrm, Xf[:, 0] - opp) < 0
    nrm[flip] *= -1
    xq_f = np.einsum("qa,fad->fqd", o.lamq, Xf)                          # (n_g, n_q, dim)
    nq_f = np.broadcast_to(nrm[:, None, :], xq_f.shape)
    Xc = o.coords[o.cells]
    xq_c = np.einsum("qa,cad->cqd", qp, Xc)                              # (n_c, n_q, dim)
    for step in range(steps):
        o.t += dt
        t = o.t
        A = o.assemble_A().tolil()
        b = o.assemble_b()
        # volume sources (KNPEMIx_problem.py:618-619, 646-647)
        side = o.cell_side
        for j, ion in enumerate(ions):
            fi, fe = T.f(f"f_{ion}_i", xq_c, t), T.f(f"f_{ion}_e", xq_c, t)
            fq = np.where(side[:, None] == 0, fi, fe)
            loc = dt * o.vol[:, None] * np.einsum("q,cq,qa->ca", qw, fq, qp)
            np.add.at(b, 4 * o.cnode + j, loc)
        fq = np.where(side[:, None] == 0, T.f("f_phi_i", xq_c, t), T.f("f_phi_e", xq_c, t))
        loc = -dt * o.vol[:, None] * np.einsum("q,cq,qa->ca", qw, fq, qp)
        np.add.at(b, 4 * o.cnode + 3, loc)
        # membrane sources (KNPEMIx_problem.py:622-626, 650-651)
        al_i, al_e = o._alpha_q(0), o._alpha_q(1)
        fgam = T.f("f_gamma", xq_f, t, nq_f)
        fpm = T.f("f_phi_m", xq_f, t, nq_f)
        vec = lambda g: o.fmeas[:, None] * np.einsum("q,fq,qa->fa", o.qw, g, o.lamq)
        for j, ion in enumerate(ions):
            z = params.z[j]
            fim = T.f(f"f_phi_{ion}", xq_f, t, nq_f)
            np.add.at(b, 4 * o.fnode_i + j, vec(+dt / z * al_i[j] * fim))
            np.add.at(b, 4 * o.fnode_e + j, vec(-dt / z * al_e[j] * fim - dt / z * al_e[j] * fgam))
        np.add.at(b, 4 * o.fnode_i + 3, vec(dt * fpm))
        np.add.at(b, 4 * o.fnode_e + 3, vec(-dt * fpm - dt * fgam))
        # Dirichlet rows (DOLFINx: zero row/col + lifting; row replacement gives the same solution)
        A = A.tocsr()
        keep = np.ones(o.n_dof, dtype=bool); keep[bc_dofs] = False
        g = np.zeros(o.n_dof); g[bc_dofs] = bc_vals
        rhs = b - A @ g
        idx = np.nonzero(keep)[0]
        xs = g.copy()
        xs[idx] = spla.splu(A[idx][:, idx].tocsc()).solve(rhs[idx])
        o.unpack(xs)
    # L2 errors with the cell rule
    t = o.t
    errs = []
    for name in NAMES:
        ion, r = name.rsplit("_", 1)
        sidx = 0 if r == "i" else 1
        uh = o.phi[sidx] if ion == "phi" else o.k[sidx][ions.index(ion)]
        sel = o.cell_side == sidx
        uq = np.einsum("qa,ca->cq", qp, uh[o.cells[sel]])
        eq = uq - T.ex(name, xq_c[sel], t)
        errs.append(math.sqrt(float((o.vol[sel][:, None] * qw[None, :] * eq ** 2).sum())))
    return np.array(errs)


if __name__ == "__main__":
    import sys
    dim = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    Ns = [int(a) for a in sys.argv[2:]] or ([8, 16, 32, 64] if dim == 2 else [4, 8, 16])
    rec = RECORDED_2D if dim == 2 else RECORDED_3D
    prev = None
    for N in Ns:
        e = run_mms(dim, N)
        rate = np.log2(prev / e) if prev is not None else np.full(8, np.nan)
        print(f"dim {dim} N {N:4d} errors " + " ".join(f"{v:.4e}" for v in e))
        print("            rates  " + " ".join(f"{v:10.2f}" for v in rate))
        prev = e
    print("recorded (errors.py):")
    for lvl in range(5):
        print("   level", lvl, " ".join(f"{v:.4e}" for v in rec[lvl]))
