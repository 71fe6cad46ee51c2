"""
CPU ORACLE for the KNP-EMI assemble-and-solve hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain NumPy/SciPy *restatement* of the algorithm that the
reference (hherlyng/knp-emi-cgx) delegates to DOLFINx / multiphenicsx / PETSc.
It is the checker for the HIP path.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it; the product package
(``knp-emi-cgx_amd/``) never does.

Pinning status
--------------
* 2D (unit square): PINNED.  ``tests/test_oracle_pins.py`` reproduces the
  reference's own known answers
  (``tests/KNPEMI/electric_potential_norms_iterative_solver.py:58-59`` to the
  reference's tolerance 1e-7 rel. on phi_i, and BOTH norms of ``..._direct_solver.py:55-56``
  to 3e-10 in the zero-mean gauge PETSc gives a preonly/LU solve with an attached null space
  -- no fitted constant).
* 3D (unit cube): PINNED through the manufactured-solution path.  ``oracle/mms_oracle.py``
  restates the reference's MMS set-up on top of this file and reproduces the reference's
  recorded L2 errors (src/CGx/utils/errors.py:8-28) to 5 significant digits for the potentials on
  N = 8, 16, 32 (3D) and N = 8 ... 128 (2D) -- ``tests/test_oracle_mms.py``.  The 3D facet rule
  here (collapsed Gauss-Jacobi 6x6, exact to degree 11) differs from basix's Xiao-Gimbutas
  25-point degree-10 rule (third-party, tables not in /root/reference); both integrate the smooth
  Gamma integrands far below those 5 digits.

Reference lines restated (all relative to /root/reference):
  meshes/markers   src/CGx/utils/generate_square_mesh.py:28-42, src/CGx/utils/misc.py:99-195,256-398
  '+' = intra      src/CGx/utils/mixed_dim_problem.py:705-733
  restrictions     src/CGx/KNPEMI/KNPEMIx_problem.py:28-94
  forms a, L       src/CGx/KNPEMI/KNPEMIx_problem.py:454-655
  form P           src/CGx/KNPEMI/KNPEMIx_problem.py:657-744
  constants        src/CGx/KNPEMI/KNPEMIx_problem.py:909-981
  mechanisms       src/CGx/KNPEMI/KNPEMIx_ionic_model.py (whole file)
  time loop        src/CGx/KNPEMI/KNPEMIx_solver.py:337-468
  null space       src/CGx/KNPEMI/KNPEMIx_solver.py:297-335
  L2 norms         src/CGx/KNPEMI/main.py:70-84

Unknown numbering used here (and by the HIP path): "nodes" are (vertex, side)
pairs; vertex v contributes an intra node if it touches an intra cell and an
extra node if it touches an extra cell (membrane vertices contribute both,
intra first).  DoF = 4*node + f with f = 0,1,2 the ions (Na, K, Cl) and f = 3
the potential.  The reference's block ordering [k_i.., phi_i | k_e.., phi_e] is a
permutation of this (``Layout.reference_permutation``).
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from scipy.special import roots_jacobi

# --------------------------------------------------------------------------------------
# Meshes and markers
# --------------------------------------------------------------------------------------

def unit_square_mesh(N: int):
    """dolfinx.mesh.create_unit_square(N, N) with the default 'right' diagonal
    (generate_square_mesh.py:28): every box is split into (v0,v1,v3),(v0,v2,v3)."""
    xs = np.arange(N + 1, dtype=np.float64) / N
    X, Y = np.meshgrid(xs, xs, indexing="xy")
    coords = np.stack([X.ravel(), Y.ravel()], axis=1)
    ix, iy = np.meshgrid(np.arange(N), np.arange(N), indexing="xy")
    v0 = (iy * (N + 1) + ix).ravel()
    v1 = v0 + 1
    v2 = v0 + (N + 1)
    v3 = v2 + 1
    cells = np.empty((2 * N * N, 3), dtype=np.int32)
    cells[0::2] = np.stack([v0, v1, v3], axis=1)
    cells[1::2] = np.stack([v0, v2, v3], axis=1)
    return coords, cells


def unit_cube_mesh(N: int):
    """dolfinx.mesh.create_unit_cube(N, N, N): six tetrahedra per box, all sharing
    the v0-v7 diagonal."""
    xs = np.arange(N + 1, dtype=np.float64) / N
    Z, Y, X = np.meshgrid(xs, xs, xs, indexing="ij")
    coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    n1 = N + 1
    iz, iy, ix = np.meshgrid(np.arange(N), np.arange(N), np.arange(N), indexing="ij")
    v0 = (iz * n1 * n1 + iy * n1 + ix).ravel()
    v1 = v0 + 1
    v2 = v0 + n1
    v3 = v1 + n1
    v4 = v0 + n1 * n1
    v5 = v1 + n1 * n1
    v6 = v2 + n1 * n1
    v7 = v3 + n1 * n1
    tets = [(v0, v1, v3, v7), (v0, v1, v7, v5), (v0, v5, v7, v4),
            (v0, v3, v2, v7), (v0, v6, v4, v7), (v0, v2, v6, v7)]
    cells = np.empty((6 * N ** 3, 4), dtype=np.int32)
    for k, t in enumerate(tets):
        cells[k::6] = np.stack(t, axis=1)
    return coords, cells


def mark_subdomains(coords, cells, lo=0.25, hi=0.75):
    """misc.py:99-135 / 256-297: tag 1 for cells with all vertices in [lo,hi]^d, else 2."""
    inside_v = np.all((coords >= lo) & (coords <= hi), axis=1)
    inside_c = np.all(inside_v[cells], axis=1)
    return np.where(inside_c, 1, 2).astype(np.int32)


def interface_facets(cells, cell_side):
    """All facets shared by an intra (side 0) and an extra (side 1) cell, as rows
    (cell+, lf+, cell-, lf-) with '+' = intra (mixed_dim_problem.py:717-729).
    Local facet lf is the one opposite local vertex lf (DOLFINx convention)."""
    n_c, nv = cells.shape
    keys = []
    for lf in range(nv):
        fv = np.delete(cells, lf, axis=1)
        fv = np.sort(fv, axis=1)
        keys.append(fv)
    keys = np.concatenate(keys, axis=0)                       # (nv*n_c, nv-1), block lf
    owner_cell = np.tile(np.arange(n_c), nv)
    owner_lf = np.repeat(np.arange(nv), n_c)
    order = np.lexsort(keys.T[::-1])
    ks = keys[order]
    same = np.all(ks[1:] == ks[:-1], axis=1)
    i0 = order[:-1][same]
    i1 = order[1:][same]
    c0, c1 = owner_cell[i0], owner_cell[i1]
    l0, l1 = owner_lf[i0], owner_lf[i1]
    s0, s1 = cell_side[c0], cell_side[c1]
    mixed = s0 != s1
    c0, c1, l0, l1, s0 = c0[mixed], c1[mixed], l0[mixed], l1[mixed], s0[mixed]
    swap = s0 == 1
    cp = np.where(swap, c1, c0)
    lp = np.where(swap, l1, l0)
    cm = np.where(swap, c0, c1)
    lm = np.where(swap, l0, l1)
    gamma = np.stack([cp, lp, cm, lm], axis=1).astype(np.int32)
    # deterministic order: by (cell+, lf+)
    o = np.lexsort((gamma[:, 1], gamma[:, 0]))
    return gamma[o]


# --------------------------------------------------------------------------------------
# Facet quadrature (degree 10, mixed_dim_problem.py:733)
# --------------------------------------------------------------------------------------

def facet_quadrature(dim: int):
    """Barycentric points (n_q, dim) and weights (sum 1) on the reference facet of a
    dim-dimensional simplex mesh.  dim=2: 6-point Gauss-Legendre on the edge
    (what basix's Gauss-Jacobi scheme gives for degree 10).  dim=3: collapsed
    6x6 Gauss-Jacobi on the triangle (exact to degree 11)."""
    m = 6
    if dim == 2:
        x, w = np.polynomial.legendre.leggauss(m)
        s = 0.5 * (x + 1.0)
        pts = np.stack([1.0 - s, s], axis=1)
        return pts, 0.5 * w
    if dim == 3:
        x0, w0 = np.polynomial.legendre.leggauss(m)            # weight 1 on [-1,1]
        x1, w1 = roots_jacobi(m, 1.0, 0.0)                     # weight (1-x) on [-1,1]
        u = 0.5 * (x1 + 1.0)
        wu = w1 / 4.0                                          # int_0^1 (1-u) f du
        t = 0.5 * (x0 + 1.0)
        wt = w0 / 2.0
        U, T = np.meshgrid(u, t, indexing="ij")
        W = np.outer(wu, wt)
        l1 = U.ravel()
        l2 = (T * (1.0 - U)).ravel()
        l0 = 1.0 - l1 - l2
        w = W.ravel()
        return np.stack([l0, l1, l2], axis=1), w / w.sum()
    raise ValueError("dim must be 2 or 3")


# --------------------------------------------------------------------------------------
# Parameters
# --------------------------------------------------------------------------------------

@dataclass
class Params:
    """Physical constants and defaults (KNPEMIx_problem.py:909-981,
    mixed_dim_problem.py:186-200,290-332)."""
    dt: float = 2.5e-5
    T: float = 300.0
    F: float = 96485.0
    R: float = 8.314
    C_M: float = 0.02
    z: tuple = (1.0, 1.0, -1.0)
    D: tuple = (1.33e-9, 1.96e-9, 2.03e-9)          # Di = De (KNPEMIx_problem.py:929-931,977-979)
    phi_rest: float = -0.065
    # conductances: defaults when a 'stimulus' section is present (mixed_dim_problem.py:311-318)
    g_Na_bar: float = 1200.0
    g_K_bar: float = 360.0
    g_leak: tuple = (0.3, 0.1, 0.25)
    g_leak_g: tuple = (1.0, 16.96, 2.0)
    g_syn_bar: float = 1e-9
    a_syn: float = 5e-4
    T_stim: float = 1.0
    scale_stimulus: bool = True
    # initial conditions (configs/tests/*.yaml)
    phi_m_init: float = -0.070
    ki_init: tuple = (12.0, 130.0, 5.0)
    ke_init: tuple = (140.0, 4.0, 125.0)
    n_init: float = 0.276
    m_init: float = 0.0379
    h_init: float = 0.688
    K_e_init: float = 4.0

    @property
    def psi(self):
        return self.R * self.T / self.F


# --------------------------------------------------------------------------------------
# Geometry + layout
# --------------------------------------------------------------------------------------

@dataclass
class Layout:
    n_v: int
    node_i: np.ndarray
    node_e: np.ndarray
    n_nodes: int
    node_vertex: np.ndarray
    node_side: np.ndarray

    @property
    def n_dof(self):
        return 4 * self.n_nodes

    def reference_permutation(self):
        """perm such that x_reference_order = x_native[perm]: blocks
        [k_i^0,k_i^1,k_i^2,phi_i,k_e^0,...,phi_e], each in vertex order
        (KNPEMIx_problem.py:46-48,92-94)."""
        out = []
        for side, nodes in ((0, self.node_i), (1, self.node_e)):
            nn = nodes[nodes >= 0]
            for f in range(4):
                out.append(4 * nn + f)
        return np.concatenate(out)


def build_layout(n_v, cells, cell_side) -> Layout:
    in_i = np.zeros(n_v, dtype=bool)
    in_e = np.zeros(n_v, dtype=bool)
    in_i[cells[cell_side == 0].ravel()] = True
    in_e[cells[cell_side == 1].ravel()] = True
    cnt = in_i.astype(np.int64) + in_e.astype(np.int64)
    start = np.cumsum(cnt) - cnt
    node_i = np.where(in_i, start, -1)
    node_e = np.where(in_e, start + in_i, -1)
    n_nodes = int(cnt.sum())
    node_vertex = np.empty(n_nodes, dtype=np.int64)
    node_side = np.empty(n_nodes, dtype=np.int8)
    vi = np.nonzero(in_i)[0]
    ve = np.nonzero(in_e)[0]
    node_vertex[node_i[vi]] = vi
    node_side[node_i[vi]] = 0
    node_vertex[node_e[ve]] = ve
    node_side[node_e[ve]] = 1
    return Layout(n_v, node_i, node_e, n_nodes, node_vertex, node_side)


def cell_geometry(coords, cells):
    """Volumes and barycentric gradients of P1 simplices."""
    d = coords.shape[1]
    X = coords[cells]                                   # (n_c, d+1, d)
    J = X[:, 1:, :] - X[:, :1, :]                       # rows = edge vectors (n_c, d, d)
    detJ = np.linalg.det(J)
    vol = np.abs(detJ) / math.factorial(d)
    Jinv = np.linalg.inv(J)                             # (n_c, d, d); columns = grad lambda_{1..d}
    G = np.empty((cells.shape[0], d + 1, d))
    G[:, 1:, :] = np.transpose(Jinv, (0, 2, 1))
    G[:, 0, :] = -G[:, 1:, :].sum(axis=1)
    return vol, G


def facet_geometry(coords, cells, gamma):
    d = coords.shape[1]
    cp, lp = gamma[:, 0], gamma[:, 1]
    nv = d + 1
    # facet vertices = vertices of cell+ except local lp, in increasing local order
    loc = np.array([[a for a in range(nv) if a != lf] for lf in range(nv)])  # (nv, d)
    fv = cells[cp[:, None], loc[lp]]                                         # (n_g, d)
    X = coords[fv]
    if d == 2:
        meas = np.linalg.norm(X[:, 1] - X[:, 0], axis=1)
    else:
        meas = 0.5 * np.linalg.norm(np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), axis=1)
    return fv, meas


# --------------------------------------------------------------------------------------
# The discrete problem
# --------------------------------------------------------------------------------------

@dataclass
class Model:
    """One membrane mechanism instance: kind in {'passive','hh','atp','neuronal_ct',
    'glial_ct','kir_nak'}; tags = membrane tags it acts on."""
    kind: str
    tags: tuple
    use_rush_larsen: bool = True
    time_steps_ode: int = 25


class OracleKNPEMI:
    """State + operators of one KNP-EMI problem on one mesh (serial)."""

    def __init__(self, coords, cells, cell_tag, intra_tags=(1,), extra_tag=2,
                 gamma=None, gamma_tag=None, params: Params | None = None,
                 models=None, stimulus_tags=None, mesh_conversion_factor=1.0,
                 stimulus_region=None):
        self.p = params or Params()
        self.dim = coords.shape[1]
        self.coords = np.ascontiguousarray(coords, dtype=np.float64) * mesh_conversion_factor
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)
        self.cell_tag = np.asarray(cell_tag)
        self.cell_side = np.where(np.isin(self.cell_tag, intra_tags), 0, 1).astype(np.int8)
        if gamma is None:
            gamma = interface_facets(self.cells, self.cell_side)
            gamma_tag = np.full(gamma.shape[0], 4, dtype=np.int32)
        self.gamma = gamma
        self.gamma_tag = np.asarray(gamma_tag)
        self.gamma_tags = tuple(sorted(set(self.gamma_tag.tolist())))
        self.models = models if models is not None else [Model("passive", self.gamma_tags)]
        self.stimulus_tags = tuple(stimulus_tags) if stimulus_tags is not None else self.gamma_tags
        self.stimulus_region = stimulus_region        # (axis, lo, hi) in scaled coords, a list of such triples, or None
        self.n_v = self.coords.shape[0]
        self.lay = build_layout(self.n_v, self.cells, self.cell_side)
        self.vol, self.G = cell_geometry(self.coords, self.cells)
        self.fv, self.fmeas = facet_geometry(self.coords, self.cells, self.gamma)
        self.qp, self.qw = facet_quadrature(self.dim)
        self.t = 0.0
        self.t_mod = 0.0
        self._set_initial_conditions()
        self._precompute()

    # ---- initial conditions (KNPEMIx_problem.py:326-338,386-450; solver :177-199)
    def _set_initial_conditions(self):
        p = self.p
        nv = self.n_v
        self.k = [[np.full(nv, p.ki_init[j]) for j in range(3)],
                  [np.full(nv, p.ke_init[j]) for j in range(3)]]
        self.phi = [np.full(nv, p.phi_m_init), np.zeros(nv)]
        self.phi_m = np.full(nv, p.phi_m_init)
        self.n = np.full(nv, p.n_init)
        self.m = np.full(nv, p.m_init)
        self.h = np.full(nv, p.h_init)

    def _precompute(self):
        d = self.dim
        nv1 = d + 1
        self.Kloc = self.vol[:, None, None] * np.einsum("cad,cbd->cab", self.G, self.G)
        Mref = (np.ones((nv1, nv1)) + np.eye(nv1)) / ((d + 1) * (d + 2))
        self.Mloc = self.vol[:, None, None] * Mref[None]
        node = np.where(self.cell_side[:, None] == 0, self.lay.node_i[self.cells], self.lay.node_e[self.cells])
        assert node.min() >= 0
        self.cnode = node                                           # (n_c, d+1)
        self.rowsA = np.repeat(node[:, :, None], nv1, axis=2)       # node of row a
        self.colsA = np.repeat(node[:, None, :], nv1, axis=1)       # node of col b
        self.lamq = self.qp                                         # (n_q, d) facet basis table
        # stimulus area (KNPEMIx_ionic_model.py:591-601)
        stim = np.isin(self.gamma_tag, self.stimulus_tags)
        self.stim_facet = stim
        self.stimulus_area = float((self.fmeas * self._facet_mask_integral())[stim].sum()) if stim.any() else 1.0
        # facet basis tables
        self.lamq = self.qp                                         # (n_q, d)
        self.fnode_i = self.lay.node_i[self.fv]
        self.fnode_e = self.lay.node_e[self.fv]
        assert self.fnode_i.min() >= 0 and self.fnode_e.min() >= 0

    def _mask_q(self):
        """Stimulus-region mask at facet quadrature points (ionic_model.py:557-586)."""
        if self.stimulus_region is None:
            return None
        # one (axis, lo, hi) or -- `multiple_stimulus_directions`, ionic_model.py:573-586 -- a list of them whose masks multiply
        regions = [self.stimulus_region] if np.isscalar(self.stimulus_region[0]) else list(self.stimulus_region)
        mask = 1.0
        for ax, lo, hi in regions:
            xq = np.einsum("qa,fa->fq", self.lamq, self.coords[self.fv][:, :, int(ax)])
            mask = mask * ((xq > lo) & (xq < hi)).astype(np.float64)
        return mask

    def _facet_mask_integral(self):
        m = self._mask_q()
        if m is None:
            return np.ones(self.gamma.shape[0])
        return m @ self.qw

    # ---- helpers
    @property
    def n_dof(self):
        return self.lay.n_dof

    def _at_q(self, nodal):
        """P1 field restricted to Gamma facets, evaluated at quadrature points: (n_g, n_q)."""
        return nodal[self.fv] @ self.lamq.T

    def pack(self):
        """Nodal fields -> native solution vector (solver :177-209)."""
        x = np.zeros(self.n_dof)
        L = self.lay
        for side, nodes in ((0, L.node_i), (1, L.node_e)):
            v = np.nonzero(nodes >= 0)[0]
            nn = nodes[v]
            for j in range(3):
                x[4 * nn + j] = self.k[side][j][v]
            x[4 * nn + 3] = self.phi[side][v]
        return x

    def unpack(self, x):
        """solver :452-468 -- fields are zero outside their restriction."""
        L = self.lay
        for side, nodes in ((0, L.node_i), (1, L.node_e)):
            v = np.nonzero(nodes >= 0)[0]
            nn = nodes[v]
            for j in range(3):
                a = np.zeros(self.n_v)
                a[v] = x[4 * nn + j]
                self.k[side][j] = a
            a = np.zeros(self.n_v)
            a[v] = x[4 * nn + 3]
            self.phi[side] = a
        self.phi_m = self.phi[0] - self.phi[1]

    def nullspace(self):
        ns = np.zeros(self.n_dof)
        ns[3::4] = 1.0
        return ns / np.linalg.norm(ns)

    # ---- alpha fractions at Gamma quadrature points (KNPEMIx_problem.py:512-513,582-583)
    def _alpha_q(self, side):
        p = self.p
        kq = [self._at_q(self.k[side][j]) for j in range(3)]
        den = sum(p.D[j] * p.z[j] ** 2 * kq[j] for j in range(3))
        return [p.D[j] * p.z[j] ** 2 * kq[j] / den for j in range(3)]

    def _facet_mass(self, wq=None):
        """M_Gamma[w]_ab per facet: (n_g, d, d)."""
        lam = self.lamq
        if wq is None:
            base = np.einsum("q,qa,qb->ab", self.qw, lam, lam)
            return self.fmeas[:, None, None] * base[None]
        return self.fmeas[:, None, None] * np.einsum("q,fq,qa,qb->fab", self.qw, wq, lam, lam)

    # ---- system matrix A (KNPEMIx_problem.py:586-604,633-638)
    def assemble_A(self):
        p = self.p
        dt, psi, F, C_M = p.dt, p.psi, p.F, p.C_M
        rows, cols, vals = [], [], []
        R4, C4 = 4 * self.rowsA, 4 * self.colsA
        cbar = []
        for j in range(3):
            kj = np.where(self.cell_side[:, None] == 0, self.k[0][j][self.cells], self.k[1][j][self.cells])
            cbar.append(kj.mean(axis=1))
        phiphi = np.zeros_like(self.Kloc)
        for j in range(3):
            D, z = p.D[j], p.z[j]
            rows += [R4 + j, R4 + j, R4 + 3]
            cols += [C4 + j, C4 + 3, C4 + j]
            vals += [self.Mloc + dt * D * self.Kloc,
                     dt * D * z / psi * cbar[j][:, None, None] * self.Kloc,
                     dt * z * D * self.Kloc]
            phiphi += dt * D * z * z / psi * cbar[j][:, None, None] * self.Kloc
        rows.append(R4 + 3); cols.append(C4 + 3); vals.append(phiphi)
        # membrane terms
        ni = np.repeat(self.fnode_i[:, :, None], self.dim, axis=2)       # row node a (intra copy)
        ne = np.repeat(self.fnode_e[:, :, None], self.dim, axis=2)
        nib = np.repeat(self.fnode_i[:, None, :], self.dim, axis=1)      # col node b
        neb = np.repeat(self.fnode_e[:, None, :], self.dim, axis=1)
        al_i, al_e = self._alpha_q(0), self._alpha_q(1)
        for j in range(3):
            Ci = self._facet_mass(al_i[j] * C_M / (F * p.z[j]))
            Ce = self._facet_mass(al_e[j] * C_M / (F * p.z[j]))
            rows += [4 * ni + j, 4 * ni + j, 4 * ne + j, 4 * ne + j]
            cols += [4 * nib + 3, 4 * neb + 3, 4 * neb + 3, 4 * nib + 3]
            vals += [Ci, -Ci, Ce, -Ce]
        Mg = (C_M / F) * self._facet_mass()
        rows += [4 * ni + 3, 4 * ni + 3, 4 * ne + 3, 4 * ne + 3]
        cols += [4 * nib + 3, 4 * neb + 3, 4 * neb + 3, 4 * nib + 3]
        vals += [Mg, -Mg, Mg, -Mg]
        r = np.concatenate([a.ravel() for a in rows])
        c = np.concatenate([a.ravel() for a in cols])
        v = np.concatenate([a.ravel() for a in vals])
        A = sp.coo_matrix((v, (r, c)), shape=(self.n_dof, self.n_dof)).tocsr()
        A.sort_indices()
        return A

    # ---- preconditioner matrix P, block-Jacobi form (KNPEMIx_problem.py:717-738)
    def assemble_P(self):
        p = self.p
        dt, psi, F, C_M = p.dt, p.psi, p.F, p.C_M
        rows, cols, vals = [], [], []
        R4, C4 = 4 * self.rowsA, 4 * self.colsA
        phiphi = np.zeros_like(self.Kloc)
        for j in range(3):
            D, z = p.D[j], p.z[j]
            kj = np.where(self.cell_side[:, None] == 0, self.k[0][j][self.cells], self.k[1][j][self.cells])
            cb = kj.mean(axis=1)
            rows.append(R4 + j); cols.append(C4 + j); vals.append(self.Mloc + dt * D * self.Kloc)
            phiphi += dt * D * z * z / psi * cb[:, None, None] * self.Kloc
        rows.append(R4 + 3); cols.append(C4 + 3); vals.append(phiphi)
        Mg = (C_M / F) * self._facet_mass()
        ni = np.repeat(self.fnode_i[:, :, None], self.dim, axis=2)
        ne = np.repeat(self.fnode_e[:, :, None], self.dim, axis=2)
        nib = np.repeat(self.fnode_i[:, None, :], self.dim, axis=1)
        neb = np.repeat(self.fnode_e[:, None, :], self.dim, axis=1)
        rows += [4 * ni + 3, 4 * ne + 3]; cols += [4 * nib + 3, 4 * neb + 3]; vals += [-Mg, -Mg]
        r = np.concatenate([a.ravel() for a in rows])
        c = np.concatenate([a.ravel() for a in cols])
        v = np.concatenate([a.ravel() for a in vals])
        P = sp.coo_matrix((v, (r, c)), shape=(self.n_dof, self.n_dof)).tocsr()
        P.sort_indices()
        return P

    # ---- mechanisms at quadrature points (KNPEMIx_ionic_model.py)
    def channel_currents_q(self):
        """I_ch^k at Gamma quadrature points: (3, n_g, n_q), summed over the mechanisms
        active on each facet's tag (KNPEMIx_problem.py:504-555)."""
        p = self.p
        psi = p.psi
        ki = [self._at_q(self.k[0][j]) for j in range(3)]
        ke = [self._at_q(self.k[1][j]) for j in range(3)]
        phim = self._at_q(self.phi_m)
        E = [(psi / p.z[j]) * np.log(ke[j] / ki[j]) for j in range(3)]          # :516
        nq, mq, hq = self._at_q(self.n), self._at_q(self.m), self._at_q(self.h)
        I = np.zeros((3,) + phim.shape)
        for mdl in self.models:
            on = np.isin(self.gamma_tag, mdl.tags)[:, None].astype(np.float64)
            cur = [0.0, 0.0, 0.0]
            if mdl.kind == "passive":                                           # :89-91
                cur = [phim, phim, phim]
            elif mdl.kind == "hh":                                              # :487-515
                g = [p.g_leak[0] + p.g_Na_bar * mq ** 3 * hq,
                     p.g_leak[1] + p.g_K_bar * nq ** 4,
                     p.g_leak[2] + 0.0 * phim]
                cur = [g[j] * (phim - E[j]) for j in range(3)]
                # stimulus on Na for stimulated tags (KNPEMIx_problem.py:531-544; ionic_model :548-603)
                stim_on = np.isin(self.gamma_tag, self.stimulus_tags)[:, None].astype(np.float64)
                stim = p.g_syn_bar * math.exp(-self.t_mod / p.a_syn) * (phim - E[0])
                mq_ = self._mask_q()
                if mq_ is not None:
                    stim = stim * mq_
                if p.scale_stimulus:
                    stim = stim / self.stimulus_area
                cur[0] = cur[0] + stim_on * stim
            elif mdl.kind == "atp":                                             # :389-424
                par1 = 1.0 + 1.5 / ke[1]
                par2 = 1.0 + 10.0 / ki[0]
                Iatp = 0.25 / (par1 ** 2 * par2 ** 3)
                cur = [3.0 * Iatp, -2.0 * Iatp, 0.0 * Iatp]
            elif mdl.kind == "neuronal_ct":                                     # :310-369
                Ikcc2 = 0.0068 * np.log((ki[1] * ki[2]) / (ke[1] * ke[2]))
                # f_NKCC1 always returns zero in the reference (:62-69, truthiness of a UFL expr)
                Inkcc1 = 0.0 * Ikcc2
                cur = [-Inkcc1, -Inkcc1 + Ikcc2, Inkcc1 - Ikcc2]
            elif mdl.kind == "glial_ct":                                        # :234-298
                Ikcc1 = 7e-2 * psi * np.log((ki[1] * ki[2]) / (ke[1] * ke[2]))
                Inkcc1 = 0.0 * Ikcc1
                cur = [-Inkcc1, -Inkcc1 + Ikcc1, 2.0 * Inkcc1 - Ikcc1]
            elif mdl.kind == "kir_nak":                                         # :125-222
                # E_K_init is frozen in the model constructor (:117), which the reference's drivers call
                # BEFORE set_initial_conditions (main.py:32-47): it therefore sees the class defaults
                # K_e_init = 3, K_i_g_init = 100 (KNPEMIx_problem.py:945,960), not the YAML values.
                E_K_init = psi * math.log(3.0 / 100.0)
                pump = (1.0 / (1.0 + (10.0 / ki[0]) ** 1.5)) * (1.0 / (1.0 + 1.5 / ke[1])) * (1.1 * 1.12e-6)
                Fc = p.F
                dphi = phim - E[1]
                A_ = 1 + math.exp(0.433)
                B_ = 1 + math.exp(-(0.1186 + E_K_init) / 0.0441)
                C_ = 1 + np.exp((dphi + 0.0185) / 0.0425)
                D_ = 1 + np.exp(-(0.1186 + phim) / 0.0441)
                fkir = np.sqrt(ke[1] / p.K_e_init) * A_ * B_ / (C_ * D_)
                cur = [p.g_leak_g[0] * (phim - E[0]) + 3 * p.z[0] * Fc * pump,
                       fkir * p.g_leak_g[1] * (phim - E[1]) - 2 * p.z[1] * Fc * pump,
                       p.g_leak_g[2] * (phim - E[2])]
            else:
                raise ValueError(mdl.kind)
            for j in range(3):
                I[j] += on * cur[j]
        return I

    # ---- right-hand side (KNPEMIx_problem.py:600-614,641-642)
    def set_ion_injection(self, current=5e-9):
        """``source_terms: ion_injection`` (src/CGx/utils/mixed_dim_problem.py:467-541, 806-811 and
        src/CGx/KNPEMI/KNPEMIx_problem.py:200-218): K and Cl sources f_e = I/(F vol) on the vertices of the
        cells that lie inside the cube of half-width (x_max - x_min)/10 around the mesh centre."""
        x = np.zeros((self.coords.shape[0], 3))
        x[:, :self.dim] = self.coords
        lo, hi = self.coords.min(axis=0), self.coords.max(axis=0)
        c = np.zeros(3)
        c[:self.dim] = (lo + hi) / 2
        delta = (hi[0] - lo[0]) / 10
        tol = 1e-14
        inside = ((x >= c - delta - tol) & (x <= c + delta + tol)).all(axis=1)
        cells = np.nonzero(inside[self.cells].all(axis=1))[0]
        X = self.coords[self.cells[cells]]
        vol = (np.abs(np.linalg.det(X[:, 1:, :] - X[:, :1, :])) / math.factorial(self.dim)).sum()
        self.injection_cells, self.injection_volume = cells, vol
        f = np.zeros(self.coords.shape[0])
        f[np.unique(self.cells[cells])] = current / self.p.F / vol
        self.f_e = [None, f, f.copy()]

    def assemble_b(self):
        p = self.p
        dt, F, C_M = p.dt, p.F, p.C_M
        b = np.zeros(self.n_dof)
        for j in range(3):
            kj = np.where(self.cell_side[:, None] == 0, self.k[0][j][self.cells], self.k[1][j][self.cells])
            loc = np.einsum("cab,cb->ca", self.Mloc, kj)
            np.add.at(b, 4 * self.cnode + j, loc)
        for j, fj in enumerate(getattr(self, "f_e", None) or []):       # L += dt * f_e * v dx_e (KNPEMIx_problem.py:614)
            if fj is None:
                continue
            ext = self.cell_side == 1
            loc = np.einsum("cab,cb->ca", self.Mloc[ext], fj[self.cells[ext]])
            np.add.at(b, 4 * self.cnode[ext] + j, dt * loc)
        Iq = self.channel_currents_q()
        Itot = Iq.sum(axis=0)
        phim = self._at_q(self.phi_m)
        al_i, al_e = self._alpha_q(0), self._alpha_q(1)
        lam = self.lamq

        def facet_vec(gq):
            return self.fmeas[:, None] * np.einsum("q,fq,qa->fa", self.qw, gq, lam)

        for j in range(3):
            z = p.z[j]
            gi = (dt * Iq[j] - al_i[j] * C_M * phim) / (F * z)
            ge = (dt * Iq[j] - al_e[j] * C_M * phim) / (F * z)
            np.add.at(b, 4 * self.fnode_i + j, -facet_vec(gi))
            np.add.at(b, 4 * self.fnode_e + j, +facet_vec(ge))
        g = (dt * Itot - C_M * phim) / F
        v = facet_vec(g)
        np.add.at(b, 4 * self.fnode_i + 3, -v)
        np.add.at(b, 4 * self.fnode_e + 3, +v)
        return b

    # ---- HH gating (KNPEMIx_ionic_model.py:605-674)
    def update_t_mod(self, tol=1e-12):
        self.t_mod = float(np.mod(self.t + tol, self.p.T_stim))

    def update_gating(self, mdl: Model):
        p = self.p
        dt_ode = p.dt / mdl.time_steps_ode
        with np.errstate(all="ignore"):
            V = 1000.0 * (self.phi_m - p.phi_rest)
            an = 0.01e3 * (10. - V) / (np.exp((10. - V) / 10.) - 1.)
            bn = 0.125e3 * np.exp(-V / 80.)
            am = 0.1e3 * (25. - V) / (np.exp((25. - V) / 10.) - 1)
            bm = 4.e3 * np.exp(-V / 18.)
            ah = 0.07e3 * np.exp(-V / 20.)
            bh = 1.e3 / (np.exp((30. - V) / 10.) + 1)
            if mdl.use_rush_larsen:
                for (a, b_, name) in ((an, bn, "n"), (am, bm, "m"), (ah, bh, "h")):
                    tau = 1.0 / (a + b_)
                    yinf = a * tau
                    yexp = np.exp(-dt_ode / tau)
                    y = getattr(self, name)
                    for _ in range(mdl.time_steps_ode):
                        y = yinf + (y - yinf) * yexp
                    setattr(self, name, y)
            else:
                for (a, b_, name) in ((an, bn, "n"), (am, bm, "m"), (ah, bh, "h")):
                    y = getattr(self, name)
                    for _ in range(mdl.time_steps_ode):
                        y = y + dt_ode * a * (1 - y) - dt_ode * b_ * y
                    setattr(self, name, y)

    # ---- L2 norms (main.py:70-84)
    def l2_norm(self, nodal, side):
        sel = self.cell_side == side
        u = nodal[self.cells[sel]]
        return math.sqrt(float(np.einsum("ca,cab,cb->", u, self.Mloc[sel], u)))

    def potential_norms(self):
        return self.l2_norm(self.phi[0], 0), self.l2_norm(self.phi[1], 1)

    def potential_block_of_A(self):
        """The potential block of A at the current state -- both sides and their membrane coupling (KNPEMIx_problem.py:633-638) --
        placed at the potential unknowns of an n_dof x n_dof matrix: what the native ``btcc`` preconditioner builds its potential
        hierarchy on (knp_pc_set_coupled_potential) instead of P's uncoupled block."""
        n = self.n_dof
        pidx = np.arange(3, n, 4)
        coo = self.assemble_A().tocsr()[pidx][:, pidx].tocoo()
        M = sp.csr_matrix((coo.data, (pidx[coo.row], pidx[coo.col])), shape=(n, n))
        M.eliminate_zeros()
        M.sort_indices()
        return M

    def node_coords(self):
        """coordinates of the nodes (vertex, side) in node order: input of the nested-dissection ordering"""
        L = self.lay
        X = np.zeros((self.n_dof // 4, self.dim))
        for nodes in (L.node_i, L.node_e):
            v = np.nonzero(nodes >= 0)[0]
            X[nodes[v]] = self.coords[v]
        return X

    def load_state(self, st):
        """Set the time-dependent state from a snapshot {k_i[3], k_e[3], phi_i, phi_e, phi_m, n, m, h, t}: lets a checker redo ONE
        implicit step from the state another implementation was in (single-step parity, independent of the trajectory so far)."""
        for j in range(3):
            self.k[0][j] = np.array(st["k_i"][j], dtype=np.float64)
            self.k[1][j] = np.array(st["k_e"][j], dtype=np.float64)
        self.phi = [np.array(st["phi_i"], dtype=np.float64), np.array(st["phi_e"], dtype=np.float64)]
        self.phi_m = np.array(st["phi_m"], dtype=np.float64)
        for nm in ("n", "m", "h"):
            if st.get(nm) is not None:
                setattr(self, nm, np.array(st[nm], dtype=np.float64))
        self.t = float(st["t"])

    def step_system(self):
        """Advance t, update the gating variables, assemble (A, b) of the next implicit step (KNPEMIx_solver.py:368,395-403,
        104-116) WITHOUT solving it; b is not projected (only step 1 projects, :333)."""
        self.t += self.p.dt
        for mdl in self.models:
            if mdl.kind == "hh":
                self.update_t_mod()
                self.update_gating(mdl)
        return self.assemble_A().tocsr(), self.assemble_b()

    # ---- time loop (KNPEMIx_solver.py:337-468)
    # class defaults of the reference's constants (KNPEMIx_problem.py:941-951)
    REF_DEFAULT_KI, REF_DEFAULT_KE, REF_DEFAULT_PHI_M = (10.0, 130.0, 5.0), (145.0, 3.0, 134.0), -0.070

    def dirichlet_initial_values(self, boundary_vertices, ki=None, ke=None, phi_m=None):
        """Dirichlet data of the reference's non-MMS branch (KNPEMIx_problem.py:135-160): on the exterior boundary
        every field keeps an 'initial' value (concentrations k_init, phi_i = phi_m_init, phi_e = 0).  Reference quirk:
        the BC functions are filled when the problem is CONSTRUCTED, i.e. before set_initial_conditions() copies the
        config's initial_conditions into the constants, so the values are the class defaults (REF_DEFAULT_*)."""
        p = self.p
        ki = self.REF_DEFAULT_KI if ki is None else ki
        ke = self.REF_DEFAULT_KE if ke is None else ke
        phi_m = self.REF_DEFAULT_PHI_M if phi_m is None else phi_m
        dofs, vals = [], []
        for side, nodes, kinit, phi0 in ((0, self.lay.node_i, ki, phi_m), (1, self.lay.node_e, ke, 0.0)):
            nb = nodes[boundary_vertices]
            nb = nb[nb >= 0]
            for j in range(3):
                dofs.append(4 * nb + j); vals.append(np.full(len(nb), kinit[j]))
            dofs.append(4 * nb + 3); vals.append(np.full(len(nb), phi0))
        return np.concatenate(dofs), np.concatenate(vals)

    def run_dirichlet(self, time_steps, bc_dofs, bc_vals, solver="lu", pc=None, rtol=1e-9, max_it=5000):
        """Time loop with Dirichlet rows (no null space, KNPEMIx_solver.py:380,415).
        solver 'lu': sparse LU of the reduced system (DOLFINx zeroes row and column and lifts; row replacement gives the
        same solution).  solver 'gmres': the PETSc-style GMRES(30) on the row-replaced system with the preconditioner
        built by ``pc(P_with_identity_rows)``; the preconditioner acts as the identity on the Dirichlet rows."""
        keep = np.ones(self.n_dof, dtype=bool); keep[bc_dofs] = False
        idx = np.nonzero(keep)[0]
        g = np.zeros(self.n_dof); g[bc_dofs] = bc_vals
        its = []

        def rowrep(A):
            A = A.tocsr().copy()
            for d in bc_dofs:                                  # zero the row, unit diagonal
                A.data[A.indptr[d]:A.indptr[d + 1]] = 0.0
            A = A + sp.csr_matrix((np.ones(len(bc_dofs)), (bc_dofs, bc_dofs)), shape=A.shape)
            A.eliminate_zeros()
            return A.tocsr()
        M = None
        if solver == "gmres":
            M0 = pc(rowrep(self.assemble_P()))

            def M(r):
                z = M0(r)
                z[bc_dofs] = r[bc_dofs]
                return z
            x = self.pack()
            x[bc_dofs] = bc_vals
        for step in range(1, time_steps + 1):
            self.t += self.p.dt
            for mdl in self.models:
                if mdl.kind == "hh":
                    self.update_t_mod()
                    self.update_gating(mdl)
            A = self.assemble_A().tocsr()
            b = self.assemble_b()
            if solver == "gmres":
                Ar = rowrep(A)
                b[bc_dofs] = bc_vals
                x, it, _ = gmres_left(Ar, b, x, M, ns=None, rtol=rtol, max_it=max_it)
                its.append(it)
            else:
                x = g.copy()
                x[idx] = spla.splu(A[idx][:, idx].tocsc()).solve((b - A @ g)[idx])
            self.unpack(x)
        self.dirichlet_iterations = its
        return x

    def run(self, time_steps, solver="lu_gauge", pc=None, rtol=1e-9, max_it=5000, log=None):
        """solver: 'lu_gauge' (sparse LU, l2 gauge of the iterative path; 'lu_gauge_nd': with the nested-dissection ordering),
        'lu_pin' (sparse LU with one potential DoF pinned to 0 - a MUMPS-like gauge),
        'gmres' (PETSc-like left-preconditioned GMRES(30) with pc(P) callback factory)."""
        x = self.pack()
        ns = self.nullspace()
        its = []
        M = None
        if solver == "gmres":
            P = self.assemble_P()
            M = pc(P) if pc is not None else (lambda r: r)
        for step in range(1, time_steps + 1):
            self.t += self.p.dt
            for mdl in self.models:
                if mdl.kind == "hh":
                    self.update_t_mod()
                    self.update_gating(mdl)
            A = self.assemble_A()
            self.current_A = A
            b = self.assemble_b()
            if step == 1:
                assert np.abs(A @ ns).max() <= 1e-10 * np.abs(A).max() * 100, "ns not in null space"
                b = b - ns * (ns @ b)                                  # solver :333
            if solver == "lu_gauge":
                x = solve_lu_gauge(A, b, ns, ns @ x)
            elif solver == "lu_gauge_nd":                                  # same solve, nested-dissection ordering (long 3D runs)
                x = solve_lu_gauge_nd(A, b, ns, ns @ x, self.node_coords())
            elif solver == "lu_pin":
                x = solve_lu_pin(A, b, pin=self.n_dof - 1)
            elif solver == "gmres":
                x, it, _ = gmres_left(A, b, x, M, ns=ns, rtol=rtol, max_it=max_it)
                its.append(it)
            else:
                raise ValueError(solver)
            self.unpack(x)
            if log is not None:
                log(step, self, x)
        return x, its


# --------------------------------------------------------------------------------------
# Linear solvers
# --------------------------------------------------------------------------------------

def solve_lu_gauge(A, b, ns, gauge_value):
    """Solve the singular system A x = b with ns.x = gauge_value through the bordered
    system [[A, ns],[ns^T, 0]] (the gauge PETSc's null-space projection keeps)."""
    n = A.shape[0]
    K = sp.bmat([[A, sp.csr_matrix(ns[:, None])], [sp.csr_matrix(ns[None, :]), None]], format="csc")
    rhs = np.concatenate([b, [gauge_value]])
    sol = spla.splu(K).solve(rhs)
    return sol[:n]


def nested_dissection_order(A, block=4, coords=None, leaf=96):
    """Fill-reducing ordering for the sparse LU of the block system: recursive coordinate bisection of the NODE graph (``block``
    unknowns per node) with vertex separators ordered last.  SuperLU's COLAMD needs 107 M factor entries and 16 s for the 128^2
    square, this ordering 20 M and 1.5 s; it is what makes one direct solve of the 512^2 benchmark case (the check the
    reference's own direct-solver test performs, tests/KNPEMI/electric_potential_norms_direct_solver.py:55-68 with MUMPS) affordable
    on one core.  ``coords`` [n_nodes, dim]; without them the node index is used as a 1-D coordinate (band-like splitting).
    Returns a permutation of the unknowns."""
    n = A.shape[0]
    nn = n // block
    assert nn * block == n
    coo = A.tocoo()
    G = sp.csr_matrix((np.ones(coo.nnz, dtype=np.int8), (coo.row // block, coo.col // block)), shape=(nn, nn))
    G = ((G + G.T) > 0).astype(np.int32).tocsr()
    X = np.arange(nn, dtype=np.float64)[:, None] if coords is None else np.asarray(coords, dtype=np.float64)
    mark = np.zeros(nn, dtype=np.int32)
    pieces = []

    def rec(idx):
        if idx.size <= leaf:
            pieces.append(idx)
            return
        ext = X[idx].max(axis=0) - X[idx].min(axis=0)
        c = X[idx, int(np.argmax(ext))]
        med = np.partition(c, idx.size // 2)[idx.size // 2]
        lo = c < med
        if not lo.any() or lo.all():                  # many equal coordinates: split by position
            lo = np.zeros(idx.size, dtype=bool)
            lo[np.argsort(c, kind="stable")[:idx.size // 2]] = True
        left, right = idx[lo], idx[~lo]
        mark[right] = 1
        touch = (G[left] @ mark) > 0                  # left nodes with a neighbour on the right: the separator
        mark[right] = 0
        rec(left[~touch])
        rec(right)
        pieces.append(left[touch])
    import sys as _sys
    _sys.setrecursionlimit(max(_sys.getrecursionlimit(), 10000))
    rec(np.arange(nn))
    order = np.concatenate(pieces)
    assert order.size == nn
    return (block * order[:, None] + np.arange(block)[None, :]).ravel()


def solve_lu_gauge_nd(A, b, ns, gauge_value, node_coords=None, block=4):
    """``solve_lu_gauge`` with the nested-dissection ordering above (same bordered system, the gauge row/column last)."""
    n = A.shape[0]
    perm = np.concatenate([nested_dissection_order(A, block, node_coords), [n]])
    K = sp.bmat([[A, sp.csr_matrix(ns[:, None])], [sp.csr_matrix(ns[None, :]), None]], format="csr")
    Kp = K[perm][:, perm].tocsc()
    rhs = np.concatenate([b, [gauge_value]])
    lu = spla.splu(Kp, permc_spec="NATURAL", diag_pivot_thresh=0.0)
    sol = np.empty(n + 1)
    sol[perm] = lu.solve(rhs[perm])
    # one step of iterative refinement (static pivoting on the diagonal): keeps the residual at the level of partial pivoting
    r = rhs - K @ sol
    d = np.empty(n + 1)
    d[perm] = lu.solve(r[perm])
    return (sol + d)[:n]


def single_step_check(o, state, x_new, lu=True, blocks=True):
    """Preconditioner-independent check of ONE implicit step of another implementation (the HIP path): the oracle is put into
    the state that implementation was in BEFORE the step (``state``, see OracleKNPEMI.load_state), assembles its own A and b of
    the step, and measures
      * the TRUE residual of the candidate solution ``x_new``: ||b - A x|| / ||b|| overall and per field block of the reference's
        block vector [k_i^1..3, phi_i | k_e^1..3, phi_e] (KNPEMIx_problem.py:34-48), plus the normwise backward error
        ||r|| / (|| |A| |x| || + ||b||) per block (the phi-rows have a tiny right-hand side: only membrane terms);
      * with ``lu``: the difference to the oracle's own sparse direct solve of the same system in the same gauge (what the
        reference's direct-solver test pins, tests/KNPEMI/electric_potential_norms_direct_solver.py:55-68): per field max-norm
        relative differences, the L2 norms of both potentials, phi_m on the membrane.
    Nothing here depends on the candidate's preconditioner, Krylov method or iteration history.  Leaves ``o`` advanced by the
    step (with the direct solution when ``lu``, else with x_new)."""
    import time as _time
    o.load_state(state)
    ns = o.nullspace()
    gauge = float(ns @ o.pack())
    t0 = _time.perf_counter()
    A, b = o.step_system()
    t_asm = _time.perf_counter() - t0
    x = np.asarray(x_new, dtype=np.float64)
    r = b - A @ x
    out = {"assemble_s": t_asm, "n_dof": int(o.n_dof), "rel_residual": float(np.linalg.norm(r) / np.linalg.norm(b)),
           "gauge_drift": float(abs(ns @ x - gauge) / max(abs(gauge), 1e-300))}
    if blocks:
        Aabs = abs(A)
        ax = Aabs @ np.abs(x)
        side = np.zeros(o.n_dof // 4, dtype=np.int8)
        side[o.lay.node_e[o.lay.node_e >= 0]] = 1
        names = ("Na", "K", "Cl", "phi")
        blk = {}
        for sd, sn in ((0, "i"), (1, "e")):
            nodes = np.nonzero(side == sd)[0]
            for f in range(4):
                d = 4 * nodes + f
                nb = float(np.linalg.norm(b[d]))
                blk[f"{names[f]}_{sn}"] = {"rel_to_b": float(np.linalg.norm(r[d]) / nb) if nb > 0 else None,
                                           "backward": float(np.linalg.norm(r[d]) / (np.linalg.norm(ax[d]) + nb))}
        out["blocks"] = blk
        out["max_backward"] = max(v["backward"] for v in blk.values())
    if lu:
        t0 = _time.perf_counter()
        xd = solve_lu_gauge_nd(A, b, ns, gauge, o.node_coords())
        out["lu_s"] = _time.perf_counter() - t0
        out["lu_rel_residual"] = float(np.linalg.norm(b - A @ xd) / np.linalg.norm(b))
        out["lu_field_diff"] = [float(np.abs(x[f::4] - xd[f::4]).max() / np.abs(xd[f::4]).max()) for f in range(4)]
        o.unpack(x)
        ci, ce = o.potential_norms()
        pm_c = o.phi_m.copy()
        o.unpack(xd)
        di, de = o.potential_norms()
        gam = (o.lay.node_i >= 0) & (o.lay.node_e >= 0)
        out.update(phi_i_L2=ci, phi_e_L2=ce, lu_phi_i_L2=di, lu_phi_e_L2=de,
                   rel_err_phi_i_L2=abs(ci - di) / di, rel_err_phi_e_L2=abs(ce - de) / de,
                   abs_err_phi_e_over_phi_i=abs(ce - de) / di,
                   rel_err_phi_m_max=float(np.abs(pm_c[gam] - o.phi_m[gam]).max() / np.abs(o.phi_m[gam]).max()))
    else:
        o.unpack(x)
    return out


def solve_lu_pin(A, b, pin):
    """Solve with one DoF pinned to zero (what a null-pivot direct solver returns)."""
    n = A.shape[0]
    keep = np.ones(n, dtype=bool)
    keep[pin] = False
    idx = np.nonzero(keep)[0]
    Ar = A[idx][:, idx].tocsc()
    x = np.zeros(n)
    x[idx] = spla.splu(Ar).solve(b[idx])
    return x


def gmres_left(A, b, x0, M, ns=None, rtol=1e-9, atol=1e-50, max_it=5000, restart=30):
    """PETSc-style GMRES(restart): left preconditioning, classical Gram-Schmidt without
    refinement, preconditioned-residual convergence test against max(rtol*||M b||, atol)
    (non-zero initial guess), null space removed after every preconditioner
    application (KSP_RemoveNullSpace).  Returns (x, iterations, final residual norm)."""
    def apply_M(r):
        z = M(r)
        if ns is not None:
            z = z - ns * (ns @ z)
        return z

    x = x0.copy()
    n = b.size
    bnorm = np.linalg.norm(apply_M(b))
    ttol = max(rtol * bnorm, atol)
    it = 0
    V = np.zeros((restart + 1, n))
    H = np.zeros((restart + 1, restart))
    while True:
        r = apply_M(b - A @ x)
        beta = np.linalg.norm(r)
        if beta <= ttol or it >= max_it:
            return x, it, beta
        V[0] = r / beta
        g = np.zeros(restart + 1)
        g[0] = beta
        cs = np.zeros(restart)
        sn = np.zeros(restart)
        j_done = 0
        conv = False
        for j in range(restart):
            w = apply_M(A @ V[j])
            h = V[: j + 1] @ w                       # classical Gram-Schmidt
            w = w - V[: j + 1].T @ h
            hn = np.linalg.norm(w)
            H[: j + 1, j] = h
            H[j + 1, j] = hn
            for i in range(j):                       # apply previous rotations
                t = cs[i] * H[i, j] + sn[i] * H[i + 1, j]
                H[i + 1, j] = -sn[i] * H[i, j] + cs[i] * H[i + 1, j]
                H[i, j] = t
            den = math.hypot(H[j, j], H[j + 1, j])
            cs[j] = H[j, j] / den
            sn[j] = H[j + 1, j] / den
            H[j, j] = den
            H[j + 1, j] = 0.0
            g[j + 1] = -sn[j] * g[j]
            g[j] = cs[j] * g[j]
            it += 1
            j_done = j + 1
            res = abs(g[j + 1])
            if hn > 0:
                V[j + 1] = w / hn
            if res <= ttol or it >= max_it:
                conv = True
                break
        y = np.linalg.solve(np.triu(H[:j_done, :j_done]), g[:j_done])
        x = x + V[:j_done].T @ y
        if conv:
            return x, it, res


# --------------------------------------------------------------------------------------
# Preconditioners on P used by both the oracle and (restated independently) the HIP path
# --------------------------------------------------------------------------------------

def pc_vertex_block_jacobi(layout: Layout):
    """Factory: inverse of the diagonal blocks of P gathered per mesh vertex (4x4 on
    ordinary vertices, 8x8 on membrane vertices)."""
    def factory(P):
        n = P.shape[0]
        P = P.tocsr()
        nn = layout.n_nodes
        # group id of each node = vertex
        grp = layout.node_vertex
        starts = np.nonzero(np.r_[True, grp[1:] != grp[:-1]])[0]
        sizes = np.diff(np.r_[starts, nn])
        invs = []
        for s, sz in zip(starts, sizes):
            idx = np.arange(4 * s, 4 * (s + sz))
            blk = P[idx][:, idx].toarray()
            invs.append((idx, np.linalg.inv(blk)))

        def apply(r):
            z = np.empty_like(r)
            for idx, inv in invs:
                z[idx] = inv @ r[idx]
            return z
        return apply
    return factory


def pc_exact_lu():
    """Factory: exact LU of P (stands in for a converged AMG cycle; CPU only)."""
    def factory(P):
        lu = spla.splu(P.tocsc())
        return lu.solve
    return factory


def pc_amg_vcycle(levels, coarse_inv, pre=1, post=1, cheby_degree=2, fused=False):
    """NumPy restatement of the V-cycle the HIP library applies (knp_kernels.hip: amg_vcycle /
    amg_smooth).  ``levels`` = list of objects with .A (csr), .dinv, .lambda_max, .P, .R; the
    hierarchy itself is *data* produced by the host setup and is passed in by the test.
    Smoother: Chebyshev of the given degree on D^-1 A over [0.1, 1.1]*lambda_max."""
    def smooth(lv, b, x, zero):
        lmax, lmin = 1.1 * lv.lambda_max, 0.1 * lv.lambda_max
        theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
        sigma = theta / delta
        rho_old = 1.0 / sigma
        if zero:
            d = lv.dinv * b / theta
            x = d.copy()
        else:
            d = lv.dinv * (b - lv.A @ x) / theta
            x = x + d
        for _ in range(1, cheby_degree):
            rho = 1.0 / (2.0 * sigma - rho_old)
            d = rho * rho_old * d + (2.0 * rho / delta) * (lv.dinv * (b - lv.A @ x))
            x = x + d
            rho_old = rho
        return x

    def cycle(l, b):
        lv = levels[l]
        if l == len(levels) - 1:
            if coarse_inv is not None:
                return coarse_inv @ b
            x = smooth(lv, b, None, True)
            for _ in range(1, pre + post):
                x = smooth(lv, b, x, False)
            return x
        x = None
        for s in range(pre):
            x = smooth(lv, b, x, x is None)
        if x is None:
            x = np.zeros_like(b)
        r = b - lv.A @ x
        xc = cycle(l + 1, lv.R @ r)
        x = x + lv.P @ xc
        for s in range(post):
            x = smooth(lv, b, x, False)
        return x

    if fused:
        return pc_amg_vcycle_fused(levels, coarse_inv)
    return lambda r: cycle(0, r)


def pc_amg_vcycle_fused(levels, coarse_inv):
    """The same V(1,1) / Chebyshev-degree-1 cycle in the order the library's FUSED form evaluates it
    (knp_kernels.hip: amg_cycle_fused): with x0 = c Dinv b,
        level 0 down:  r = b - c Pt b,  Pt = A Dinv           (levels[0].Pt when the test supplies the stored values)
        level l down:  x = c Dinv b ; r = b - A x
        up:            x <- x + c Dinv r + S x_coarse,  S = (I - c Dinv A) Pprol   (levels[l].S)
    When every intermediate level carries the composite operators Rt and U (hierarchies built in round 3 and later) those levels are
    evaluated as the library does: b_{l+1} = Rt b_l, x_l = U [b_l ; x_{l+1}].
    Algebraically identical to pc_amg_vcycle(levels, coarse_inv, 1, 1, 1); the rounding of operators stored in fp32 (Pt, S, Rt, U
    instead of A, P, R) is what differs, and is what the iterate-parity tests need to reproduce."""
    assert coarse_inv is not None
    nl = len(levels)
    c = [1.0 / (0.5 * (1.1 + 0.1) * lv.lambda_max) for lv in levels]
    coarse_fused = nl >= 3 and all(getattr(levels[l], "Rt", None) is not None and getattr(levels[l], "U", None) is not None for l in range(1, nl - 1))

    def apply(b):
        bs, rs, xs = [b], [], []
        lv = levels[0]
        Pt = getattr(lv, "Pt", None)
        if Pt is None:
            Pt = lv.A @ sp.diags(lv.dinv)
        rs.append(b - c[0] * (Pt @ b))
        xs.append(c[0] * lv.dinv * b)
        if coarse_fused:
            # intermediate levels as two plain products (knp_kernels.hip amg_cycle_fused, cgx_hip/amg.py coarse_fused_operators):
            # b_{l+1} = Rt_l b_l on the way down, x_l = U_l [b_l ; x_{l+1}] on the way up -- no level iterate, no level residual
            bs.append(levels[0].R @ rs[0])
            for l in range(1, nl - 1):
                bs.append(levels[l].Rt @ bs[l])
            xc = coarse_inv @ bs[nl - 1]
            for l in range(nl - 2, 0, -1):
                xc = levels[l].U @ np.concatenate([bs[l], xc])
            return xs[0] + c[0] * lv.dinv * rs[0] + lv.S @ xc
        for l in range(nl - 1):
            bc = levels[l].R @ rs[l]
            bs.append(bc)
            if l + 1 == nl - 1:
                xs.append(coarse_inv @ bc)
            else:
                C = levels[l + 1]
                xc = c[l + 1] * C.dinv * bc
                xs.append(xc)
                rs.append(bc - C.A @ xc)
        for l in range(nl - 2, -1, -1):
            lv = levels[l]
            xs[l] = xs[l] + c[l] * lv.dinv * rs[l] + lv.S @ xs[l + 1]
        return xs[0]
    return apply


def pc_block_lower(o, hier_k, hier_p, pre=1, post=1, cheby_degree=2, fused=False):
    """The reference's preconditioner matrix with ``use_block_jacobi=False`` (KNPEMIx_problem.py:720-722): P keeps the (phi,k)
    blocks dt z_j D_j K, which are the (phi,k) blocks of A itself.  Applied the way the library applies it (KNP_PC_AMG_LT): block
    forward substitution  z_k = V_k r_k ;  z_phi = V_phi (r_phi - P_{phi,k} z_k)  with the two hierarchies of the diagonal blocks."""
    Vk = pc_amg_vcycle(hier_k.levels, hier_k.coarse_inv, pre, post, cheby_degree, fused=fused)
    Vp = pc_amg_vcycle(hier_p.levels, hier_p.coarse_inv, pre, post, cheby_degree, fused=fused)
    n = o.n_dof
    pidx = np.arange(3, n, 4)
    kmask = np.ones(n, dtype=bool)
    kmask[pidx] = False

    def apply(r):
        A = o.current_A
        z = Vk(r)
        z[pidx] = 0.0
        zk = np.where(kmask, z, 0.0)
        t = np.zeros_like(r)
        t[pidx] = r[pidx] - (A @ zk)[pidx] + 0.0          # only the (phi,k) blocks of A act on zk's ion entries ...
        # ... but A's phi rows also hold (phi,phi) columns, which zk does not touch (its potential entries are zero)
        w = Vp(t)
        z[pidx] = w[pidx]
        return z
    return apply


def pc_btcc(o, hier_k, hier_p, pre=1, post=1, cheby_degree=2, bc_dofs=None, fused=False):
    """NumPy restatement of the library's block lower-triangular preconditioner (KNP_PC_AMG_BT):
        z_k   = V_k r                                         (V-cycle of the ion-field hierarchy)
        t_phi = r_phi - sum_j z_j r_kj + M (sum_j z_j z_kj)   (== r_phi - A_{phi,k} z_k for exact ion solves,
                 because A_{phi,kj} = z_j (A_{kj,kj} - M); this form does not amplify the V-cycle error)
        z_phi = V_phi t + cc * t_phi,  cc = psi / (sum_j z_j^2 k_j) / M_lumped   (Cahouet-Chabard Schur term)
    The hierarchies are data built by the host setup and passed in by the test."""
    Vk = pc_amg_vcycle(hier_k.levels, hier_k.coarse_inv, pre, post, cheby_degree, fused=fused)
    Vp = pc_amg_vcycle(hier_p.levels, hier_p.coarse_inv, pre, post, cheby_degree, fused=fused)
    n = o.n_dof
    nn = o.lay.n_nodes
    nv1 = o.dim + 1
    # same-side P1 mass matrix on nodes
    Mn = sp.coo_matrix((o.Mloc.ravel(), (o.rowsA.ravel(), o.colsA.ravel())), shape=(nn, nn)).tocsr()
    ML = np.asarray(Mn.sum(axis=1)).ravel()
    pidx = np.arange(3, n, 4)
    zz = np.array(o.p.z)

    def apply(r):
        p = o.p
        s = np.zeros(nn)
        for side, nodes in ((0, o.lay.node_i), (1, o.lay.node_e)):
            v = np.nonzero(nodes >= 0)[0]
            s[nodes[v]] = sum(p.z[j] ** 2 * o.k[side][j][v] for j in range(3))
        cc = p.psi / (s * ML)
        z = Vk(r)
        z[pidx] = 0.0
        zr = sum(zz[j] * r[j::4] for j in range(3))
        zk = sum(zz[j] * z[j::4] for j in range(3))
        t = np.zeros_like(r)
        t[pidx] = r[pidx] - zr + Mn @ zk
        if bc_dofs is not None:                 # pinned potentials: no Schur coupling (library: k_bc_copy)
            bp = bc_dofs[bc_dofs % 4 == 3]
            t[bp] = r[bp]
        w = Vp(t)
        z[pidx] = w[pidx] + cc * t[pidx]
        return z
    return apply


# --------------------------------------------------------------------------------------
# Convenience constructors for the BASELINE configs
# --------------------------------------------------------------------------------------

def make_square(N, models=None, params=None, scale=1e-6):
    coords, cells = unit_square_mesh(N)
    tag = mark_subdomains(coords, cells)
    return OracleKNPEMI(coords, cells, tag, params=params, models=models, mesh_conversion_factor=scale)


def make_cube(N, models=None, params=None, scale=1e-6):
    coords, cells = unit_cube_mesh(N)
    tag = mark_subdomains(coords, cells)
    return OracleKNPEMI(coords, cells, tag, params=params, models=models, mesh_conversion_factor=scale)


CI_MODELS = lambda: [Model("neuronal_ct", (4,)), Model("hh", (4,)), Model("atp", (4,))]

# Known answers held by the reference's own tests (the pins)
PIN_ITERATIVE = (3.510994056704844e-08, 6.369472309249516e-11)   # tests/KNPEMI/electric_potential_norms_iterative_solver.py:58-59
PIN_DIRECT = (2.6337161145147203e-08, 1.5258564901943312e-08)     # tests/KNPEMI/electric_potential_norms_direct_solver.py:55-56


if __name__ == "__main__":
    t0 = time.perf_counter()
    o = make_square(32, models=CI_MODELS())
    o.run(10, solver="lu_gauge")
    ni, ne = o.potential_norms()
    print("n_dof", o.n_dof, "time", time.perf_counter() - t0)
    print("phi_i", ni, "pin", PIN_ITERATIVE[0], "rel", abs(ni - PIN_ITERATIVE[0]) / PIN_ITERATIVE[0])
    print("phi_e", ne, "pin", PIN_ITERATIVE[1], "rel", abs(ne - PIN_ITERATIVE[1]) / PIN_ITERATIVE[1])
