"""Smoothed-aggregation AMG hierarchy for the block-diagonal preconditioner matrix P.

Setup only (runs once, like ``ksp.setUp()`` with ``pc_type hypre`` in the reference,
src/CGx/KNPEMI/KNPEMIx_solver.py:386-389): builds the level operators on the host with
SciPy sparse products and hands them to the HIP library (``knp_amg_set_level``), which
applies the V-cycle on the GPU with its CSR SpMV / Chebyshev kernels.

P couples neither different fields nor the two sides (KNPEMIx_problem.py:717-738), so a scalar
aggregation on P's own strength graph automatically yields one hierarchy per elliptic block
(k_i^1..3, phi_i, k_e^1..3, phi_e) -- the "block-Jacobi" structure the reference feeds to BoomerAMG.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def _row_max(indptr, vals_at_cols, fill):
    """max over each CSR row of vals_at_cols (already gathered by column); empty rows -> fill."""
    n = indptr.size - 1
    out = np.full(n, fill, dtype=vals_at_cols.dtype)
    nonempty = indptr[1:] > indptr[:-1]
    if vals_at_cols.size:
        red = np.maximum.reduceat(vals_at_cols, indptr[:-1][nonempty])
        out[nonempty] = red
    return out


def strength_graph(A: sp.csr_matrix, theta: float) -> sp.csr_matrix:
    """Symmetric SA strength: |a_ij| >= theta*sqrt(|a_ii a_jj|), diagonal excluded."""
    A = A.tocsr()
    d = np.abs(A.diagonal())
    coo = A.tocoo()
    keep = (coo.row != coo.col) & (np.abs(coo.data) >= theta * np.sqrt(d[coo.row] * d[coo.col])) & (coo.data != 0)
    S = sp.csr_matrix((np.ones(int(keep.sum())), (coo.row[keep], coo.col[keep])), shape=A.shape)
    S = ((S + S.T) > 0).astype(np.float64).tocsr()
    S.sort_indices()
    return S


def aggregate(S: sp.csr_matrix, seed: int = 0, distance: int = 2) -> tuple[np.ndarray, int]:
    """Distance-2 maximal-independent-set aggregation (parallel-friendly Vanek scheme); ``distance=1`` takes the roots
    from an MIS of S itself: smaller aggregates (slower coarsening, better interpolation).

    Roots form an MIS of S^2 (Luby rounds with fixed pseudo-random priorities), every root takes
    its strong neighbours; leftovers join the neighbouring aggregate of highest priority; isolated
    nodes become singletons.  Deterministic for a given seed."""
    n = S.shape[0]
    indptr, indices = S.indptr, S.indices
    rng = np.random.default_rng(seed)
    prio = rng.permutation(n).astype(np.float64) + 1.0       # unique priorities in [1, n]
    state = np.zeros(n, dtype=np.int8)                       # 0 undecided, 1 root, 2 removed
    for _ in range(200):
        und = state == 0
        if not und.any():
            break
        w = np.where(und, prio, 0.0)
        m1 = np.maximum(w, _row_max(indptr, w[indices], 0.0))
        m2 = np.maximum(m1, _row_max(indptr, m1[indices], 0.0)) if distance >= 2 else m1
        new_root = und & (w >= m2)
        state[new_root] = 1
        r = new_root.astype(np.float64)
        r1 = np.maximum(r, _row_max(indptr, r[indices], 0.0))
        r2 = np.maximum(r1, _row_max(indptr, r1[indices], 0.0)) if distance >= 2 else r1
        state[(state == 0) & (r2 > 0)] = 2
    roots = np.nonzero(state == 1)[0]
    agg = np.full(n, -1, dtype=np.int64)
    agg[roots] = np.arange(roots.size)
    # phase 1: strong neighbours of a root join it (distance-2 independence => no conflicts)
    rootid = np.where(state == 1, agg, -1).astype(np.float64)
    nb = _row_max(indptr, rootid[indices], -1.0)
    take = (agg < 0) & (nb >= 0)
    agg[take] = nb[take].astype(np.int64)
    # phase 2: remaining nodes join the aggregate of their highest-priority aggregated neighbour
    for _ in range(3):
        left = agg < 0
        if not left.any():
            break
        key = np.where(agg >= 0, prio, -1.0)
        best = _row_max(indptr, key[indices], -1.0)
        # find which neighbour achieved the max: priorities are unique -> map priority -> node
        node_of_prio = np.empty(n + 2, dtype=np.int64)
        node_of_prio[prio.astype(np.int64)] = np.arange(n)
        ok = left & (best > 0)
        agg[ok] = agg[node_of_prio[best[ok].astype(np.int64)]]
    left = np.nonzero(agg < 0)[0]
    nagg = roots.size
    if left.size:
        agg[left] = nagg + np.arange(left.size)
        nagg += left.size
    return agg, int(nagg)


import os as _os
_LAMBDA_PAD = float(_os.environ.get("KNP_AMG_LAMBDA_PAD", "1.05"))     # safety factor on the power-iteration estimate (developer knob)


def estimate_lambda_max(A: sp.csr_matrix, dinv: np.ndarray, iters: int = 20, seed: int = 1) -> float:
    """Power iteration on D^{-1} A (largest magnitude eigenvalue), padded by 5 %."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(A.shape[0])
    x /= np.linalg.norm(x)
    lam = 1.0
    for _ in range(iters):
        y = dinv * (A @ x)
        lam = float(np.linalg.norm(y))
        if lam == 0.0:
            return 1.0
        x = y / lam
    return _LAMBDA_PAD * lam


def _dist(agg_distance, level):
    """aggregation distance of a level: an int for all levels or a sequence per level (last entry repeated)"""
    if isinstance(agg_distance, (list, tuple)):
        return int(agg_distance[min(level, len(agg_distance) - 1)])
    return int(agg_distance)


class Level:
    """One level: operator A, inverse diagonal, lambda_max(D^-1 A), prolongator P, restrictor R = P^T and
    S = (I - c2 D^-1 A) P with c2 = 1 / (0.6 lambda_max): prolongation followed by the post-smoothing step of the
    V(1,1) / Chebyshev-degree-1 cycle, as ONE operator (the library's fused cycle applies it in one gather)."""
    __slots__ = ("A", "dinv", "lambda_max", "P", "R", "S", "Pt", "Rt", "U")

    def __init__(self, A, dinv, lambda_max, P=None, R=None, S=None, Rt=None, U=None):
        self.A, self.dinv, self.lambda_max, self.P, self.R, self.S = A, dinv, lambda_max, P, R, S
        self.Pt = None      # A Dinv as stored by the library on level 0 (filled by fp32_stored for checkers)
        # levels 1 .. n-2 of the fused cycle as two plain sparse products (coarse_fused_operators)
        self.Rt, self.U = Rt, U


def cheby_first_coefficient(lambda_max: float) -> float:
    """1/theta of the smoothing interval [0.1, 1.1] * lambda_max: the coefficient of a degree-1 Chebyshev (= damped Jacobi) step"""
    return 1.0 / (0.5 * (1.1 + 0.1) * lambda_max)


def post_smoothed_prolongator(A, dinv, lambda_max, Pm, AP=None):
    AP = (A @ Pm) if AP is None else AP
    S = (Pm - sp.diags(cheby_first_coefficient(lambda_max) * dinv) @ AP).tocsr()
    S.sort_indices()
    return S


def coarse_fused_operators(A, dinv, lambda_max, R, S):
    """The fused V(1,1) cycle on an INTERMEDIATE level l (neither the finest nor the coarsest), written without the level's
    iterate and residual: with c = 1/theta, x0 = c Dinv b and r = b - A x0 = (I - c A Dinv) b,
        down:  b_{l+1} = R r                           = Rt b,            Rt = R (I - c A Dinv)
        up:    x = x0 + c Dinv r + S x_{l+1}           = U [b ; x_{l+1}],  U = [ c Dinv (2 I - c A Dinv) | S ]
    i.e. ONE sparse product per leg instead of restriction + residual (two kernels at the launch floor) and a gather with three
    extra vectors: on these levels every kernel is latency bound, so the extra entries of Rt and U cost nothing measurable."""
    c = cheby_first_coefficient(lambda_max)
    n = A.shape[0]
    Dinv = sp.diags(dinv)
    AD = (A @ Dinv).tocsr()
    Rt = (R - c * (R @ AD)).tocsr()
    Rt.sort_indices()
    W = (2.0 * c * Dinv - (c * c) * (Dinv @ AD)).tocsr()
    U = sp.hstack([W, S], format="csr")
    U.sort_indices()
    return Rt, U


class Hierarchy:
    def __init__(self, levels, coarse_inv):
        self.levels = levels
        self.coarse_inv = coarse_inv
        self.node_fields = 0

    def describe(self):
        rows = [lv.A.shape[0] for lv in self.levels]
        nnz = [lv.A.nnz for lv in self.levels]
        return {"rows": rows, "nnz": nnz,
                "operator_complexity": float(sum(nnz)) / max(nnz[0], 1),
                "grid_complexity": float(sum(rows)) / max(rows[0], 1)}


def restrict_to_fields(P: sp.csr_matrix, fields, block: int = 4) -> sp.csr_matrix:
    """P with the rows and columns of all other fields zeroed (same size): the level-0 operator of a
    hierarchy that only acts on one field class (ions: (0,1,2); potential: (3,))."""
    P = sp.csr_matrix(P)
    n = P.shape[0]
    # direct filtering of the CSR arrays (two sparse products with a 0/1 diagonal cost 1.9 s on the 10^7-unknown cube, this 0.3 s)
    fmask = np.zeros(block, dtype=bool)
    fmask[list(fields)] = True
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(P.indptr))
    keep = fmask[rows % block] & fmask[P.indices % block] & (P.data != 0.0)
    cnt = np.bincount(rows[keep], minlength=n)
    indptr = np.concatenate([[0], np.cumsum(cnt)]).astype(P.indptr.dtype if cnt.sum() < 2 ** 31 else np.int64)
    out = sp.csr_matrix((P.data[keep], P.indices[keep], indptr), shape=P.shape)
    out.has_sorted_indices = P.has_sorted_indices
    return out


def _replicate_pattern(Sn: sp.csr_matrix, stride: int, fields, n_dof: int) -> sp.csr_matrix:
    """node-level pattern Sn copied onto every field's unknowns: entry (stride*r + f, stride*c + f) for every (r, c) of Sn, f in fields"""
    coo = Sn.tocoo()
    rows = np.concatenate([stride * coo.row + f for f in fields])
    cols = np.concatenate([stride * coo.col + f for f in fields])
    out = sp.csr_matrix((np.ones(rows.size), (rows, cols)), shape=(n_dof, n_dof))
    out.sort_indices()
    return out


import os as _os
_THETA_DECAY = float(_os.environ.get("KNP_AMG_THETA_DECAY", "0.25"))   # strength threshold of level l = theta * decay^l (developer knob)
DENSE_LIMIT = 6000      # largest coarsest level that gets a dense pseudo-inverse
ELIMINATION_LEVEL_MAX = 200000   # levels up to this many coupled unknowns are searched for an independent set to eliminate exactly


def _decoupled_rows(A: sp.csr_matrix, diag: np.ndarray) -> np.ndarray:
    """active rows without any off-diagonal entry: 1x1 blocks of the level operator"""
    coo = A.tocoo()
    off = (coo.row != coo.col) & (coo.data != 0.0)
    cnt = np.bincount(coo.row[off], minlength=A.shape[0])
    return (diag != 0.0) & (cnt == 0)


def low_degree_independent_set(A: sp.csr_matrix, active: np.ndarray, seed: int = 0) -> np.ndarray:
    """Maximal independent set of the graph of A restricted to the ``active`` rows, built greedily from the rows of LOWEST degree
    (Luby rounds with priority = -degree, ties broken by a fixed pseudo-random permutation: deterministic for a given seed).  In
    the potential block of a tissue mesh, once every biological cell has collapsed into ONE unknown, those unknowns couple only
    to extracellular aggregates, never to each other: they are the low-degree rows and form the independent set by themselves."""
    n = A.shape[0]
    coo = A.tocoo()
    m = (coo.row != coo.col) & (coo.data != 0.0) & active[coo.row] & active[coo.col]
    G = sp.csr_matrix((np.ones(int(m.sum())), (coo.row[m], coo.col[m])), shape=(n, n))
    G = ((G + G.T) > 0).astype(np.float64).tocsr()
    deg = np.diff(G.indptr).astype(np.float64)
    perm = np.random.default_rng(seed).permutation(n).astype(np.float64)
    prio = (deg.max() + 1.0 - deg) * (n + 1.0) + perm + 1.0          # larger = earlier; unique
    state = np.where(active, 0, 2).astype(np.int8)                   # 0 undecided, 1 in the set, 2 out
    for _ in range(200):
        und = state == 0
        if not und.any():
            break
        w = np.where(und, prio, 0.0)
        nbmax = _row_max(G.indptr, w[G.indices], 0.0)
        new = und & (w > nbmax)
        state[new] = 1
        hit = _row_max(G.indptr, new.astype(np.float64)[G.indices], 0.0) > 0
        state[(state == 0) & hit] = 2
    return state == 1


def elimination_level(A: sp.csr_matrix, diag: np.ndarray, F: np.ndarray, C: np.ndarray, lam: float):
    """Exact elimination of an independent set F as one level of the cycle: prolongator [-D_F^-1 A_FC ; I_C] (the ideal
    interpolation: the Galerkin operator is the Schur complement on C) and a smoother that solves the F rows exactly and leaves
    the C rows alone (inverse diagonal 1/(c d) on F so that the damped-Jacobi step c Dinv r is exact there, 0 on C).  With these the
    V(1,1) step on this level is a direct block factorisation, whatever the degree of the eliminated unknowns.
    Returns (dinv, prolongator)."""
    n = A.shape[0]
    c = cheby_first_coefficient(lam)
    dinv = np.zeros(n)
    dinv[F] = 1.0 / (c * diag[F])
    iC = np.nonzero(C)[0]
    iF = np.nonzero(F)[0]
    colmap = np.full(n, -1, dtype=np.int64)
    colmap[iC] = np.arange(iC.size)
    AFC = A[iF][:, iC].tocoo()
    rows = np.concatenate([iF[AFC.row], iC])
    cols = np.concatenate([AFC.col, np.arange(iC.size)])
    vals = np.concatenate([-AFC.data / diag[iF[AFC.row]], np.ones(iC.size)])
    Pm = sp.csr_matrix((vals, (rows, cols)), shape=(n, iC.size))
    Pm.sort_indices()
    return dinv, Pm


def try_elimination_level(A, diag, dinv, active, iso, lam, coarse_size, n_levels_so_far):
    """The elimination level of ``build_hierarchy`` (shared by the host and the device builder, which hands its small levels to
    SciPy for this): (Level, Galerkin operator) when the coupled unknowns of this level contain an independent set whose
    elimination leaves at most ``coarse_size`` unknowns, else None."""
    core = active & ~iso
    F = low_degree_independent_set(A, core, seed=n_levels_so_far)
    C = core & ~F
    nC, nF = int(C.sum()), int(F.sum())
    # worth it only when the set is most of the level (a mesh-like graph leaves more than half of its unknowns outside any independent
    # set) and the remainder is small enough for the dense inverse
    if not (0 < nC <= max(coarse_size, DENSE_LIMIT // 2) and nF > 0 and nC <= 0.3 * (nC + nF)):
        return None
    dinv_e, Pm = elimination_level(A, diag, F, C, lam)
    dinv_e = np.where(iso, dinv / cheby_first_coefficient(lam), dinv_e)      # decoupled unknowns: solved by the smoother as well
    R = Pm.T.tocsr()
    R.sort_indices()
    AP = (A @ Pm).tocsr()
    Ac = (R @ AP).tocsr()
    Ac.sort_indices()
    S_ = post_smoothed_prolongator(A, dinv_e, lam, Pm, AP)
    Rt_, U_ = coarse_fused_operators(A, dinv_e, lam, R, S_) if n_levels_so_far else (None, None)
    return Level(A, dinv_e, lam, Pm, R, S_, Rt_, U_), Ac


def build_hierarchy(P: sp.csr_matrix, theta: float = 0.08, max_levels: int = 12, coarse_size: int = 2500,
                    smooth_prolongator: bool = True, agg_distance=2, node_fields=None, split_decoupled: bool = True,
                    smoother_degree: int = 1, eliminate_independent: bool = True) -> Hierarchy:
    """Rows with a zero diagonal are inactive: they get no aggregate (zero rows in the prolongator, zero inverse
    diagonal in the smoother), so a field-restricted P yields a hierarchy of that field class only.

    ``node_fields = (stride, fields)`` (several fields per node, unknown = stride*node + field, the fields decoupled and with the
    same graph -- the three ion blocks of P): the aggregation is done ONCE, on the node graph of the first field, and every field
    uses the same aggregates and the same strength pattern.  Prolongators, restrictors, S and all coarse operators then have
    the same sparsity pattern for every field (coarse unknown = nf*aggregate + field index), which lets the library store one
    column index per node entry with nf values behind it.  The hierarchy is still an ordinary list of scalar CSR levels.

    ``split_decoupled``: unknowns whose row of the level operator has no off-diagonal entry (after a few levels every
    intracellular cell of a tissue mesh is ONE aggregate per field, decoupled from everything else in P; Dirichlet rows) are not
    carried to coarser levels: they get no aggregate and their inverse diagonal is divided by the Chebyshev coefficient of
    the level, which makes the degree-1 smoothing step solve them exactly (x = c (Dinv / c) b = b / d, residual zero).  When
    only such unknowns keep a level above ``DENSE_LIMIT`` the others are injected into a last level of their own, so that the
    coarsest operator is small enough for the dense inverse -- without this a mesh with tens of thousands of cells ends on a
    level of that many 1x1 blocks plus the extracellular part, smoothed but never solved.

    ``eliminate_independent``: scalar hierarchies only; on a level whose coupled unknowns contain an independent set F (no F-F
    coupling) such that at most ``coarse_size`` unknowns remain, F is eliminated exactly (``elimination_level``) instead of being
    aggregated: the coupled potential block of a tissue mesh arrives there as "one unknown per biological cell + the extracellular
    aggregates", which plain aggregation cannot coarsen (16 505 -> 13 881 -> 11 017 unknowns on the 97^3 surrogate) and which is
    far too large for a dense inverse, while its Schur complement on the 2 681 extracellular aggregates is small.  Same
    split-off rule as for decoupled unknowns, with a coupling: the smoother is exact on F (degree-1 smoother only).

    ``smoother_degree`` = the Chebyshev degree the cycle will run with: the split relies on every smoothing step being the damped
    Jacobi step ``x += c Dinv (b - A x)`` (degree 1: any number of pre/post sweeps then leaves ``b/d`` in place).  The momentum term of
    a degree >= 2 polynomial overshoots a zero residual, so for those the split is switched off (the unknowns are carried as
    singletons, as before the split existed)."""
    split_decoupled = bool(split_decoupled) and int(smoother_degree) == 1
    eliminate_independent = bool(eliminate_independent) and int(smoother_degree) == 1
    A = sp.csr_matrix(P, dtype=np.float64)
    A.sort_indices()
    levels = []
    after_elimination = False
    sync = node_fields is not None and len(node_fields[1]) > 1
    stride, fields = (int(node_fields[0]), tuple(int(f) for f in node_fields[1])) if sync else (1, (0,))
    while True:
        diag = A.diagonal()
        dinv = np.where(diag != 0.0, 1.0 / np.where(diag != 0.0, diag, 1.0), 0.0)
        lam = estimate_lambda_max(A, dinv)
        n = A.shape[0]
        active = diag != 0.0
        iso = _decoupled_rows(A, diag) if split_decoupled else np.zeros(n, dtype=bool)
        if sync and iso.any():       # a node is split off only if all of its fields are decoupled
            iso_n = np.logical_and.reduce([iso[f::stride] for f in fields])
            iso = np.zeros(n, dtype=bool)
            for f in fields:
                iso[f::stride] = iso_n
        n_core = int((active & ~iso).sum())
        n_all = int(active.sum())
        if n_all <= coarse_size or n_core == 0 or len(levels) >= max_levels - 1 or after_elimination:
            levels.append(Level(A, dinv, lam))      # (the remainder of an elimination level is sized for the dense inverse: last level)
            break
        # only decoupled unknowns keep this level too large for the dense inverse: the coupled ones move to a level of their own
        inject = n_core <= coarse_size and n_all > DENSE_LIMIT
        # a large INDEPENDENT set whose elimination leaves at most a dense-solvable remainder (collapsed cells coupled to the
        # extracellular aggregates through the membrane: the coupled potential block of a tissue mesh): eliminate it exactly
        if not inject and not sync and eliminate_independent and n_core <= ELIMINATION_LEVEL_MAX:
            el = try_elimination_level(A, diag, dinv, active, iso, lam, coarse_size, len(levels))
            if el is not None:
                levels.append(el[0])
                A = el[1]
                after_elimination = True
                continue
        if iso.any():
            dinv = np.where(iso, dinv / cheby_first_coefficient(lam), dinv)
            active = active & ~iso
        # strength threshold decays with the level (Galerkin operators of smoothed aggregation get denser and
        # their entries more uniform; a fixed threshold stalls the coarsening in 3D)
        if sync:
            # node graph of the first field; aggregates of nodes; every field follows them
            nf = len(fields)
            nn = n // stride
            A0 = A[fields[0]::stride][:, fields[0]::stride].tocsr()
            Sn = strength_graph(A0, theta * _THETA_DECAY ** len(levels))
            if not levels:
                # the other fields must be able to follow the first one's aggregates: a node that has strong neighbours there
                # but none in field f (a row of f dominated by its diagonal, e.g. the membrane mass of the potential in the
                # non-dimensional MMS setting) is smoothed well by Jacobi alone and spoils the coarse operators when it is
                # aggregated with its neighbours; so does a field whose diagonal is not positive where the first one's is (the
                # potential block of P in the MMS setting is indefinite: minus sign on its membrane mass)
                deg0 = np.diff(Sn.indptr)
                for f in fields[1:]:
                    Sf = strength_graph(A[f::stride][:, f::stride].tocsr(), theta)
                    if np.any((np.diff(Sf.indptr) == 0) & (deg0 > 0)) or np.any((diag[f::stride] <= 0.0) & (diag[fields[0]::stride] > 0.0)):
                        return build_hierarchy(P, theta, max_levels, coarse_size, smooth_prolongator, agg_distance, None, split_decoupled,
                                               smoother_degree, eliminate_independent)   # unsynchronised
            act_n = active[fields[0]::stride]
            ian = np.nonzero(act_n)[0]
            if inject:
                agg_n, nagg_n = np.arange(ian.size), int(ian.size)
            elif act_n.all():
                agg_n, nagg_n = aggregate(Sn, seed=len(levels), distance=_dist(agg_distance, len(levels)))
            else:
                agg_n, nagg_n = aggregate(Sn[ian][:, ian].tocsr(), seed=len(levels), distance=_dist(agg_distance, len(levels)))
            S = _replicate_pattern(Sn, stride, fields, n)
            rows_t = np.concatenate([stride * ian + f for f in fields])
            agg = np.concatenate([nf * agg_n + k for k in range(nf)])
            nagg = nf * nagg_n
        else:
            S = strength_graph(A, theta * _THETA_DECAY ** len(levels))
            if inject:
                rows_t = np.nonzero(active)[0]
                agg, nagg = np.arange(rows_t.size), int(rows_t.size)
            elif active.all():
                agg, nagg = aggregate(S, seed=len(levels), distance=_dist(agg_distance, len(levels)))
                rows_t = np.arange(n)
            else:
                ia = np.nonzero(active)[0]
                agg, nagg = aggregate(S[ia][:, ia].tocsr(), seed=len(levels), distance=_dist(agg_distance, len(levels)))
                rows_t = ia
        if nagg >= 0.9 * rows_t.size and not inject:          # coarsening stalled
            levels.append(Level(A, dinv, lam))
            break
        T = sp.csr_matrix((np.ones(rows_t.size), (rows_t, agg)), shape=(n, nagg))
        if smooth_prolongator and not inject:
            # filtered matrix: weak off-diagonals lumped onto the diagonal
            Sp = S + sp.identity(n, format="csr")
            AF = A.multiply(Sp).tocsr()
            lump = np.asarray(A.sum(axis=1)).ravel() - np.asarray(AF.sum(axis=1)).ravel()
            AF = AF + sp.diags(lump)
            dF = AF.diagonal()
            dFinv = np.where(dF != 0.0, 1.0 / np.where(dF != 0.0, dF, 1.0), 0.0)
            lamF = estimate_lambda_max(AF.tocsr(), dFinv, iters=15)
            omega = 4.0 / (3.0 * lamF)
            Pm = (T - sp.diags(omega * dFinv) @ (AF @ T)).tocsr()
        else:
            Pm = T
        Pm.sort_indices()
        R = Pm.T.tocsr()
        R.sort_indices()
        AP = (A @ Pm).tocsr()
        Ac = (R @ AP).tocsr()
        Ac.sort_indices()
        S_ = post_smoothed_prolongator(A, dinv, lam, Pm, AP)
        Rt_, U_ = coarse_fused_operators(A, dinv, lam, R, S_) if levels else (None, None)      # intermediate levels only
        levels.append(Level(A, dinv, lam, Pm, R, S_, Rt_, U_))
        A = Ac
        if sync:
            stride, fields = len(fields), tuple(range(len(fields)))      # coarse unknowns: nf * aggregate + field index
    coarse_inv = dense_pseudo_inverse(levels[-1].A) if levels[-1].A.shape[0] <= DENSE_LIMIT else None
    h = Hierarchy(levels, coarse_inv)
    h.node_fields = len(node_fields[1]) if sync else 0                   # > 0: same pattern for every field on every level
    return h


def cpu_share() -> int:
    """CPUs this process may really use: the cgroup quota where there is one (a GPU box shows all logical CPUs of its host but grants a
    job a fraction), else the affinity mask -- the Python-side twin of the library's ``knp_host_threads``"""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max" and float(per) > 0:
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            if q > 0 and per > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    try:        # one process per GPU: the ranks of a node share that budget
        k = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
        if k > 1:
            n = max(1, n // k)
    except ValueError:
        pass
    return max(1, min(n, 32))


def dense_pseudo_inverse(A: sp.spmatrix) -> np.ndarray:
    """Pseudo-inverse of the coarsest operator (the potential blocks are singular up to the membrane term).  The
    Galerkin operators of the symmetric blocks of P are symmetric: the eigenvalue route is 3-4x cheaper than the SVD,
    which used to be half of the whole host setup; operators with Dirichlet (identity) rows take the general route."""
    D = A.toarray()
    sym = np.abs(D - D.T).max() <= 1e-12 * max(np.abs(D).max(), 1e-300)
    try:        # LAPACK on the CPU share of the process, not on every logical CPU of the host
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=cpu_share()):
            return np.linalg.pinv(D, rcond=1e-13, hermitian=bool(sym))
    except ImportError:
        return np.linalg.pinv(D, rcond=1e-13, hermitian=bool(sym))


def upload(lib, ctx, check, hier: Hierarchy, pre: int = 1, post: int = 1, cheby_degree: int = 2, index: int = 0, level0_native: bool = False):
    """Hand the hierarchy to libknpemi_hip (host arrays are copied by the library).  ``level0_native``: level 0 will run on the
    library's own pair-major P (``knp_amg_use_native_level0`` modes 1-3), so its CSR -- the largest array of the hierarchy, 1.4 GB on
    the 10^7-unknown cube -- is not shipped: an empty pattern takes its place (inverse diagonal, eigenvalue bound and the transfer
    operators are what the cycle needs of that level)."""
    import ctypes as C
    i32p = C.POINTER(C.c_int32)
    f64p = C.POINTER(C.c_double)

    def ip(a):
        return a.ctypes.data_as(i32p)

    def fp(a):
        return a.ctypes.data_as(f64p)

    nl = len(hier.levels)
    check(lib.knp_amg_reset(ctx, index, nl, pre, post, cheby_degree))
    if getattr(hier, "node_fields", 0) in (3, 4):      # node-synchronised: the library keeps node-blocked copies (rows sorted by column)
        check(lib.knp_amg_set_node_fields(ctx, index, int(hier.node_fields)))
        for lv in hier.levels:
            for M in (lv.A, lv.R, getattr(lv, "S", None)):
                if M is not None and not M.has_sorted_indices:
                    M.sort_indices()
    keep = []
    for l, lv in enumerate(hier.levels):
        A = lv.A
        if l == 0 and level0_native and len(hier.levels) > 1:
            rp = np.zeros(A.shape[0] + 1, dtype=np.int32)
            ci = np.zeros(1, dtype=np.int32)
            va = np.zeros(1, dtype=np.float64)
        else:
            rp = np.ascontiguousarray(A.indptr, dtype=np.int32)
            ci = np.ascontiguousarray(A.indices, dtype=np.int32)
            va = np.ascontiguousarray(A.data, dtype=np.float64)
        dinv = np.ascontiguousarray(lv.dinv, dtype=np.float64)
        keep += [rp, ci, va, dinv]
        if lv.P is not None:
            Prp = np.ascontiguousarray(lv.P.indptr, dtype=np.int32)
            Pci = np.ascontiguousarray(lv.P.indices, dtype=np.int32)
            Pv = np.ascontiguousarray(lv.P.data, dtype=np.float64)
            Rrp = np.ascontiguousarray(lv.R.indptr, dtype=np.int32)
            Rci = np.ascontiguousarray(lv.R.indices, dtype=np.int32)
            Rv = np.ascontiguousarray(lv.R.data, dtype=np.float64)
            keep += [Prp, Pci, Pv, Rrp, Rci, Rv]
            check(lib.knp_amg_set_level(ctx, index, l, A.shape[0], A.shape[0], ip(rp), ip(ci), fp(va), fp(dinv),
                                        float(lv.lambda_max), lv.P.shape[1], ip(Prp), ip(Pci), fp(Pv),
                                        ip(Rrp), ip(Rci), fp(Rv)))
            S = getattr(lv, "S", None)
            if S is not None:
                Srp = np.ascontiguousarray(S.indptr, dtype=np.int32)
                Sci = np.ascontiguousarray(S.indices, dtype=np.int32)
                Sv = np.ascontiguousarray(S.data, dtype=np.float64)
                keep += [Srp, Sci, Sv]
                check(lib.knp_amg_set_level_smoothed(ctx, index, l, S.shape[0], ip(Srp), ip(Sci), fp(Sv)))
            Rt, U = getattr(lv, "Rt", None), getattr(lv, "U", None)
            if Rt is not None and U is not None and l >= 1:
                arrs = []
                for M in (Rt, U):
                    if not M.has_sorted_indices:
                        M.sort_indices()
                    arrs += [np.ascontiguousarray(M.indptr, dtype=np.int32), np.ascontiguousarray(M.indices, dtype=np.int32),
                             np.ascontiguousarray(M.data, dtype=np.float64)]
                keep += arrs
                check(lib.knp_amg_set_level_coarse_fused(ctx, index, l, Rt.shape[0], ip(arrs[0]), ip(arrs[1]), fp(arrs[2]),
                                                         U.shape[0], ip(arrs[3]), ip(arrs[4]), fp(arrs[5])))
        else:
            check(lib.knp_amg_set_level(ctx, index, l, A.shape[0], A.shape[0], ip(rp), ip(ci), fp(va), fp(dinv),
                                        float(lv.lambda_max), 0, None, None, None, None, None, None))
    if hier.coarse_inv is not None:
        ci_ = np.ascontiguousarray(hier.coarse_inv, dtype=np.float64)
        check(lib.knp_amg_set_coarse(ctx, index, ci_.shape[0], fp(ci_)))


def fp32_stored(h: Hierarchy, coarse: bool = False, level0_uploaded: bool = False) -> Hierarchy:
    """The hierarchy as the library holds it with ``amg_fp32`` (default): level and transfer operator VALUES rounded to
    fp32 (diagonals, vectors and arithmetic stay fp64).  ``coarse``: also the dense coarse inverse -- what the library does for
    the ion-field hierarchy of the block-triangular preconditioner when its cycle runs fused (the potential hierarchy's and
    the all-field hierarchy's coarse inverses stay fp64).  ``level0_uploaded``: the hierarchy's level 0 runs on the uploaded operator
    (coupled potential block of ``btcc``).  For checkers that restate the V-cycle."""
    import copy

    def rnd(M):
        if M is None:
            return None
        M = M.copy()
        M.data = M.data.astype(np.float32).astype(np.float64)
        return M
    out = copy.copy(h)
    out.levels = []
    for lv in h.levels:
        l2 = copy.copy(lv)
        l2.A, l2.P, l2.R = rnd(lv.A), rnd(lv.P), rnd(lv.R)
        l2.S = rnd(getattr(lv, "S", None))
        l2.Rt, l2.U = rnd(getattr(lv, "Rt", None)), rnd(getattr(lv, "U", None))
        out.levels.append(l2)
    if coarse and h.coarse_inv is not None:
        out.coarse_inv = h.coarse_inv.astype(np.float32).astype(np.float64)
    if out.levels:      # level 0 of the fused cycle applies Pt = A Dinv, computed in fp64 from the fp64 P and rounded once
        l0 = h.levels[0]
        out.levels[0].Pt = rnd((l0.A @ sp.diags(l0.dinv)).tocsr())
        if level0_uploaded:
            # potential hierarchy on its uploaded (coupled) level-0 operator (knp_amg_use_native_level0 mode 4): the library scales the
            # fp32-stored operator, c A32 Dinv, and rounds that to fp32; the fused cycle restated by the oracle multiplies by c again
            c = cheby_first_coefficient(l0.lambda_max)
            Pt = rnd((c * (rnd(l0.A) @ sp.diags(l0.dinv))).tocsr())
            Pt.data /= c
            out.levels[0].Pt = Pt
    return out
