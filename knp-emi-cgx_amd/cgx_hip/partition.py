"""Multilevel k-way partition of the mesh's nodal graph with vertex and edge weights.

The reference inherits its domain decomposition from DOLFINx's graph partitioner (SCOTCH / ParMETIS / KaHIP on the dual graph,
``xdmf.read_mesh(ghost_mode=shared_facet)``, src/CGx/utils/mixed_dim_problem.py:21,649,666).  None of those libraries is available
here; this module is the native counterpart for meshes that are READ (reconstructions), where coordinate bisection is blind to the
two things that matter:

  * membrane (Gamma) vertices carry two nodes -- twice the unknowns, plus the facet quadrature: vertex weight = unknowns per vertex;
  * an edge cut through a biological cell is worse than one through extracellular space: each cut intracellular component costs a
    deflation mode / a coarse coupling (``parallel.cut_component_modes``) -- intracellular and membrane edges get a larger weight.

The owner-computes layout partitions VERTICES (every rank then takes the cells that touch its vertices as ghost layer), so the
graph is the nodal graph (METIS_PartMeshNodal's view), not the dual graph.

Scheme (the classical one: Karypis & Kumar): heavy-edge matching coarsening (a vectorised handshake variant: every vertex proposes
to its heaviest unmatched neighbour, mutual proposals match) until a few thousand vertices remain; initial partition of the coarsest
graph by weighted recursive coordinate bisection of the coarse vertices' centroids; projection back with greedy boundary refinement
at every level (move a boundary vertex to the neighbouring part it is most connected to when that lowers the cut and keeps the
balance, independent sets of moves per sweep).  Deterministic (fixed seeds).  NumPy / SciPy on the host: it runs once, at setup.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def nodal_graph(n_v: int, cells: np.ndarray, edge_weight_of_cell: np.ndarray | None = None) -> sp.csr_matrix:
    """Symmetric vertex graph of a simplicial mesh; the weight of an edge is the sum of ``edge_weight_of_cell`` (default 1) over
    the cells that contain it (so interior edges of heavy regions are heavy)."""
    cells = np.asarray(cells)
    nv1 = cells.shape[1]
    w = np.ones(cells.shape[0]) if edge_weight_of_cell is None else np.asarray(edge_weight_of_cell, dtype=np.float64)
    rows, cols, vals = [], [], []
    for a in range(nv1):
        for b in range(nv1):
            if a != b:
                rows.append(cells[:, a]); cols.append(cells[:, b]); vals.append(w)
    G = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n_v, n_v)).tocsr()
    G.sum_duplicates()
    return G


def _row_argmax(G: sp.csr_matrix, score: np.ndarray):
    """for every row the column with the largest ``score`` entry (one score per stored entry), -1 for rows without a positive one"""
    n = G.shape[0]
    indptr, indices = G.indptr, G.indices
    best = np.full(n, -1, dtype=np.int64)
    nonempty = indptr[1:] > indptr[:-1]
    if not score.size:
        return best
    rmax = np.zeros(n)
    rmax[nonempty] = np.maximum.reduceat(score, indptr[:-1][nonempty])
    rows = np.repeat(np.arange(n), np.diff(indptr))
    hit = (score == rmax[rows]) & (score > 0)
    # first hit per row
    idx = np.nonzero(hit)[0]
    r = rows[idx]
    first = np.ones(idx.size, dtype=bool)
    first[1:] = r[1:] != r[:-1]
    best[r[first]] = indices[idx[first]]
    return best


def heavy_edge_matching(G: sp.csr_matrix, vwgt: np.ndarray, max_vwgt: float, seed: int = 0, rounds: int = 4) -> np.ndarray:
    """match[v] = partner (or v itself): handshake rounds on the heaviest admissible edge (combined weight <= max_vwgt)"""
    n = G.shape[0]
    rng = np.random.default_rng(seed)
    match = np.arange(n, dtype=np.int64)
    free = np.ones(n, dtype=bool)
    rows = np.repeat(np.arange(n), np.diff(G.indptr))
    cols = G.indices
    jitter = 1e-9 * rng.random(G.nnz)                       # deterministic tie-break
    for _ in range(rounds):
        ok = free[rows] & free[cols] & (rows != cols) & (vwgt[rows] + vwgt[cols] <= max_vwgt)
        score = np.where(ok, G.data * (1.0 + jitter), 0.0)
        prop = _row_argmax(G, score)
        v = np.nonzero(prop >= 0)[0]
        mutual = v[prop[prop[v]] == v]
        if mutual.size == 0:
            break
        a = mutual[mutual < prop[mutual]]
        b = prop[a]
        match[a] = b
        match[b] = a
        free[a] = False
        free[b] = False
    return match


def _coarsen(G, vwgt, coords, match):
    n = G.shape[0]
    rep = np.minimum(np.arange(n), match)
    uniq, cmap = np.unique(rep, return_inverse=True)
    nc = uniq.size
    T = sp.csr_matrix((np.ones(n), (np.arange(n), cmap)), shape=(n, nc))
    Gc = (T.T @ G @ T).tocsr()
    Gc.setdiag(0.0)
    Gc.eliminate_zeros()
    vw = np.bincount(cmap, weights=vwgt, minlength=nc)
    cc = None
    if coords is not None:
        cc = np.stack([np.bincount(cmap, weights=coords[:, k] * vwgt, minlength=nc) / vw for k in range(coords.shape[1])], axis=1)
    return Gc, vw, cc, cmap


def edge_cut(G: sp.csr_matrix, part: np.ndarray) -> float:
    coo = G.tocoo()
    return float(coo.data[part[coo.row] != part[coo.col]].sum()) / 2.0


def _refine(G, vwgt, part, k, target, imbalance, sweeps=6, seed=0):
    """greedy k-way boundary refinement: per sweep, every boundary vertex looks at the part it is most connected to; the moves with
    positive gain (or zero gain towards a lighter part) that keep both parts within the balance band are applied on an
    independent set (no two adjacent vertices move in the same sweep)."""
    n = G.shape[0]
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(n), np.diff(G.indptr))
    cols = G.indices
    pw = np.bincount(part, weights=vwgt, minlength=k)
    hi = target * (1.0 + imbalance)
    for sw in range(sweeps):
        pc = part[cols]
        # connection of every vertex to every part it touches: sparse (vertex, part) sums
        key = rows * k + pc
        order = np.argsort(key, kind="stable")
        ks = key[order]
        starts = np.ones(ks.size, dtype=bool)
        starts[1:] = ks[1:] != ks[:-1]
        seg = np.nonzero(starts)[0]
        conn = np.add.reduceat(G.data[order], seg)
        cv, cp = ks[seg] // k, ks[seg] % k
        own = np.zeros(n)
        m_own = cp == part[cv]
        own[cv[m_own]] = conn[m_own]
        # best foreign part per vertex
        foreign = ~m_own
        if not foreign.any():
            break
        fv, fp, fc = cv[foreign], cp[foreign], conn[foreign]
        o2 = np.lexsort((-fc, fv))
        fv, fp, fc = fv[o2], fp[o2], fc[o2]
        first = np.ones(fv.size, dtype=bool)
        first[1:] = fv[1:] != fv[:-1]
        bv, bp, bc = fv[first], fp[first], fc[first]
        gain = bc - own[bv]
        cand = (gain > 0) | ((gain == 0) & (pw[bp] + vwgt[bv] < pw[part[bv]]))
        bv, bp, gain = bv[cand], bp[cand], gain[cand]
        if bv.size == 0:
            break
        # independent set among the candidates: a vertex moves only if it has the best (gain, priority) among its candidate neighbours
        prio = np.zeros(n)
        prio[bv] = gain + 1.0 + 1e-6 * rng.random(bv.size) * (np.abs(gain).max() + 1.0)
        nb_best = np.zeros(n)
        np.maximum.at(nb_best, rows, prio[cols])
        sel = prio[bv] > nb_best[bv]
        bv, bp = bv[sel], bp[sel]
        if bv.size == 0:
            break
        # balance: accept moves in order of gain while the destination stays below the upper bound
        order2 = np.argsort(-prio[bv], kind="stable")
        bv, bp = bv[order2], bp[order2]
        add = np.zeros(k)
        accepted = np.zeros(bv.size, dtype=bool)
        # vectorised prefix test per destination part
        for q in np.unique(bp):
            idx = np.nonzero(bp == q)[0]
            room = hi - pw[q]
            csum = np.cumsum(vwgt[bv[idx]])
            accepted[idx] = csum <= room
        bv, bp = bv[accepted], bp[accepted]
        if bv.size == 0:
            break
        np.subtract.at(pw, part[bv], vwgt[bv])
        np.add.at(pw, bp, vwgt[bv])
        part[bv] = bp
    return part


def _rebalance(G, vwgt, part, k, target, imbalance, coords=None):
    """push weight out of overloaded parts along their boundaries (used after projection when a level starts out of balance)"""
    n = G.shape[0]
    rows = np.repeat(np.arange(n), np.diff(G.indptr))
    cols = G.indices
    hi = target * (1.0 + imbalance)
    for _ in range(40):
        pw = np.bincount(part, weights=vwgt, minlength=k)
        over = np.nonzero(pw > hi)[0]
        if over.size == 0:
            break
        moved = False
        for q in over:
            # boundary vertices of q adjacent to lighter parts: move the ones with the largest outside connection first
            m = (part[rows] == q) & (part[cols] != q) & (pw[part[cols]] + 0.0 < pw[q])
            if not m.any():
                continue
            v, dest, wgt = rows[m], part[cols[m]], G.data[m]
            o = np.lexsort((-wgt, v))
            v, dest = v[o], dest[o]
            first = np.ones(v.size, dtype=bool)
            first[1:] = v[1:] != v[:-1]
            v, dest = v[first], dest[first]
            excess = pw[q] - target
            take = np.cumsum(vwgt[v]) <= excess
            if not take.any():
                take[0] = True
            v, dest = v[take], dest[take]
            # do not overload the destinations
            for d in np.unique(dest):
                idx = np.nonzero(dest == d)[0]
                room = hi - pw[d]
                ok = np.cumsum(vwgt[v[idx]]) <= room
                sel = idx[ok]
                if sel.size:
                    part[v[sel]] = d
                    pw[d] += vwgt[v[sel]].sum()
                    pw[q] -= vwgt[v[sel]].sum()
                    moved = True
        if not moved:
            break
    return part


def graph_partition(G: sp.csr_matrix, k: int, vwgt: np.ndarray | None = None, coords: np.ndarray | None = None, imbalance: float = 0.03,
                    coarsen_to: int | None = None, seed: int = 0) -> np.ndarray:
    """Part (0..k-1) of every vertex of the weighted graph G: multilevel k-way, vertex weights ``vwgt`` balanced to
    ``1 + imbalance``, edge weights = G.data.  ``coords`` (recommended for meshes) seed the initial partition of the coarsest graph
    by weighted coordinate bisection; without them the coarsest graph is grown from k seeds."""
    from .parallel import vertex_partition
    n = G.shape[0]
    if k <= 1:
        return np.zeros(n, dtype=np.int32)
    G = sp.csr_matrix(G, dtype=np.float64)
    G.setdiag(0.0)
    G.eliminate_zeros()
    vw = np.ones(n) if vwgt is None else np.asarray(vwgt, dtype=np.float64)
    total = vw.sum()
    target = total / k
    coarsen_to = coarsen_to or max(40 * k, 2000)
    levels = []
    Gl, vl, cl = G, vw, (None if coords is None else np.asarray(coords, dtype=np.float64))
    lvl = 0
    while Gl.shape[0] > coarsen_to and lvl < 40:
        match = heavy_edge_matching(Gl, vl, max_vwgt=1.5 * total / coarsen_to, seed=seed + lvl)
        Gc, vc, cc, cmap = _coarsen(Gl, vl, cl, match)
        if Gc.shape[0] > 0.92 * Gl.shape[0]:          # matching stalled (star-like graph)
            break
        levels.append((Gl, vl, cmap))
        Gl, vl, cl = Gc, vc, cc
        lvl += 1
    # initial partition on the coarsest graph
    if cl is not None:
        part = vertex_partition(cl, k, method="rcb", weights=vl).astype(np.int64)
    else:
        part = _grow_initial(Gl, vl, k, seed)
    part = _rebalance(Gl, vl, part, k, target, imbalance)
    part = _refine(Gl, vl, part, k, target, imbalance, sweeps=10, seed=seed)
    # uncoarsen
    for Gf, vf, cmap in reversed(levels):
        part = part[cmap]
        part = _rebalance(Gf, vf, part, k, target, imbalance)
        part = _refine(Gf, vf, part, k, target, imbalance, sweeps=4, seed=seed)
    return part.astype(np.int32)


def _grow_initial(G, vwgt, k, seed):
    """k seeds far apart (farthest-first in hop distance), parts grown breadth-first by weight"""
    from scipy.sparse.csgraph import breadth_first_order
    n = G.shape[0]
    rng = np.random.default_rng(seed)
    order, _ = breadth_first_order(G, int(rng.integers(n)), directed=False)
    order = np.concatenate([order, np.setdiff1d(np.arange(n), order)])     # other components at the end
    cw = np.cumsum(vwgt[order])
    part = np.empty(n, dtype=np.int64)
    part[order] = np.minimum((cw / (cw[-1] / k + 1e-300)).astype(np.int64), k - 1)
    return part


def mesh_vertex_weights(n_v, cells, cell_is_intra):
    """unknowns per vertex: one node per side the vertex touches (membrane vertices: two)"""
    cells = np.asarray(cells)
    ti = np.zeros(n_v, dtype=bool)
    te = np.zeros(n_v, dtype=bool)
    ti[cells[cell_is_intra].ravel()] = True
    te[cells[~cell_is_intra].ravel()] = True
    w = ti.astype(np.float64) + te.astype(np.float64)
    w[w == 0.0] = 1.0
    return w


def partition_mesh_vertices(coords, cells, k, cell_is_intra=None, intra_edge_weight: float = 4.0, imbalance: float = 0.03, seed: int = 0):
    """Owner rank of every vertex of a simplicial mesh: nodal graph with vertex weight = unknowns per vertex and edges of
    intracellular cells weighted ``intra_edge_weight`` times the extracellular ones (cutting a cell costs a coarse coupling)."""
    n_v = coords.shape[0]
    if cell_is_intra is None:
        G = nodal_graph(n_v, cells)
        vw = np.ones(n_v)
    else:
        cell_is_intra = np.asarray(cell_is_intra, dtype=bool)
        G = nodal_graph(n_v, cells, np.where(cell_is_intra, intra_edge_weight, 1.0))
        vw = mesh_vertex_weights(n_v, cells, cell_is_intra)
    return graph_partition(G, k, vw, coords, imbalance=imbalance, seed=seed)
