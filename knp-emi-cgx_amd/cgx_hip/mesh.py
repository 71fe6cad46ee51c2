"""Mesh containers and the synthetic square / cube generators of the reference.

Replaces (host side, setup only):
  * ``dolfinx.mesh.create_unit_square / create_unit_cube`` + the subdomain / facet markers of
    reference src/CGx/utils/misc.py:99-195 (square) and :256-398 (cube), as driven by
    src/CGx/utils/generate_square_mesh.py:28-42;
  * the '+' = intracellular ordering of interior-facet integration entities,
    src/CGx/utils/mixed_dim_problem.py:705-729.

Mesh input (``load_mesh``): an existing ``.xdmf`` file (+ its ``.h5``) is read by cgx_hip/xdmf.py (DOLFINx and meshio layouts,
decoded without an HDF5 library); ``*.npz`` files with arrays ``coords, cells, cell_tags, facets, facet_tags`` are loaded as
they are; a ``cell_tag_file`` that does not exist and is named ``square<N>.xdmf`` / ``cube<N>.xdmf`` /
``tissue<dim>d_<N>_<m>[_g<gap>|_w<width>].xdmf`` is generated natively (what generate_square_mesh.py would have written).
"""
from __future__ import annotations

import os
import re

import numpy as np
import torch


class _Geometry:
    def __init__(self, x, dim):
        self.x = x
        self.dim = dim


class _Topology:
    def __init__(self, dim):
        self.dim = dim


class MeshTags:
    """dolfinx.mesh.MeshTags stand-in: ``values[i]`` tags entity ``indices[i]``."""

    def __init__(self, dim, indices, values, name="tags"):
        self.dim = dim
        self.indices = np.asarray(indices, dtype=np.int32)
        self.values = np.asarray(values, dtype=np.int32)
        self.name = name

    def find(self, tag):
        return self.indices[self.values == tag]


class Mesh:
    """P1 simplex mesh. ``geometry.x`` is (n_v, dim) float64 (metres after scaling)."""

    def __init__(self, coords, cells, device=None):
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)
        self.geometry = _Geometry(coords, coords.shape[1])
        self.topology = _Topology(coords.shape[1])
        self.num_vertices = coords.shape[0]
        self.num_cells = self.cells.shape[0]
        self.device = device if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.comm = None


def create_unit_square(N):
    """Right-diagonal split: box (v0,v1,v2,v3) -> (v0,v1,v3),(v0,v2,v3) (DOLFINx default)."""
    t = np.linspace(0.0, 1.0, N + 1)
    t = np.arange(N + 1) / float(N)
    xx, yy = np.meshgrid(t, t)
    coords = np.column_stack([xx.reshape(-1), yy.reshape(-1)])
    j, i = np.divmod(np.arange(N * N), N)
    v0 = j * (N + 1) + i
    quad = np.column_stack([v0, v0 + 1, v0 + N + 1, v0 + N + 2])
    tri = np.empty((N * N, 2, 3), dtype=np.int32)
    tri[:, 0, :] = quad[:, [0, 1, 3]]
    tri[:, 1, :] = quad[:, [0, 2, 3]]
    return coords, tri.reshape(-1, 3)


def create_unit_cube(N):
    """Six tetrahedra per box around the v0-v7 diagonal (DOLFINx create_unit_cube)."""
    t = np.arange(N + 1) / float(N)
    zz, yy, xx = np.meshgrid(t, t, t, indexing="ij")
    coords = np.column_stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)])
    s = N + 1
    idx = np.arange(N ** 3)
    k, rem = np.divmod(idx, N * N)
    j, i = np.divmod(rem, N)
    v0 = k * s * s + j * s + i
    c = np.column_stack([v0, v0 + 1, v0 + s, v0 + s + 1, v0 + s * s, v0 + s * s + 1, v0 + s * s + s, v0 + s * s + s + 1])
    pat = np.array([[0, 1, 3, 7], [0, 1, 7, 5], [0, 5, 7, 4], [0, 3, 2, 7], [0, 6, 4, 7], [0, 2, 6, 7]])
    tets = c[:, pat]                                         # (N^3, 6, 4)
    return coords, tets.reshape(-1, 4).astype(np.int32)


def mark_subdomains_box(coords, cells, lo=0.25, hi=0.75):
    """INTRA=1 where every vertex of the cell lies in [lo,hi]^d, EXTRA=2 elsewhere
    (misc.py:99-135, :256-297: locate_entities marks an entity when all its vertices satisfy the marker)."""
    ok = ((coords >= lo) & (coords <= hi)).all(axis=1)
    inside = ok[cells].all(axis=1)
    return np.where(inside, 1, 2).astype(np.int32)


def mark_subdomains_sheets(coords, cells, N, m, width=1):
    """Membrane-dominated tissue surrogate: m^d cubic cells of (B - width)^d voxels, B = (N - width) // m, separated from each
    other and from the exterior boundary by extracellular SHEETS ``width`` voxels thick.  With width 1 the extracellular space
    has no vertex off the membranes -- the regime of the reference's reconstructions, where 73-92 % of all vertices are membrane
    vertices (src/CGx/utils/emimesh_data.xlsx: npoints_membrane / npoints): B = 4 gives 87.5 %, B = 5 gives 78 %."""
    if (N - width) % m:
        raise ValueError("sheet lattice: N - width must be divisible by m")
    B = (N - width) // m
    if B - width < 1:
        raise ValueError("sheet lattice: blocks too small for the sheet width")
    dim = coords.shape[1]
    cen = coords[cells].mean(axis=1)
    span = coords.max(axis=0) - coords.min(axis=0)
    vox = np.minimum((np.floor((cen - coords.min(axis=0)) / span * N)).astype(np.int64), N - 1)
    blk, loc = vox // B, vox % B
    inside = ((loc >= width) & (blk < m)).all(axis=1)
    lin = np.zeros(len(cells), dtype=np.int64)
    for a in range(dim):
        lin = lin * m + np.minimum(blk[:, a], m - 1)
    return np.where(inside, 2 + lin, 1).astype(np.int32)


def mark_subdomains_lattice(coords, cells, N, m, gap=1):
    """Tissue surrogate: the N^d voxel grid is cut into m^d blocks; each block holds one
    cubic 'cell' separated from its neighbours by ``gap`` voxels of extracellular space on every side.  Cell b gets
    tag 2 + b (one tag per cell, like ``ics_tags: !range [2, K]`` of the reference's tissue configs), ECS tag 1."""
    if N % m:
        raise ValueError("lattice mesh: N must be divisible by m")
    B = N // m
    if B - 2 * gap < 1:
        raise ValueError("lattice mesh: blocks too small for the gap")
    dim = coords.shape[1]
    cen = coords[cells].mean(axis=1)
    span = coords.max(axis=0) - coords.min(axis=0)
    vox = np.minimum((np.floor((cen - coords.min(axis=0)) / span * N)).astype(np.int64), N - 1)
    blk, loc = vox // B, vox % B
    inside = ((loc >= gap) & (loc < B - gap)).all(axis=1)
    lin = np.zeros(len(cells), dtype=np.int64)
    for a in range(dim):
        lin = lin * m + blk[:, a]
    return np.where(inside, 2 + lin, 1).astype(np.int32)


def _facet_table(cells):
    """All (cell, local facet) pairs with their sorted vertex tuples; local facet i is opposite vertex i."""
    nc, nv = cells.shape
    blocks = []
    for lf in range(nv):
        keep = [a for a in range(nv) if a != lf]
        blocks.append(np.sort(cells[:, keep], axis=1))
    verts = np.vstack(blocks)
    cell_of = np.tile(np.arange(nc, dtype=np.int64), nv)
    lf_of = np.repeat(np.arange(nv, dtype=np.int64), nc)
    return verts, cell_of, lf_of


def build_facets(cells):
    """Unique facets of the mesh: returns (facet_vertices (n_f, d), c0, l0, c1, l1) with c1 = -1 on
    the exterior boundary."""
    verts, cell_of, lf_of = _facet_table(cells)
    # lexicographic order of the sorted vertex tuples through ONE integer key per facet where it fits into 64 bits (any 2D mesh, 3D
    # meshes below 2^21 vertices), through a two-key sort otherwise -- instead of one stable sort per vertex column
    nvert = int(cells.max()) + 1 if cells.size else 1
    d = verts.shape[1]
    if nvert ** d < 2 ** 63:
        key = verts[:, 0].astype(np.int64)
        for k in range(1, d):
            key = key * nvert + verts[:, k]
        order = np.argsort(key, kind="stable")
    elif d == 3 and nvert ** 2 < 2 ** 63:
        order = np.lexsort((verts[:, 2], verts[:, 0].astype(np.int64) * nvert + verts[:, 1]))
    else:
        order = np.lexsort(tuple(verts[:, k] for k in range(d - 1, -1, -1)))
    sv = verts[order]
    first = np.ones(len(order), dtype=bool)
    first[1:] = (sv[1:] != sv[:-1]).any(axis=1)
    fid = np.cumsum(first) - 1
    nf = int(fid[-1]) + 1
    c0 = np.full(nf, -1, dtype=np.int64); l0 = np.full(nf, -1, dtype=np.int64)
    c1 = np.full(nf, -1, dtype=np.int64); l1 = np.full(nf, -1, dtype=np.int64)
    pos_first = np.nonzero(first)[0]
    c0[:] = cell_of[order[pos_first]]
    l0[:] = lf_of[order[pos_first]]
    second = ~first
    c1[fid[second]] = cell_of[order[second]]
    l1[fid[second]] = lf_of[order[second]]
    return sv[pos_first], c0, l0, c1, l1


def gamma_integration_entities(cells, cell_tags, intra_tags, extra_tags, facet_tags=None):
    """Rows (cell+, lf+, cell-, lf-) of all interior facets separating an intra from an extra cell,
    '+' = intracellular (mixed_dim_problem.py:717-729), plus the tag of each facet.

    ``facet_tags``: optional MeshTags-like (vertices (n,d) sorted, values) restricting/tagging the
    membrane; when absent every intra/extra interface facet is a membrane facet with tag 4
    (what misc.py:137-195 / :299-398 produce on the synthetic meshes)."""
    fverts, c0, l0, c1, l1 = build_facets(cells)
    interior = c1 >= 0
    is_i0 = np.isin(cell_tags[c0], intra_tags)
    is_e0 = np.isin(cell_tags[c0], extra_tags)
    is_i1 = np.zeros_like(is_i0); is_e1 = np.zeros_like(is_e0)
    is_i1[interior] = np.isin(cell_tags[c1[interior]], intra_tags)
    is_e1[interior] = np.isin(cell_tags[c1[interior]], extra_tags)
    g = interior & ((is_i0 & is_e1) | (is_e0 & is_i1))
    idx = np.nonzero(g)[0]
    swap = is_e0[idx]
    cp = np.where(swap, c1[idx], c0[idx]); lp = np.where(swap, l1[idx], l0[idx])
    cm = np.where(swap, c0[idx], c1[idx]); lm = np.where(swap, l0[idx], l1[idx])
    ent = np.column_stack([cp, lp, cm, lm]).astype(np.int32)
    tags = np.full(len(idx), 4, dtype=np.int32)
    if isinstance(facet_tags, str) and facet_tags == "intra":      # 'membrane tag = tag of its cell' (tissue configs)
        tags = cell_tags[cp].astype(np.int32)
    elif facet_tags is not None:
        fv, fvals = facet_tags
        key = {tuple(r): v for r, v in zip(np.sort(np.asarray(fv), axis=1).tolist(), np.asarray(fvals).tolist())}
        tags = np.array([key.get(tuple(r), -1) for r in fverts[idx].tolist()], dtype=np.int32)
    order = np.lexsort((ent[:, 1], ent[:, 0]))
    return ent[order], tags[order], fverts[idx][order]


def exterior_facets(cells):
    fverts, c0, l0, c1, l1 = build_facets(cells)
    ext = c1 < 0
    return fverts[ext], c0[ext], l0[ext]


def facet_quadrature(dim, degree=10):
    """Quadrature on the reference facet exact to ``degree`` (mixed_dim_problem.py:733: degree 10).
    Edges: Gauss-Legendre with m = degree//2 + 1 points (basix's Gauss-Jacobi scheme).
    Triangles: collapsed Gauss-Jacobi m x m (basix uses a 25-point Xiao-Gimbutas rule of the same
    degree; both are exact for the polynomial part)."""
    from scipy.special import roots_jacobi
    m = degree // 2 + 1
    xg, wg = np.polynomial.legendre.leggauss(m)
    if dim == 2:
        s = (xg + 1.0) / 2.0
        return np.column_stack([1.0 - s, s]), wg / 2.0
    xj, wj = roots_jacobi(m, 1.0, 0.0)
    u = (xj + 1.0) / 2.0
    t = (xg + 1.0) / 2.0
    pts, wts = [], []
    for a in range(m):
        for b in range(m):
            l1 = u[a]
            l2 = t[b] * (1.0 - u[a])
            pts.append((1.0 - l1 - l2, l1, l2))
            wts.append((wj[a] / 4.0) * (wg[b] / 2.0))
    wts = np.array(wts)
    return np.array(pts), wts / wts.sum()


_SYN = re.compile(r"(square|cube)(\d+)")
_TISSUE = re.compile(r"tissue(\d)d_(\d+)_(\d+)(?:_g(\d+))?(?:_w(\d+))?")     # tissue<dim>d_<N>_<m>[_g<gap> | _w<sheet width>]


def load_mesh(mesh_file, facet_file, conversion_factor=1.0):
    """Returns (coords, cells, cell_tags, facet_tags or None, description)."""
    base = os.path.basename(mesh_file)
    if mesh_file.endswith(".npz") and os.path.exists(mesh_file):
        d = np.load(mesh_file, allow_pickle=False)
        ft = (d["facets"], d["facet_tags"]) if "facets" in d.files else None
        return d["coords"] * conversion_factor, d["cells"].astype(np.int32), d["cell_tags"].astype(np.int32), ft, mesh_file
    if mesh_file.endswith(".xdmf") and os.path.exists(mesh_file):
        from . import xdmf
        ff = facet_file if (facet_file and os.path.exists(facet_file)) else mesh_file
        coords, cells, tags, ft = xdmf.read_mesh_and_tags(mesh_file, ff)
        return coords * conversion_factor, cells, tags, ft, f"{mesh_file} (XDMF)"
    m = _TISSUE.search(base)
    if m:
        dim, N, nb, gap = int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4) or 1)
        coords, cells = create_unit_square(N) if dim == 2 else create_unit_cube(N)
        if m.group(5):
            width = int(m.group(5))
            tags = mark_subdomains_sheets(coords, cells, N, nb, width)
            return coords * conversion_factor, cells, tags, "intra", f"generated tissue surrogate {dim}D N={N}, {nb}^{dim} cells, extracellular sheets {width} wide"
        tags = mark_subdomains_lattice(coords, cells, N, nb, gap)
        return coords * conversion_factor, cells, tags, "intra", f"generated tissue surrogate {dim}D N={N}, {nb}^{dim} cells, gap {gap}"
    m = _SYN.search(base)
    if m:
        kind, N = m.group(1), int(m.group(2))
        coords, cells = create_unit_square(N) if kind == "square" else create_unit_cube(N)
        tags = mark_subdomains_box(coords, cells)
        return coords * conversion_factor, cells, tags, None, f"generated {kind}{N}"
    raise RuntimeError(f"Cannot read mesh '{mesh_file}': the file does not exist and its name is not one of the generated "
                       "meshes ('square<N>', 'cube<N>', 'tissue<dim>d_<N>_<m>...'); .xdmf (+ .h5) and .npz meshes are read from disk.")
