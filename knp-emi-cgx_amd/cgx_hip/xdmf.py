"""XDMF mesh / mesh-tag input for the native path (host side, setup only).

Replaces ``dolfinx.io.XDMFFile.read_mesh`` + ``read_meshtags(mesh, name=...)`` as used by the reference's
``MixedDimensionalProblem.setup_domain`` (src/CGx/utils/mixed_dim_problem.py:634-681) with the tag-name rule of
``:137-145``: files written by DOLFINx (``generate_square_mesh.py:37-42``: grid "mesh" + tag grids "ct" / "ft" that list their
own entities) and files whose tags sit "under the same hierarchy as the mesh" (one grid "mesh" carrying the tags as an
Attribute, the meshio layout of the tissue reconstructions).  Heavy data: ``Format="HDF"`` through cgx_hip/hdf5_min.py,
``Format="XML"`` inline.  Output only: plain NumPy arrays (the device never sees these files).
"""
from __future__ import annotations

import os
import xml.etree.ElementTree as ET

import numpy as np

from . import hdf5_min

_NODES = {"polyvertex": 1, "polyline": 2, "triangle": 3, "tetrahedron": 4, "quadrilateral": 4, "hexahedron": 8}


class XdmfError(RuntimeError):
    pass


def _local(tag):
    return tag.rsplit("}", 1)[-1]


class XdmfFile:
    def __init__(self, path):
        self.path = str(path)
        self.dir = os.path.dirname(os.path.abspath(self.path))
        try:
            self.root = ET.parse(self.path).getroot()
        except ET.ParseError as exc:
            raise XdmfError(f"{self.path}: not an XDMF (XML) file: {exc}") from exc
        self._h5 = {}
        self.grids = {}
        for g in self.root.iter():
            if _local(g.tag) == "Grid" and g.get("GridType", "Uniform") == "Uniform":
                self.grids.setdefault(g.get("Name", ""), g)

    def _child(self, node, name):
        for c in node:
            if _local(c.tag) == name:
                return c
        return None

    def _data(self, item):
        if item is None:
            return None
        fmt = item.get("Format", "XML").upper()
        dims = [int(v) for v in item.get("Dimensions", "").split()]
        text = (item.text or "").strip()
        if fmt == "HDF":
            fname, _, dset = text.partition(":")
            fpath = os.path.join(self.dir, fname.strip())
            if fpath not in self._h5:
                if not os.path.exists(fpath):
                    raise XdmfError(f"{self.path}: heavy-data file '{fname.strip()}' not found next to it")
                self._h5[fpath] = hdf5_min.Hdf5File(fpath)
            arr = self._h5[fpath].read(dset.strip())
        elif fmt == "XML":
            kind = item.get("NumberType", item.get("DataType", "Float")).lower()
            arr = np.array(text.split(), dtype=np.int64 if kind in ("int", "uint") else np.float64)
        else:
            raise XdmfError(f"{self.path}: DataItem format '{fmt}' is not supported (HDF and XML are)")
        if dims and int(np.prod(dims)) == arr.size:
            arr = arr.reshape(dims)
        return arr

    def grid(self, name):
        """(topology (n, nodes) int64, geometry (n_points, gdim) or None, attribute values or None) of the uniform grid ``name``"""
        if name not in self.grids:
            raise XdmfError(f"{self.path}: no grid named '{name}' (grids: {sorted(self.grids)})")
        g = self.grids[name]
        topo = self._child(g, "Topology")
        if topo is None:
            raise XdmfError(f"{self.path}: grid '{name}' has no Topology")
        ttype = topo.get("TopologyType", topo.get("Type", "")).lower()
        if ttype not in _NODES or ttype in ("quadrilateral", "hexahedron"):
            raise XdmfError(f"{self.path}: topology type '{ttype}' is not supported (P1 simplices only)")
        cells = np.asarray(self._data(self._child(topo, "DataItem")), dtype=np.int64).reshape(-1, _NODES[ttype])
        geo = self._child(g, "Geometry")
        pts = None
        if geo is not None:
            pts = np.asarray(self._data(self._child(geo, "DataItem")), dtype=np.float64)
            if pts.ndim == 1:
                pts = pts.reshape(-1, 3 if geo.get("GeometryType", "XYZ").upper() == "XYZ" else 2)
        att = self._child(g, "Attribute")
        vals = None
        if att is not None:
            vals = np.asarray(self._data(self._child(att, "DataItem"))).reshape(-1)
        return cells, pts, vals


def _match_rows(reference_rows, rows):
    """index into ``reference_rows`` of every row of ``rows`` (vertex tuples compared as sets); -1 where absent"""
    a = np.sort(np.asarray(reference_rows, dtype=np.int64), axis=1)
    b = np.sort(np.asarray(rows, dtype=np.int64), axis=1)
    if a.shape == b.shape and np.array_equal(a, b):
        return np.arange(len(a))
    nv = int(max(a.max(initial=0), b.max(initial=0))) + 1
    if float(nv) ** a.shape[1] < 2 ** 62:
        ka = np.zeros(len(a), dtype=np.int64)
        kb = np.zeros(len(b), dtype=np.int64)
        for c in range(a.shape[1]):
            ka = ka * nv + a[:, c]
            kb = kb * nv + b[:, c]
    else:                                       # huge meshes: compare through a structured view
        ka = np.ascontiguousarray(a).view([("", np.int64)] * a.shape[1]).reshape(-1)
        kb = np.ascontiguousarray(b).view([("", np.int64)] * b.shape[1]).reshape(-1)
    order = np.argsort(ka, kind="stable")
    pos = np.searchsorted(ka[order], kb)
    pos = np.clip(pos, 0, len(order) - 1)
    hit = ka[order][pos] == kb
    return np.where(hit, order[pos], -1)


def read_mesh_and_tags(mesh_file, facet_file, ct_name=None, ft_name=None):
    """Returns (coords (n_v, dim), cells (n_c, dim+1) int32, cell_tags (n_c,) int32, (facet_vertices, facet_values) or None).

    ``ct_name`` / ``ft_name`` default to the reference's rule (mixed_dim_problem.py:137-145): "ct"/"ft" when the file name
    contains "square" or both tags live in one file, "mesh" otherwise."""
    same = os.path.abspath(mesh_file) == os.path.abspath(facet_file)
    if ct_name is None:
        ct_name = "ct" if ("square" in str(mesh_file) or same) else "mesh"
    if ft_name is None:
        ft_name = "ft" if ("square" in str(mesh_file) or same) else "mesh"
    mf = XdmfFile(mesh_file)
    mesh_grid = "mesh" if "mesh" in mf.grids else next(iter(mf.grids), None)
    if mesh_grid is None:
        raise XdmfError(f"{mesh_file}: no uniform grid")
    cells, coords, vals0 = mf.grid(mesh_grid)
    if coords is None:
        raise XdmfError(f"{mesh_file}: grid '{mesh_grid}' has no Geometry")
    dim = cells.shape[1] - 1
    if dim not in (2, 3):
        raise XdmfError(f"{mesh_file}: cells with {cells.shape[1]} vertices (triangles or tetrahedra expected)")
    coords = np.ascontiguousarray(coords[:, :dim])          # DOLFINx pads 2D geometry to XYZ in some versions
    if cells.min(initial=0) < 0 or cells.max(initial=0) >= len(coords):
        raise XdmfError(f"{mesh_file}: topology refers to points outside the geometry")
    # cell tags: an Attribute of the mesh grid itself, or a tag grid with its own list of cells.  A file whose only grid has
    # another name (meshio writes "Grid") is taken as "tags under the same hierarchy as the mesh".
    if ct_name not in mf.grids and len(mf.grids) == 1 and vals0 is not None:
        ct_name = mesh_grid
    if ct_name == mesh_grid:
        if vals0 is None:
            raise XdmfError(f"{mesh_file}: grid '{mesh_grid}' carries no Attribute with the cell tags")
        cell_tags = np.asarray(vals0, dtype=np.int64)
        if cell_tags.size != len(cells):
            raise XdmfError(f"{mesh_file}: {cell_tags.size} cell tags for {len(cells)} cells")
    else:
        tcells, _, tvals = mf.grid(ct_name)
        if tvals is None or tcells.shape[1] != cells.shape[1]:
            raise XdmfError(f"{mesh_file}: grid '{ct_name}' is not a cell-tag grid of this mesh")
        idx = _match_rows(cells, tcells)
        if (idx < 0).any():
            raise XdmfError(f"{mesh_file}: grid '{ct_name}' lists cells that are not in the mesh")
        cell_tags = np.zeros(len(cells), dtype=np.int64)
        cell_tags[idx] = tvals
    # facet tags
    ff = mf if same else XdmfFile(facet_file)
    facet_tags = None
    if ft_name not in ff.grids and len(ff.grids) == 1:
        ft_name = next(iter(ff.grids))
    if ft_name in ff.grids:
        fcells, fpts, fvals = ff.grid(ft_name)
        if fvals is not None and fcells.shape[1] == dim:
            if not same and ft_name == "mesh" and fpts is not None and (len(fpts) != len(coords) or not np.allclose(fpts[:, :dim], coords)):
                raise XdmfError(f"{facet_file}: the facet file's points differ from the mesh file's")
            facet_tags = (np.asarray(fcells, dtype=np.int64), np.asarray(fvals, dtype=np.int64))
    return coords, cells.astype(np.int32), cell_tags.astype(np.int32), facet_tags
