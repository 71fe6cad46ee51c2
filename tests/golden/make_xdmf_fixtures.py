"""Generates tests/golden/xdmf/*: small XDMF + HDF5 mesh files in the two layouts the reference reads
(src/CGx/utils/mixed_dim_problem.py:634-681, tag-name rule :137-145), written with a REAL HDF5 library so that the pure-Python
decoder of the native path (cgx_hip/hdf5_min.py, cgx_hip/xdmf.py) is checked against libhdf5's own encoding:

  square8.xdmf / square8.h5, square8_facets.xdmf / .h5   DOLFINx XDMFFile layout (generate_square_mesh.py:37-42):
        grids "mesh", "ct" / "mesh", "ft"; datasets /Mesh/mesh/{geometry,topology}, /MeshTags/<name>/{topology,Values};
        contiguous int64 / float64, the tag grids list their entities in a permuted order
  cube3_mesh.xdmf / cube3_mesh.h5, cube3_facets.xdmf / .h5   meshio layout (tags "under the same hierarchy as the mesh"): one grid
        "mesh" with the tags as an Attribute; chunked + shuffle + gzip datasets, int32 tags, a big-endian dataset
  deep.h5             chunk and group B-trees with more than one node (81 chunks; 300 members)
  latest.h5           libver="latest" file (layout message v4): the decoder must refuse it by name
  expected.npz        the arrays that were written

Run in THIS container with the image's conda interpreter (h5py 3.3.0 on libhdf5 1.10.6):
    /opt/conda/bin/python3.9 tests/golden/make_xdmf_fixtures.py
The mesh arrays come from the repo's own generators (no reference code involved)."""
import os
import sys

import h5py
import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
out = os.path.join(here, "xdmf")
os.makedirs(out, exist_ok=True)


def unit_square(N):
    xs = np.linspace(0.0, 1.0, N + 1)
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    coords = np.column_stack([X.ravel(), Y.ravel()])
    vid = lambda i, j: i * (N + 1) + j
    cells = []
    for i in range(N):
        for j in range(N):
            a, b, c, d = vid(i, j), vid(i + 1, j), vid(i, j + 1), vid(i + 1, j + 1)
            cells += [(a, b, d), (a, c, d)]
    return coords, np.array(cells, dtype=np.int64)


def unit_cube(N):
    xs = np.linspace(0.0, 1.0, N + 1)
    X, Y, Z = np.meshgrid(xs, xs, xs, indexing="ij")
    coords = np.column_stack([X.ravel(), Y.ravel(), Z.ravel()])
    vid = lambda i, j, k: (i * (N + 1) + j) * (N + 1) + k
    cells = []
    for i in range(N):
        for j in range(N):
            for k in range(N):
                v = [vid(i + a, j + b, k + c) for a in (0, 1) for b in (0, 1) for c in (0, 1)]
                for t in ((0, 1, 3, 7), (0, 1, 5, 7), (0, 2, 3, 7), (0, 2, 6, 7), (0, 4, 5, 7), (0, 4, 6, 7)):
                    cells.append([v[q] for q in t])
    return coords, np.array(cells, dtype=np.int64)


def box_tags(coords, cells, lo=0.25, hi=0.75):
    mid = coords[cells].mean(axis=1)
    inside = np.all((mid > lo) & (mid < hi), axis=1)
    return np.where(inside, 1, 2).astype(np.int32)


def facets_between(cells, tags):
    """sorted vertex tuples of the facets between cells of different tags (value 4) and on the exterior boundary (value 5)"""
    nv = cells.shape[1]
    table = {}
    for c, cell in enumerate(cells.tolist()):
        for lf in range(nv):
            key = tuple(sorted(cell[:lf] + cell[lf + 1:]))
            table.setdefault(key, []).append(c)
    fv, val = [], []
    for key, cs in sorted(table.items()):
        if len(cs) == 1:
            fv.append(key); val.append(5)
        elif tags[cs[0]] != tags[cs[1]]:
            fv.append(key); val.append(4)
    return np.array(fv, dtype=np.int64), np.array(val, dtype=np.int32)


def xdmf_text(grids):
    s = ['<?xml version="1.0"?>', '<!DOCTYPE Xdmf SYSTEM "Xdmf.dtd" []>', '<Xdmf Version="3.0" xmlns:xi="http://www.w3.org/2001/XInclude">', "  <Domain>"]
    s += grids
    s += ["  </Domain>", "</Xdmf>", ""]
    return "\n".join(s)


expected = {}
rng = np.random.default_rng(7)

# ---- DOLFINx layout, 2D --------------------------------------------------------------------------------------------------
coords, cells = unit_square(8)
ct = box_tags(coords, cells)
fv, fval = facets_between(cells, ct)
perm_c = rng.permutation(len(cells))
perm_f = rng.permutation(len(fv))
for fname, tagname, ttype, npe, topo, vals in (("square8", "ct", "Triangle", 3, cells[perm_c], ct[perm_c]),
                                                ("square8_facets", "ft", "PolyLine", 2, fv[perm_f], fval[perm_f])):
    with h5py.File(os.path.join(out, fname + ".h5"), "w") as h:
        h.create_dataset("/Mesh/mesh/geometry", data=coords)
        h.create_dataset("/Mesh/mesh/topology", data=cells)
        h.create_dataset(f"/MeshTags/{tagname}/topology", data=topo)
        h.create_dataset(f"/MeshTags/{tagname}/Values", data=vals.reshape(-1, 1))
    g = [f'    <Grid Name="mesh" GridType="Uniform">',
         f'      <Topology TopologyType="Triangle" NumberOfElements="{len(cells)}" NodesPerElement="3">',
         f'        <DataItem Dimensions="{len(cells)} 3" NumberType="Int" Format="HDF">{fname}.h5:/Mesh/mesh/topology</DataItem>',
         f'      </Topology>',
         f'      <Geometry GeometryType="XY">',
         f'        <DataItem Dimensions="{len(coords)} 2" Format="HDF">{fname}.h5:/Mesh/mesh/geometry</DataItem>',
         f'      </Geometry>',
         f'    </Grid>',
         f'    <Grid Name="{tagname}" GridType="Uniform">',
         f'      <xi:include xpointer="xpointer(/Xdmf/Domain/Grid/Geometry)" />',
         f'      <Topology TopologyType="{ttype}" NumberOfElements="{len(topo)}" NodesPerElement="{npe}">',
         f'        <DataItem Dimensions="{len(topo)} {npe}" NumberType="Int" Format="HDF">{fname}.h5:/MeshTags/{tagname}/topology</DataItem>',
         f'      </Topology>',
         f'      <Attribute Name="{tagname}" AttributeType="Scalar" Center="Cell">',
         f'        <DataItem Dimensions="{len(topo)} 1" Format="HDF">{fname}.h5:/MeshTags/{tagname}/Values</DataItem>',
         f'      </Attribute>',
         f'    </Grid>']
    open(os.path.join(out, fname + ".xdmf"), "w").write(xdmf_text(g))
expected.update(sq_coords=coords, sq_cells=cells, sq_ct=ct, sq_fv=fv, sq_fval=fval, sq_perm_c=perm_c, sq_perm_f=perm_f)

# ---- meshio layout, 3D, compressed ------------------------------------------------------------------------------------------
coords, cells = unit_cube(3)
ct = box_tags(coords, cells, 0.3, 0.7)
fv, fval = facets_between(cells, ct)
with h5py.File(os.path.join(out, "cube3_mesh.h5"), "w") as h:
    h.create_dataset("data0", data=coords, chunks=(16, 3), compression="gzip", compression_opts=4, shuffle=True)
    h.create_dataset("data1", data=cells, chunks=(50, 4), compression="gzip", compression_opts=4)
    h.create_dataset("data2", data=ct, chunks=(64,), compression="gzip", shuffle=True, fletcher32=True)
    h.create_dataset("big_endian", data=coords.astype(">f8"))
    h.create_dataset("small", data=np.arange(5, dtype=np.int16))          # compact-size candidates stay contiguous in h5py; kept as a dtype case
g = [f'    <Grid Name="mesh" GridType="Uniform">',
     f'      <Geometry GeometryType="XYZ">',
     f'        <DataItem DataType="Float" Dimensions="{len(coords)} 3" Format="HDF" Precision="8">cube3_mesh.h5:/data0</DataItem>',
     f'      </Geometry>',
     f'      <Topology NumberOfElements="{len(cells)}" TopologyType="Tetrahedron">',
     f'        <DataItem DataType="Int" Dimensions="{len(cells)} 4" Format="HDF" Precision="8">cube3_mesh.h5:/data1</DataItem>',
     f'      </Topology>',
     f'      <Attribute AttributeType="Scalar" Center="Cell" Name="label">',
     f'        <DataItem DataType="Int" Dimensions="{len(cells)}" Format="HDF" Precision="4">cube3_mesh.h5:/data2</DataItem>',
     f'      </Attribute>',
     f'    </Grid>']
open(os.path.join(out, "cube3_mesh.xdmf"), "w").write(xdmf_text(g))
with h5py.File(os.path.join(out, "cube3_facets.h5"), "w") as h:
    h.create_dataset("data0", data=coords, chunks=(16, 3), compression="gzip", shuffle=True)
    h.create_dataset("data1", data=fv, chunks=(40, 3), compression="gzip", shuffle=True)
    h.create_dataset("data2", data=fval, chunks=(32,), compression="gzip")
# inline XML data items for the tags: the other DataItem format of the XDMF standard
g = [f'    <Grid Name="mesh" GridType="Uniform">',
     f'      <Geometry GeometryType="XYZ">',
     f'        <DataItem DataType="Float" Dimensions="{len(coords)} 3" Format="HDF" Precision="8">cube3_facets.h5:/data0</DataItem>',
     f'      </Geometry>',
     f'      <Topology NumberOfElements="{len(fv)}" TopologyType="Triangle">',
     f'        <DataItem DataType="Int" Dimensions="{len(fv)} 3" Format="HDF" Precision="8">cube3_facets.h5:/data1</DataItem>',
     f'      </Topology>',
     f'      <Attribute AttributeType="Scalar" Center="Cell" Name="label">',
     f'        <DataItem DataType="Int" Dimensions="{len(fv)}" Format="XML" Precision="4">' + " ".join(str(int(v)) for v in fval) + '</DataItem>',
     f'      </Attribute>',
     f'    </Grid>']
open(os.path.join(out, "cube3_facets.xdmf"), "w").write(xdmf_text(g))
expected.update(cu_coords=coords, cu_cells=cells, cu_ct=ct, cu_fv=fv, cu_fval=fval)

# ---- index structures with more than one node: 81 one-row chunks (chunk B-tree of depth 2), a group with 300 members ---------
with h5py.File(os.path.join(out, "deep.h5"), "w") as h:
    h.create_dataset("many_chunks", data=expected["sq_coords"], chunks=(1, 2))
    h.create_dataset("many_chunks_z", data=expected["sq_cells"], chunks=(2, 3), compression="gzip", shuffle=True)
    g = h.create_group("wide")
    for i in range(300):
        g.create_dataset(f"d{i:03d}", data=np.array([i, 2 * i], dtype=np.int32))

with h5py.File(os.path.join(out, "latest.h5"), "w", libver="latest") as h:
    h.create_dataset("x", data=np.arange(100.0), chunks=(10,))

np.savez_compressed(os.path.join(out, "expected.npz"), **expected)
print("wrote", sorted(os.listdir(out)), "h5py", h5py.__version__, "hdf5", h5py.version.hdf5_version)
