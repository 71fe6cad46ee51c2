"""Mesh input in the reference's file format: XDMF + HDF5 as read by ``XDMFFile.read_mesh / read_meshtags``
in src/CGx/utils/mixed_dim_problem.py:634-681, decoded by cgx_hip/hdf5_min.py + cgx_hip/xdmf.py without an HDF5 library.
The fixtures under tests/golden/xdmf were written by libhdf5 1.10.6 (h5py) with tests/golden/make_xdmf_fixtures.py;
``expected.npz`` holds the arrays that went in."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden", "xdmf")


@pytest.fixture(scope="module")
def expected():
    return np.load(os.path.join(G, "expected.npz"))


def test_hdf5_decoder_reads_what_libhdf5_wrote(expected):
    from cgx_hip import hdf5_min
    f = hdf5_min.Hdf5File(os.path.join(G, "square8.h5"))
    assert f.keys("/") == ["Mesh", "MeshTags"] and f.keys("/Mesh/mesh") == ["geometry", "topology"]
    geo, topo = f.read("/Mesh/mesh/geometry"), f.read("/Mesh/mesh/topology")
    assert geo.dtype == np.float64 and topo.dtype == np.int64
    assert np.array_equal(geo, expected["sq_coords"]) and np.array_equal(topo, expected["sq_cells"])
    assert np.array_equal(f.read("/MeshTags/ct/Values").ravel(), expected["sq_ct"][expected["sq_perm_c"]])
    # chunked + shuffle + deflate (+ fletcher32), int32, big-endian: the meshio layout
    f = hdf5_min.Hdf5File(os.path.join(G, "cube3_mesh.h5"))
    assert np.array_equal(f.read("data0"), expected["cu_coords"])
    assert np.array_equal(f.read("/data1"), expected["cu_cells"])
    v = f.read("data2")
    assert v.dtype == np.int32 and np.array_equal(v, expected["cu_ct"])
    be = f.read("big_endian")
    assert be.dtype == np.float64 and np.array_equal(be, expected["cu_coords"])
    assert np.array_equal(f.read("small"), np.arange(5, dtype=np.int16))
    with pytest.raises(KeyError):
        f.read("/no/such/dataset")


def test_hdf5_decoder_walks_multi_node_btrees(expected):
    """81 one-row chunks and a group of 300 members: chunk and group B-trees of depth two"""
    from cgx_hip import hdf5_min
    f = hdf5_min.Hdf5File(os.path.join(G, "deep.h5"))
    assert np.array_equal(f.read("many_chunks"), expected["sq_coords"])
    assert np.array_equal(f.read("many_chunks_z"), expected["sq_cells"])
    assert len(f.keys("/wide")) == 300
    for i in (0, 7, 8, 150, 299):
        assert np.array_equal(f.read(f"/wide/d{i:03d}"), [i, 2 * i])


def test_hdf5_decoder_refuses_what_it_cannot_decode(tmp_path):
    from cgx_hip import hdf5_min
    with pytest.raises(hdf5_min.Hdf5FormatError, match="layout message version 4"):
        hdf5_min.Hdf5File(os.path.join(G, "latest.h5")).read("x")
    bad = tmp_path / "not.h5"
    bad.write_bytes(b"hello" * 300)
    with pytest.raises(hdf5_min.Hdf5FormatError, match="not an HDF5 file"):
        hdf5_min.Hdf5File(bad)


def _as_sets(rows):
    return {tuple(sorted(r)) for r in np.asarray(rows).tolist()}


def test_xdmf_reader_dolfinx_layout(expected):
    """grids "mesh" + "ct" / "ft" (tag rule mixed_dim_problem.py:137-141); the tag grids list their entities in another order"""
    from cgx_hip import xdmf
    coords, cells, ct, ft = xdmf.read_mesh_and_tags(os.path.join(G, "square8.xdmf"), os.path.join(G, "square8_facets.xdmf"))
    assert coords.shape == (81, 2) and cells.dtype == np.int32 and ct.dtype == np.int32
    assert np.array_equal(coords, expected["sq_coords"]) and np.array_equal(cells, expected["sq_cells"])
    assert np.array_equal(ct, expected["sq_ct"])                  # values matched back to the mesh's own cell order
    want = dict(zip(map(tuple, expected["sq_fv"].tolist()), expected["sq_fval"].tolist()))
    got = dict(zip((tuple(sorted(r)) for r in ft[0].tolist()), ft[1].tolist()))
    assert got == want


def test_xdmf_reader_meshio_layout_and_inline_xml(expected):
    """one grid "mesh" carrying the tags as an Attribute (mixed_dim_problem.py:142-145); the facet values as inline XML"""
    from cgx_hip import xdmf
    coords, cells, ct, ft = xdmf.read_mesh_and_tags(os.path.join(G, "cube3_mesh.xdmf"), os.path.join(G, "cube3_facets.xdmf"))
    assert np.array_equal(coords, expected["cu_coords"]) and np.array_equal(cells, expected["cu_cells"])
    assert np.array_equal(ct, expected["cu_ct"])
    assert np.array_equal(ft[0], expected["cu_fv"]) and np.array_equal(ft[1], expected["cu_fval"])


def test_xdmf_reader_accepts_a_single_grid_of_any_name(expected, tmp_path):
    """meshio names its grid "Grid": a file with one grid that carries the tags is read as 'tags under the mesh hierarchy'"""
    import shutil
    from cgx_hip import xdmf
    for stem in ("cube3_mesh", "cube3_facets"):
        shutil.copy(os.path.join(G, stem + ".h5"), tmp_path / (stem + ".h5"))
        text = open(os.path.join(G, stem + ".xdmf")).read().replace('Grid Name="mesh"', 'Grid Name="Grid"')
        (tmp_path / (stem + ".xdmf")).write_text(text)
    coords, cells, ct, ft = xdmf.read_mesh_and_tags(str(tmp_path / "cube3_mesh.xdmf"), str(tmp_path / "cube3_facets.xdmf"))
    assert np.array_equal(cells, expected["cu_cells"]) and np.array_equal(ct, expected["cu_ct"])
    assert np.array_equal(ft[0], expected["cu_fv"]) and np.array_equal(ft[1], expected["cu_fval"])


def test_load_mesh_from_xdmf_feeds_the_same_membrane_as_the_arrays(expected):
    """``mesh.load_mesh`` on the XDMF pair gives the arrays, scaled, and the membrane facets derived from the file's facet tags
    (value 4) are the intra/extra interface the generator marks on the same mesh"""
    from cgx_hip import mesh as M
    coords, cells, ct, ft, desc = M.load_mesh(os.path.join(G, "square8.xdmf"), os.path.join(G, "square8_facets.xdmf"), 1e-6)
    assert "XDMF" in desc and np.allclose(coords, 1e-6 * expected["sq_coords"], rtol=0, atol=0)
    gam, gtags, gverts = M.gamma_integration_entities(cells, ct, (1,), (2,), ft)
    gam0, gtags0, gverts0 = M.gamma_integration_entities(cells, ct, (1,), (2,), None)
    assert np.array_equal(gam, gam0) and np.array_equal(gtags, gtags0) and (gtags == 4).all() and len(gam) == 16
    # '+' side intracellular
    assert (ct[gam[:, 0]] == 1).all() and (ct[gam[:, 2]] == 2).all()
    with pytest.raises(RuntimeError, match="does not exist"):
        M.load_mesh(os.path.join(G, "nowhere.xdmf"), os.path.join(G, "nowhere.xdmf"))


def test_xdmf_errors_name_the_problem(tmp_path):
    from cgx_hip import xdmf
    p = tmp_path / "m.xdmf"
    p.write_text('<Xdmf><Domain><Grid Name="mesh"><Topology TopologyType="Hexahedron"><DataItem Format="XML" Dimensions="1 8">0 1 2 3 4 5 6 7</DataItem></Topology></Grid></Domain></Xdmf>')
    with pytest.raises(xdmf.XdmfError, match="not supported"):
        xdmf.read_mesh_and_tags(str(p), str(p))
    p.write_text('<Xdmf><Domain><Grid Name="mesh"><Topology TopologyType="Triangle"><DataItem Format="HDF" Dimensions="1 3">gone.h5:/x</DataItem></Topology></Grid></Domain></Xdmf>')
    with pytest.raises(xdmf.XdmfError, match="not found"):
        xdmf.read_mesh_and_tags(str(p), str(p))
    p.write_text("this is not xml")
    with pytest.raises(xdmf.XdmfError, match="not an XDMF"):
        xdmf.read_mesh_and_tags(str(p), str(p))
