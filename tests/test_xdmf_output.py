"""The HDF5 writer behind ``save_xdmf`` (cgx_hip/hdf5_write.py; reference output: KNPEMIx_solver.py:766-797 through
dolfinx.io.XDMFFile).  Files are read back with the package's own decoder and -- when the image's conda interpreter with
h5py is present -- with libhdf5 itself."""
import os
import subprocess
import sys

import numpy as np
import pytest

CONDA_PY = "/opt/conda/bin/python3.9"


def _sample():
    rng = np.random.default_rng(11)
    data = {"/Mesh/mesh/geometry": rng.standard_normal((81, 2)), "/Mesh/mesh/topology": rng.integers(0, 81, (128, 3)),
            "/MeshTags/ct/Values": rng.integers(0, 5, (128, 1)).astype(np.int32), "/f32": rng.standard_normal(7).astype(np.float32),
            "/u8": np.arange(5, dtype=np.uint8), "/scalar_like": np.array([3.5])}
    for k in range(300):                               # a group whose B-tree needs two levels
        data[f"/Function/phi_i/{k}"] = rng.standard_normal((10, 1))
    return data


def test_writer_round_trip_through_the_own_decoder(tmp_path):
    from cgx_hip import hdf5_min, hdf5_write
    data = _sample()
    path = tmp_path / "w.h5"
    with hdf5_write.Hdf5Writer(path) as w:
        for k, v in data.items():
            w.write(k, v)
        with pytest.raises(ValueError):
            w.write("/u8", np.zeros(3))                # exists
        with pytest.raises(ValueError):
            w.write("/c", np.zeros(3, dtype=np.complex128))
    f = hdf5_min.Hdf5File(path)
    assert f.keys("/") == ["Function", "Mesh", "MeshTags", "f32", "scalar_like", "u8"] and len(f.keys("/Function/phi_i")) == 300
    for k, v in data.items():
        a = f.read(k)
        assert a.dtype == v.dtype and np.array_equal(a, v), k


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="no HDF5 library in this environment (image's conda h5py not present)")
def test_libhdf5_reads_the_writers_files(tmp_path):
    """h5py 3.3 on libhdf5 1.10.6 (a separate interpreter of the image) opens the file and finds the same arrays"""
    from cgx_hip import hdf5_write
    data = _sample()
    path = tmp_path / "w.h5"
    with hdf5_write.Hdf5Writer(path) as w:
        for k, v in data.items():
            w.write(k, v)
    np.savez(tmp_path / "expected.npz", **{k.replace("/", "|"): v for k, v in data.items()})
    code = ("import h5py, numpy as np, sys\n"
            "E = np.load(sys.argv[2]); h = h5py.File(sys.argv[1], 'r')\n"
            "bad = [k for k in E.files if not (np.array_equal(h[k.replace('|', '/')][...], E[k]) and h[k.replace('|', '/')].dtype == E[k].dtype)]\n"
            "print('BAD' if bad else 'OK', len(E.files), len(h['/Function/phi_i']), bad[:3])\n")
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    r = subprocess.run([CONDA_PY, "-c", code, str(path), str(tmp_path / "expected.npz")], capture_output=True, text=True, timeout=120, env=env)
    if r.returncode != 0 and "No module named" in r.stderr:
        pytest.skip("conda interpreter without h5py")
    assert r.returncode == 0, r.stderr[-800:]
    assert r.stdout.split()[:3] == ["OK", str(len(data)), "300"], r.stdout


def test_solution_xdmf_is_a_complete_document_after_every_save(tmp_path):
    """solution.xdmf is appended to (O(steps) text) and parses as XML after every save; a run that stops early and closes
    the writer in its ``finally`` leaves a readable .h5 with the steps written so far."""
    import xml.etree.ElementTree as ET
    from types import SimpleNamespace as NS
    from cgx_hip import hdf5_min
    from cgx_hip.output import RunOutput

    class F:
        def __init__(self, name, n):
            self.name, self.v = name, np.zeros(n)

        def numpy(self):
            return self.v
    n_pts = 9
    coords = np.random.default_rng(0).random((n_pts, 2))
    cells = np.array([[0, 1, 2], [2, 3, 4], [4, 5, 6], [6, 7, 8]])
    p = NS(local_mesh=NS(coords=coords, cells=cells, cell_tags=np.array([1, 1, 2, 2])), comm=NS(rank=0, size=1), num_variables=2,
           wh=[[F("Na_i", n_pts), F("phi_i", n_pts)], [F("Na_e", n_pts), F("phi_e", n_pts)]], t=NS(value=0.0), print=lambda *a, **k: None)
    out = RunOutput.__new__(RunOutput)
    out.p, out.prefix, out.xdmf = p, str(tmp_path) + os.sep, None
    out.init_xdmf_savefile()                              # step 0
    sizes = []
    for k in range(1, 6):
        p.t.value = 0.1 * k
        for side in p.wh:
            for f in side:
                f.v[:] = k
        out.save_xdmf()
        root = ET.parse(tmp_path / "solution.xdmf").getroot()
        steps = root.findall(".//Grid[@CollectionType='Temporal']/Grid")
        assert len(steps) == k + 1 and float(steps[-1].find("Time").get("Value")) == pytest.approx(0.1 * k)
        sizes.append(os.path.getsize(tmp_path / "solution.xdmf"))
    assert len(set(np.diff(sizes))) <= 2                  # constant growth per step (the time value's repr may vary by a character or two)
    out.close_xdmf()                                      # what SolverKNPEMI.solve does in its finally block
    out.close_xdmf()                                      # idempotent
    f = hdf5_min.Hdf5File(tmp_path / "solution.h5")
    assert len(f.keys("/Function/phi_i")) == 6 and np.all(f.read("/Function/phi_e/5") == 5.0)
    assert len(ET.parse(tmp_path / "solution.xdmf").getroot().findall(".//Grid[@CollectionType='Temporal']/Grid")) == 6
