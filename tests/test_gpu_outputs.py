"""SURVEY 8 f4 without an I/O library: probe-point evaluation, membrane traces, .npy exports and .npz checkpoints
(reference KNPEMIx_solver.py:551-643, 799-821, 833-866), driven through the reference's command-line interface
(main.py --config <yaml>)."""
import os

import numpy as np
import pytest
import yaml

from parity_utils import ci_config, run_oracle

pytestmark = pytest.mark.gpu


def test_point_evaluation_traces_and_checkpoints_match_oracle(tmp_path):
    from CGx.KNPEMI import main as cli
    N, steps = 16, 3
    out_dir = str(tmp_path) + os.sep
    cfg = ci_config(N=N, steps=steps, rtol=1e-12)
    cfg["quiet"] = True
    cfg["output_dir"] = out_dir
    # mesh units (the config multiplies by mesh_conversion_factor): two vertices and one cell-interior point per side,
    # one membrane vertex and one mid-edge point of the membrane
    h = 1.0 / N
    cfg["point_evaluation"] = {"ics_points": [[0.5, 0.5], [0.5 + h / 3, 0.5 + h / 4]],
                               "ecs_points": [[0.125, 0.125], [0.125 + h / 3, 0.125 + h / 4]],
                               "gamma_points": [[0.25, 0.5], [0.25, 0.5 + h / 2]]}
    cfg["solver"]["output"].update({"save_dat": True, "save_cpoints": True, "save_pngs": True, "save_interval": 2})
    yml = tmp_path / "square_probe.yaml"
    yml.write_text(yaml.safe_dump(cfg))
    solver = cli.main(["--config", str(yml), "--view", "0"])
    assert all(r > 0 for r in solver.reasons)

    # oracle: nodal values after every step (sparse LU), P1-interpolated by hand
    import knpemi_oracle as K
    from parity_utils import make_oracle
    o = make_oracle(N)
    X = o.coords / o.coords.max()
    snaps = []

    def log(step, oo, x):
        snaps.append(([[oo.k[s][j].copy() for j in range(3)] + [oo.phi[s].copy()] for s in range(2)], oo.phi_m.copy(),
                      oo.n.copy(), oo.m.copy(), oo.h.copy()))
    o.run(steps, solver="lu_gauge", log=log)

    def interp(field, pt, side):
        """P1 value at pt from nodal values: vertex -> nodal value; otherwise barycentric in a containing cell of `side`"""
        from cgx_hip.output import _barycentric
        cells = o.cells if side is None else o.cells[o.cell_side == side]
        c, w = _barycentric(X, cells, np.array([pt]))
        return float((field[cells[c[0]]] * w[0]).sum())

    ics = np.load(out_dir + "ics_point_values.npy")
    ecs = np.load(out_dir + "ecs_point_values.npy")
    gam = np.load(out_dir + "gamma_point_values.npy")
    assert ics.shape == (steps + 1, 4, 2) and ecs.shape == (steps + 1, 4, 2) and gam.shape == (steps + 1, 2)
    for i in range(1, steps + 1):
        fields, phim = snaps[i - 1][0], snaps[i - 1][1]
        for j in range(4):
            # concentrations: 1e-7 of their own scale (stopping rule); potentials: 1e-6 of the potential scale (phi_i ~ 0.07 V)
            tol_i = 1e-7 * np.abs(fields[0][j]).max() if j < 3 else 1e-6 * 0.07
            tol_e = 1e-7 * np.abs(fields[1][j]).max() if j < 3 else 1e-6 * 0.07
            for k, pt in enumerate(cfg["point_evaluation"]["ics_points"]):
                assert abs(ics[i, j, k] - interp(fields[0][j], pt, 0)) <= tol_i, (i, j, k)
            for k, pt in enumerate(cfg["point_evaluation"]["ecs_points"]):
                assert abs(ecs[i, j, k] - interp(fields[1][j], pt, 1)) <= tol_e, (i, j, k)
        for k, pt in enumerate(cfg["point_evaluation"]["gamma_points"]):
            assert abs(gam[i, k] - interp(phim, pt, None)) <= 1e-6 * np.abs(phim).max(), (i, k)
    # initial data row
    assert np.allclose(ics[0, 0], 12.0) and np.allclose(ecs[0, 1], 4.0) and np.allclose(gam[0], -0.070)

    # membrane trace at the measurement vertex (closest membrane vertex to the mesh centre) + gating variables there
    p = solver.problem
    v = np.load(out_dir + "phi_m.npy")
    assert v.shape == (steps + 1,)
    gv = np.nonzero((o.lay.node_i >= 0) & (o.lay.node_e >= 0))[0]
    d2 = ((X[gv] - 0.5) ** 2).sum(axis=1)
    assert np.isclose(d2[list(gv).index(p.png_dof)], d2.min())
    for i in range(1, steps + 1):
        assert abs(v[i] - 1000.0 * snaps[i - 1][1][p.png_dof]) <= 1e-6 * 1000.0 * 0.07
    for nm, idx in (("n", 2), ("m", 3), ("h", 4)):
        tr = np.load(out_dir + nm + ".npy")
        assert np.allclose(tr[1:], [snaps[i][idx][p.png_dof] for i in range(steps)], rtol=1e-9)
    assert np.load(out_dir + "iterations.npy").shape == (steps,)
    assert np.load(out_dir + "assembly_time.npy").shape == (steps,) and np.load(out_dir + "solve_time.npy").shape == (steps,)

    # checkpoints at steps 0 and 2 (save_interval 2): nodal fields of the owned vertices
    files = sorted(os.listdir(out_dir + "checkpoints"))
    assert files == ["step_000000_rank0.npz", "step_000002_rank0.npz"]
    ck = np.load(out_dir + "checkpoints/step_000002_rank0.npz")
    assert int(ck["step"]) == 2 and np.isclose(float(ck["t"]), 2 * cfg["dt"])
    vi = o.lay.node_i >= 0
    assert np.allclose(ck["Na_i"][vi], snaps[1][0][0][0][vi], rtol=1e-7)
    assert np.allclose(ck["phi_m"][gv], snaps[1][1][gv], rtol=1e-6)


def test_save_xdmf_writes_mesh_tags_and_time_series(tmp_path):
    """``save_xdmf: True`` (reference KNPEMIx_solver.py:766-797): subdomains.xdmf and solution.xdmf + .h5, written without an
    HDF5 library; read back through the package's XDMF/HDF5 reader: the mesh and tags are the run's, the last time level holds
    the solver's final fields, the first one the initial state."""
    from CGx.KNPEMI import main as cli
    from cgx_hip import hdf5_min, xdmf
    import xml.etree.ElementTree as ET
    N, steps = 8, 4
    out_dir = str(tmp_path) + os.sep
    cfg = ci_config(N=N, steps=steps, rtol=1e-10)
    cfg["quiet"] = True
    cfg["output_dir"] = out_dir
    cfg["solver"]["output"].update({"save_xdmf": True, "save_interval": 2})
    yml = tmp_path / "square_xdmf.yaml"
    yml.write_text(yaml.safe_dump(cfg))
    solver = cli.main(["--config", str(yml), "--view", "0"])
    p = solver.problem
    lm = p.local_mesh
    # the tag file is a mesh file the reader (and the reference's read_mesh / read_meshtags) accepts
    coords, cells, ct, _ = xdmf.read_mesh_and_tags(out_dir + "subdomains.xdmf", out_dir + "subdomains.xdmf")
    assert np.array_equal(coords, lm.coords) and np.array_equal(cells, lm.cells) and np.array_equal(ct, lm.cell_tags)
    # time series: steps 0, 2, 4
    root = ET.parse(out_dir + "solution.xdmf").getroot()
    times = [float(t.get("Value")) for t in root.iter("Time")]
    assert np.allclose(times, [0.0, 2 * cfg["dt"], 4 * cfg["dt"]], rtol=1e-12, atol=0)
    names = [a.get("Name") for a in root.iter("Attribute") if a.get("Center") == "Node"]
    assert set(names) == {"Na_i", "Na_e", "K_i", "K_e", "Cl_i", "Cl_e", "phi_i", "phi_e"} and len(names) == 3 * 8
    h = hdf5_min.Hdf5File(out_dir + "solution.h5")
    assert h.keys("/Function/phi_i") == ["0", "1", "2"]
    for side in range(2):
        for idx in range(4):
            f = p.wh[side][idx]
            assert np.array_equal(h.read(f"/Function/{f.name}/2").ravel(), f.numpy())
    vi = p.local_mesh.cells[p.cell_side == 0]
    assert np.allclose(h.read("/Function/Na_i/0").ravel()[np.unique(vi)], 12.0) and np.allclose(h.read("/Function/K_e/0").ravel()[np.unique(p.local_mesh.cells[p.cell_side == 1])], 4.0)
