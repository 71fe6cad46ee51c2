"""Several membrane tags with different mechanism sets (the structure of the reference's tissue configs,
src/CGx/KNPEMI/main.py:32-39): neuron = HH + ATP pump + neuronal cotransporters with a stimulus restricted to a
region, glial cell = Kir/NaK pump + glial cotransporters, cell-specific initial conditions.  HIP path vs oracle."""
import numpy as np
import pytest
import torch

from parity_utils import two_cell_config, two_cell_mesh

pytestmark = pytest.mark.gpu


def _models(p):
    from CGx.KNPEMI.KNPEMIx_ionic_model import (ATPPump, GlialCotransporters, HodgkinHuxley, KirNaKPumpModel,
                                                NeuronalCotransporters)
    # same order as the reference's main.py:40 for glia configs
    return [HodgkinHuxley(p, tags=p.neuron_tags), ATPPump(p, tags=p.neuron_tags), NeuronalCotransporters(p, tags=p.neuron_tags),
            GlialCotransporters(p, tags=p.glia_tags), KirNaKPumpModel(p, tags=p.glia_tags)]


def _oracle_from(problem, coords, cells, tags, gamma_tags_by_facet, region=(1, 0.3e-6, 0.6e-6)):
    import knpemi_oracle as K
    lm = problem.local_mesh
    params = K.Params(ki_init=(12.0, 130.0, 5.0), ke_init=(140.0, 4.0, 125.0))
    models = [K.Model("hh", (2,)), K.Model("atp", (2,)), K.Model("neuronal_ct", (2,)), K.Model("glial_ct", (3,)), K.Model("kir_nak", (3,))]
    o = K.OracleKNPEMI(lm.coords, lm.cells, lm.cell_tags, intra_tags=(2, 3), extra_tag=1, gamma=lm.gamma, gamma_tag=lm.gamma_tags,
                       params=params, models=models, stimulus_tags=(2,), mesh_conversion_factor=1.0,
                       stimulus_region=region)
    # cell-specific initial state copied from the product (KNPEMIx_problem.py:396-441)
    for side in range(2):
        for j in range(3):
            o.k[side][j] = problem.wh[side][j].numpy().copy()
        o.phi[side] = problem.wh[side][3].numpy().copy()
    o.phi_m = problem.phi_m_prev.numpy().copy()
    return o


def test_two_cells_two_programs(tmp_path):
    from CGx.KNPEMI.KNPEMIx_problem import ProblemKNPEMI
    from CGx.KNPEMI.KNPEMIx_solver import SolverKNPEMI
    coords, cells, tags, fverts, ftags = two_cell_mesh(16)
    path = str(tmp_path / "twocells.npz")
    np.savez(path, coords=coords, cells=cells, cell_tags=tags, facets=fverts, facet_tags=ftags)
    cfg = two_cell_config(path)
    p = ProblemKNPEMI(cfg)
    assert p.glia_flag and p.gamma_tags == (2, 3) and p.stimulus_region
    models = _models(p)
    p.set_initial_conditions()
    p.init_ionic_models(models)
    p.setup_variational_form()
    assert len(p.programs) == 2
    o = _oracle_from(p, coords, cells, tags, ftags)
    assert abs(p.stimulus_area - o.stimulus_area) <= 1e-12 * o.stimulus_area
    # one assembly against the oracle
    be = p.create_backend()
    p.t.value = float(p.dt.value)
    o.t = o.p.dt
    o.update_t_mod()
    for m in models:
        if hasattr(m, "update_t_mod"):
            m.update_t_mod()
    be.assemble_matrix()
    be.assemble_rhs()
    A, Ao = be.csr(), o.assemble_A()
    D = (A - Ao).tocoo()
    assert np.abs(D.data).max() <= 1e-12 * np.abs(Ao.data).max()
    b, bo = be.b.cpu().numpy(), o.assemble_b()
    for f in range(4):
        assert np.max(np.abs(b[f::4] - bo[f::4])) <= 1e-10 * np.max(np.abs(bo[f::4])), f
    # the two programs really differ: the glial membrane carries no HH/stimulus current
    assert not np.array_equal(p.programs[0].code, p.programs[1].code)
    # full run
    p.t.value = 0.0
    p.solver_config["view_ksp"] = False
    s = SolverKNPEMI(p, solver_config=p.solver_config)
    s.solve()
    o.t = 0.0
    o2 = _oracle_from_fresh(cfg, coords, cells, tags, ftags)
    o2.run(2, solver="lu_gauge")
    gam = (o2.lay.node_i >= 0) & (o2.lay.node_e >= 0)
    assert np.allclose(s.problem.phi_m_prev.numpy()[gam], o2.phi_m[gam], rtol=1e-6)
    ni, ne = s.potential_norms()
    oi, oe = o2.potential_norms()
    assert abs(ni - oi) <= 1e-6 * oi


def _oracle_from_fresh(cfg, coords, cells, tags, ftags, region=(1, 0.3e-6, 0.6e-6)):
    """oracle with the cell-specific ICs of the config, built without the product's solver state"""
    from CGx.KNPEMI.KNPEMIx_problem import ProblemKNPEMI
    p = ProblemKNPEMI(cfg)
    _ = _models(p)
    p.set_initial_conditions()
    return _oracle_from(p, coords, cells, tags, ftags, region)


def test_stimulus_restricted_in_several_directions(tmp_path):
    """``stimulus_region: {multiple: True, direction: [x, y], range: [[..], [..]]}`` -- the product of one mask per direction
    (reference KNPEMIx_ionic_model.py:573-586, mixed_dim_problem.py:346-351): stimulus area, right-hand side of the first step and
    the solution after two steps against the oracle (sparse LU)."""
    from CGx.KNPEMI.KNPEMIx_problem import ProblemKNPEMI
    from CGx.KNPEMI.KNPEMIx_solver import SolverKNPEMI
    coords, cells, tags, fverts, ftags = two_cell_mesh(16)
    path = str(tmp_path / "twocells.npz")
    np.savez(path, coords=coords, cells=cells, cell_tags=tags, facets=fverts, facet_tags=ftags)
    cfg = two_cell_config(path)
    cfg["stimulus_region"] = {"multiple": True, "direction": ["x", "y"], "range": [[0.1, 0.3], [0.3, 0.6]]}
    region = [(0, 0.1e-6, 0.3e-6), (1, 0.3e-6, 0.6e-6)]
    p = ProblemKNPEMI(cfg)
    assert p.multiple_stimulus_directions and p.stimulus_region_directions == [0, 1]
    models = _models(p)
    p.set_initial_conditions()
    p.init_ionic_models(models)
    p.setup_variational_form()
    o = _oracle_from(p, coords, cells, tags, ftags, region)
    o1 = _oracle_from(p, coords, cells, tags, ftags)              # the one-direction region of the other test: a different area
    assert abs(p.stimulus_area - o.stimulus_area) <= 1e-12 * o.stimulus_area and o.stimulus_area < 0.9 * o1.stimulus_area
    be = p.create_backend()
    p.t.value = float(p.dt.value)
    o.t = o.p.dt
    o.update_t_mod()
    for m in models:
        if hasattr(m, "update_t_mod"):
            m.update_t_mod()
    be.assemble_rhs()
    b, bo = be.b.cpu().numpy(), o.assemble_b()
    for f in range(4):
        assert np.max(np.abs(b[f::4] - bo[f::4])) <= 1e-10 * np.max(np.abs(bo[f::4])), f
    p.t.value = 0.0
    p.solver_config["view_ksp"] = False
    s = SolverKNPEMI(p, solver_config=p.solver_config)
    s.solve()
    o2 = _oracle_from_fresh(cfg, coords, cells, tags, ftags, region)
    o2.run(2, solver="lu_gauge")
    gam = (o2.lay.node_i >= 0) & (o2.lay.node_e >= 0)
    assert np.allclose(s.problem.phi_m_prev.numpy()[gam], o2.phi_m[gam], rtol=1e-6)
    oi, oe = o2.potential_norms()
    assert abs(s.potential_norms()[0] - oi) <= 1e-6 * oi
