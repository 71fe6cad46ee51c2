"""Tissue surrogate: a lattice of cells, one tag per cell, membrane tag = cell tag, the
same mechanism list on every cell (one shared membrane program), stimulus restricted to an x-range.
HIP path vs the oracle's sparse-LU run on the same mesh."""
import numpy as np
import pytest

from parity_utils import make_problem, tissue_config

pytestmark = pytest.mark.gpu


def _oracle(problem, cfg, models, steps):
    import knpemi_oracle as K
    lm = problem.local_mesh
    tags = tuple(cfg["ics_tags"])
    mods = {"passive": [K.Model("passive", tags)],
            "ci": [K.Model("neuronal_ct", tags), K.Model("hh", tags), K.Model("atp", tags)]}[models]
    kw = {}
    if "stimulus_region" in cfg:
        lo, hi = cfg["stimulus_region"]["range"]
        kw = dict(stimulus_tags=tags, stimulus_region=(0, lo * 1e-6, hi * 1e-6))
    o = K.OracleKNPEMI(lm.coords, lm.cells, lm.cell_tags, intra_tags=tags, extra_tag=1, gamma=lm.gamma, gamma_tag=lm.gamma_tags,
                       models=mods, mesh_conversion_factor=1.0, **kw)
    o.run(steps, solver="lu_gauge")
    return o


@pytest.mark.parametrize("dim,N,m,models,pc", [(2, 16, 2, "ci", "hypre"), (2, 24, 3, "passive", "btcc"),
                                               (3, 8, 2, "ci", "btcc"), (3, 8, 2, "passive", "hypre")])
def test_tissue_lattice_matches_oracle(dim, N, m, models, pc):
    from CGx.KNPEMI.KNPEMIx_solver import SolverKNPEMI
    cfg = tissue_config(dim, N, m, steps=2, rtol=1e-12, pc=pc, stimulus=(models == "ci"))
    p = make_problem(cfg, models)
    assert len(p.gamma_tags) == m ** dim
    assert len(p.programs) == 1, "identical mechanism lists must share one membrane program"
    p.solver_config["view_ksp"] = False
    s = SolverKNPEMI(p, solver_config=p.solver_config)
    s.solve()
    assert all(r > 0 for r in s.reasons)
    o = _oracle(p, cfg, models, 2)
    if models == "ci":
        assert abs(p.stimulus_area - o.stimulus_area) <= 1e-12 * o.stimulus_area
    gam = (o.lay.node_i >= 0) & (o.lay.node_e >= 0)
    assert np.allclose(s.problem.phi_m_prev.numpy()[gam], o.phi_m[gam], rtol=1e-6)
    ni, ne = s.potential_norms()
    oi, oe = o.potential_norms()
    assert abs(ni - oi) <= 1e-6 * oi
    for j in range(3):
        vi = o.lay.node_i >= 0
        assert np.allclose(s.problem.wh[0][j].numpy()[vi], o.k[0][j][vi], rtol=1e-7)


@pytest.mark.parametrize("dim,N,m", [(2, 20, 2), (3, 10, 2)])
def test_ion_injection_source_matches_oracle(dim, N, m):
    """``source_terms: ion_injection`` of the reference's tissue configs (configs/20m/*.yaml): K and Cl injected into
    the extracellular space around the mesh centre."""
    from CGx.KNPEMI.KNPEMIx_solver import SolverKNPEMI
    # rtol 1e-14 + btcc: the stopping rule is relative to the preconditioned right-hand side, which the concentrations
    # dominate; at 1e-12 the potentials still carry ~1e-7 V of solver error on these small lattices
    cfg = tissue_config(dim, N, m, steps=2, rtol=1e-14, pc="btcc", stimulus=False)
    cfg["source_terms"] = "ion_injection"
    p = make_problem(cfg, "passive")
    assert len(p.injection_cells) > 0
    p.solver_config["view_ksp"] = False
    s = SolverKNPEMI(p, solver_config=p.solver_config)
    s.solve()
    import knpemi_oracle as K
    lm = p.local_mesh
    tags = tuple(cfg["ics_tags"])
    o = K.OracleKNPEMI(lm.coords, lm.cells, lm.cell_tags, intra_tags=tags, extra_tag=1, gamma=lm.gamma, gamma_tag=lm.gamma_tags,
                       models=[K.Model("passive", tags)], mesh_conversion_factor=1.0)
    o.set_ion_injection()
    assert np.array_equal(np.sort(o.injection_cells), np.sort(p.injection_cells))
    assert abs(o.injection_volume - p.injection_volume) <= 1e-12 * o.injection_volume
    o.run(2, solver="lu_gauge")
    # the source really acts: more extracellular K where it is injected than in the same run without it
    ve = o.lay.node_e >= 0
    o0 = K.OracleKNPEMI(lm.coords, lm.cells, lm.cell_tags, intra_tags=tags, extra_tag=1, gamma=lm.gamma, gamma_tag=lm.gamma_tags,
                        models=[K.Model("passive", tags)], mesh_conversion_factor=1.0)
    o0.run(2, solver="lu_gauge")
    inj = np.unique(lm.cells[o.injection_cells])
    inj = inj[o.lay.node_e[inj] >= 0]
    assert (o.k[1][1][inj] - o0.k[1][1][inj]).min() > 1e-7
    for j in range(3):
        assert np.allclose(s.problem.wh[1][j].numpy()[ve], o.k[1][j][ve], rtol=1e-8)
    gam = (o.lay.node_i >= 0) & (o.lay.node_e >= 0)
    assert np.allclose(s.problem.phi_m_prev.numpy()[gam], o.phi_m[gam], rtol=1e-6)
    ni, ne = s.potential_norms()
    oi, oe = o.potential_norms()
    assert abs(ni - oi) <= 1e-6 * oi
