"""Shared helpers of the parity tests: the CI physics as a config dict for the native path and as
an oracle instance (test infrastructure; the oracle is only ever the checker)."""
from __future__ import annotations

import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "knp-emi-cgx_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

# physics of reference src/CGx/KNPEMI/configs/tests/electric_potential_norms_iterative_solver.yaml
CI_BASE = {
    "problem_type": "KNP-EMI",
    "quiet": True,
    "dt": 0.000025,
    "time_steps": 10,
    "physical_constants": {"T": 300, "F": 96485, "R": 8.314},
    "C_M": 0.02,
    "cell_tag_file": "square32.xdmf",
    "facet_tag_file": "square32_facets.xdmf",
    "ics_tags": [1], "ecs_tags": [2], "boundary_tags": [3], "membrane_tags": [4],
    "mesh_conversion_factor": 1e-6,
    "initial_conditions": {"phi_m": -0.070, "Na_i": 12, "Na_e": 140, "K_i": 130, "K_e": 4, "Cl_i": 5, "Cl_e": 125,
                           "n": 0.276, "m": 0.0379, "h": 0.688},
    "stimulus": {"conductance": {"g_syn_bar": 1e-9}, "a_syn": 5e-4, "T_stim": 1.0, "scale": True},
    "solver": {"direct": False,
               "ksp_settings": {"strong_threshold": 0.5, "ksp_rtol": 1e-9, "ksp_type": "gmres", "pc_type": "hypre",
                                "norm_type": "preconditioned", "non_zero_init_guess": True},
               "output": {"save_xdmf": False, "save_cpoints": False, "save_pngs": False, "save_dat": False}},
}


def ci_config(N=32, steps=10, rtol=1e-9, pc="hypre", kind="square", direct=False):
    cfg = copy.deepcopy(CI_BASE)
    cfg["time_steps"] = steps
    cfg["cell_tag_file"] = f"{kind}{N}.xdmf"
    cfg["facet_tag_file"] = f"{kind}{N}_facets.xdmf"
    cfg["solver"]["direct"] = direct
    cfg["solver"]["ksp_settings"]["ksp_rtol"] = rtol
    cfg["solver"]["ksp_settings"]["pc_type"] = pc
    return cfg


def make_problem(cfg, models="ci", local_mesh=None):
    """Construction order of the reference's test scripts (tests/KNPEMI/electric_potential_norms_*.py:27-36)."""
    from CGx.KNPEMI.KNPEMIx_ionic_model import ATPPump, HodgkinHuxley, NeuronalCotransporters, PassiveModel
    from CGx.KNPEMI.KNPEMIx_problem import ProblemKNPEMI
    problem = ProblemKNPEMI(cfg, local_mesh=local_mesh)
    if models == "ci":
        ionic_models = [NeuronalCotransporters(problem), HodgkinHuxley(problem), ATPPump(problem)]
    elif models == "passive":
        ionic_models = [PassiveModel(problem)]
    else:
        ionic_models = models(problem)
    problem.set_initial_conditions()
    problem.init_ionic_models(ionic_models)
    problem.setup_variational_form()
    return problem


def run_native(cfg, models="ci", local_mesh=None):
    from CGx.KNPEMI.KNPEMIx_solver import SolverKNPEMI
    problem = make_problem(cfg, models, local_mesh)
    problem.solver_config["view_ksp"] = False
    solver = SolverKNPEMI(problem, solver_config=problem.solver_config)
    solver.solve()
    return solver


def make_oracle(N=32, kind="square", models="ci"):
    import knpemi_oracle as K
    mk = K.make_square if kind == "square" else K.make_cube
    mdl = K.CI_MODELS() if models == "ci" else [K.Model("passive", (4,))]
    return mk(N, models=mdl)


def run_oracle(N=32, steps=10, kind="square", models="ci", solver="lu_gauge"):
    o = make_oracle(N, kind, models)
    o.run(steps, solver=solver)
    return o


def two_cell_mesh(N=16):
    """Unit square with two inclusions: a 'neuron' (tag 2) and a 'glial cell' (tag 3) in extracellular space (tag 1);
    membrane facets carry the tag of their cell (the convention of the reference's tissue configs:
    'membrane tag = intra tag', mixed_dim_problem.py:403-407)."""
    import numpy as np
    from cgx_hip import mesh as meshmod
    coords, cells = meshmod.create_unit_square(N)
    cx = coords[cells].mean(axis=1)
    allv = lambda lo, hi: ((coords[cells] >= lo) & (coords[cells] <= hi)).all(axis=(1, 2))
    tags = np.full(cells.shape[0], 1, dtype=np.int32)
    box = lambda x0, x1, y0, y1: ((coords[cells][:, :, 0] >= x0) & (coords[cells][:, :, 0] <= x1) &
                                  (coords[cells][:, :, 1] >= y0) & (coords[cells][:, :, 1] <= y1)).all(axis=1)
    tags[box(0.125, 0.4375, 0.25, 0.75)] = 2
    tags[box(0.5625, 0.875, 0.25, 0.75)] = 3
    gamma, _, fverts = meshmod.gamma_integration_entities(cells, tags, (2, 3), (1,))
    ftags = tags[gamma[:, 0]]
    return coords, cells, tags, fverts, ftags


def two_cell_config(path, steps=2, rtol=1e-11, pc="hypre"):
    import copy
    cfg = copy.deepcopy(CI_BASE)
    cfg.update({"time_steps": steps, "cell_tag_file": path, "facet_tag_file": path, "input_dir": "",
                "ics_tags": [2, 3], "ecs_tags": [1], "membrane_tags": [2, 3], "glia_tags": [3], "stimulus_tags": [2],
                "stimulus_region": {"direction": "y", "range": [0.3, 0.6]},
                "initial_conditions": {"phi_m_n": -0.070, "phi_m_g": -0.082, "Na_i_n": 12, "Na_i_g": 15, "Na_e": 140,
                                       "K_i_n": 130, "K_i_g": 100, "K_e": 4, "Cl_i_n": 5, "Cl_i_g": 6, "Cl_e": 125,
                                       "n": 0.276, "m": 0.0379, "h": 0.688}})
    cfg["solver"]["ksp_settings"]["ksp_rtol"] = rtol
    cfg["solver"]["ksp_settings"]["pc_type"] = pc
    return cfg


def mms_config(dim=2, N=8, dt=1e-5, steps=1, rtol=1e-12):
    """reference src/CGx/KNPEMI/configs/mms_config.yaml (+ the 'solver' section it lacks)"""
    return {"problem_type": "KNP-EMI", "quiet": True, "dt": dt, "time_steps": steps,
            "physical_constants": {"T": 1.0, "F": 1.0, "R": 1.0}, "C_M": 1.0,
            "cell_tag_file": f"{'square' if dim == 2 else 'cube'}{N}.xdmf", "facet_tag_file": f"{'square' if dim == 2 else 'cube'}{N}_facets.xdmf",
            "ics_tags": [1], "ecs_tags": [2], "boundary_tags": [8], "membrane_tags": [1, 2, 3, 4], "stimulus_tags": [],
            "MMS_test": {"N_mesh": N, "dim": dim},
            "solver": {"direct": False,
                       "ksp_settings": {"ksp_rtol": rtol, "ksp_type": "gmres", "pc_type": "hypre", "ksp_max_it": 2000},
                       "output": {"save_xdmf": False, "save_cpoints": False, "save_pngs": False, "save_dat": False}}}


def fp32_stored(h):
    """The hierarchy as the library holds it with ``amg_fp32`` (default): level, transfer and coarse-inverse
    VALUES rounded to fp32 (diagonals, vectors, arithmetic and the dense coarse inverse stay fp64)."""
    import copy
    import numpy as np

    def rnd(M):
        if M is None:
            return None
        M = M.copy()
        if hasattr(M, "data"):
            M.data = M.data.astype(np.float32).astype(np.float64)
            return M
        return M.astype(np.float32).astype(np.float64)
    out = copy.copy(h)
    out.levels = []
    for lv in h.levels:
        l2 = copy.copy(lv)
        l2.A, l2.P, l2.R = rnd(lv.A), rnd(lv.P), rnd(lv.R)
        out.levels.append(l2)
    return out


def tissue_config(dim=2, N=16, m=2, steps=2, rtol=1e-11, pc="hypre", stimulus=True):
    """Tissue surrogate (lattice of cells, one tag per cell = its membrane tag; the shape of the reference's
    configs/5m/100c.yaml: ics_tags range, membrane tags = cell tags, stimulus restricted to an x-range)."""
    cfg = copy.deepcopy(CI_BASE)
    K = m ** dim
    cells = list(range(2, 2 + K))
    name = f"tissue{dim}d_{N}_{m}.xdmf"
    cfg.update({"time_steps": steps, "cell_tag_file": name, "facet_tag_file": name,
                "ics_tags": cells, "ecs_tags": [1], "membrane_tags": cells})
    if stimulus:
        cfg["stimulus_tags"] = cells
        cfg["stimulus_region"] = {"direction": "x", "range": [0.0, 0.5]}
    cfg["solver"]["ksp_settings"]["ksp_rtol"] = rtol
    cfg["solver"]["ksp_settings"]["pc_type"] = pc
    return cfg
