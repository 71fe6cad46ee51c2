"""Run configurations of the reference as Python dicts (the YAML schema of src/CGx/utils/mixed_dim_problem.py:86-374)
and the construction order of its drivers (src/CGx/KNPEMI/main.py:24-63, tests/KNPEMI/electric_potential_norms_*.py:27-41).
Used by ``CGx.KNPEMI.main``, ``bench.py`` and the tests; nothing here touches the GPU."""
from __future__ import annotations

import copy

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


def tissue_config(dim=2, N=16, m=2, steps=2, rtol=1e-11, pc="hypre", stimulus=True, gap=None, width=None):
    """Tissue surrogate (lattice of cells, one tag per cell = its membrane tag; the shape of the reference's
    configs/5m/100c.yaml: ics_tags range, membrane tags = cell tags, stimulus restricted to an x-range).
    ``gap``: extracellular gap around every cell in mesh cells (default: the generator's); ``width``: instead, extracellular
    sheets of that many mesh cells between the cells (membrane-dominated variant, N = m*B + width)."""
    cfg = copy.deepcopy(CI_BASE)
    K = m ** dim
    cells = list(range(2, 2 + K))
    name = f"tissue{dim}d_{N}_{m}" + (f"_g{gap}" if gap is not None else "") + (f"_w{width}" if width is not None else "") + ".xdmf"
    cfg.update({"time_steps": steps, "cell_tag_file": name, "facet_tag_file": name,
                "ics_tags": cells, "ecs_tags": [1], "membrane_tags": cells})
    if stimulus:
        cfg["stimulus_tags"] = cells
        cfg["stimulus_region"] = {"direction": "x", "range": [0.0, 0.5]}
    cfg["solver"]["ksp_settings"]["ksp_rtol"] = rtol
    cfg["solver"]["ksp_settings"]["pc_type"] = pc
    return cfg


def default_ionic_models(problem, config_name=""):
    """Mechanism set selection of the reference driver (main.py:27-45)."""
    from .ionic_models import (ATPPump, GlialCotransporters, HodgkinHuxley, KirNaKPumpModel, NeuronalCotransporters)
    if "square_config" in str(config_name):
        return [NeuronalCotransporters(problem), HodgkinHuxley(problem), ATPPump(problem)]
    if problem.glia_flag:
        nt, gt = problem.neuron_tags, problem.glia_tags
        return [HodgkinHuxley(problem, tags=nt), ATPPump(problem, tags=nt), NeuronalCotransporters(problem, tags=nt),
                GlialCotransporters(problem, tags=gt), KirNaKPumpModel(problem, tags=gt)]
    return [HodgkinHuxley(problem), ATPPump(problem), NeuronalCotransporters(problem)]


def make_problem(cfg, models="ci", local_mesh=None):
    """Construction order of the reference's drivers: problem -> models -> initial conditions -> init_ionic_models
    -> setup_variational_form.  ``models``: "ci" (the CI scripts' set and order), "passive", or a callable(problem) -> list."""
    from .ionic_models import ATPPump, HodgkinHuxley, NeuronalCotransporters, PassiveModel
    from .problem import ProblemKNPEMI
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
