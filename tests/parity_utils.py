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

from cgx_hip.configs import CI_BASE, ci_config, make_problem, tissue_config  # noqa: E402,F401


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


from cgx_hip.amg import fp32_stored  # noqa: E402,F401


def oracle_gmres_same_algorithm(kind, N, steps, pc, rtol, solver, models="ci"):
    """The oracle stepping with PETSc-style GMRES(30) and the SAME preconditioner algorithm as the native solver
    (hierarchies rebuilt on the host with the solver's parameters, applied by the NumPy V-cycle, operator values rounded to fp32
    like the library stores them).  Returns (oracle, iterations)."""
    import knpemi_oracle as K
    from cgx_hip import amg
    o = make_oracle(N, kind, models)
    pre, post, deg = solver.amg_pre, solver.amg_post, solver.amg_cheby_degree
    fused = bool(solver.backend.stats()["fused"])
    rnd = fp32_stored if solver.amg_fp32 else (lambda h, **k: h)

    def fac(P):
        if pc == "btcc":
            hk = rnd(amg.build_hierarchy(amg.restrict_to_fields(P, (0, 1, 2)), theta=solver.amg_theta, coarse_size=solver.amg_coarse_size,
                                         node_fields=solver.ion_node_fields(), agg_distance=solver.ion_agg_distance()), coarse=fused)
            coupled = bool(getattr(solver, '_coupled_phi', False))      # potential hierarchy on the potential block of A (both sides, coupled)
            hp = rnd(amg.build_hierarchy(o.potential_block_of_A() if coupled else amg.restrict_to_fields(P, (3,)),
                                         theta=solver.amg_theta, coarse_size=solver.amg_coarse_size, agg_distance=solver.phi_agg_distance()),
                     level0_uploaded=coupled)
            return K.pc_btcc(o, hk, hp, pre, post, deg, fused=fused)
        h = rnd(amg.build_hierarchy(P, theta=solver.amg_theta, coarse_size=solver.amg_coarse_size, node_fields=solver.all_node_fields(),
                                    agg_distance=solver.ion_agg_distance()))
        return K.pc_amg_vcycle(h.levels, h.coarse_inv, pre, post, deg, fused=fused)
    _, its = o.run(steps, solver="gmres", pc=fac, rtol=rtol)
    return o, its


def snapshot_state(problem):
    """The time-dependent state of the native problem as host arrays (input of ``OracleKNPEMI.load_state``): the checker redoes
    ONE step from exactly the state the GPU path was in, so the comparison does not depend on the trajectory before it."""
    p = problem
    st = {"k_i": [p.wh[0][j].numpy().copy() for j in range(3)], "k_e": [p.wh[1][j].numpy().copy() for j in range(3)],
          "phi_i": p.wh[0][3].numpy().copy(), "phi_e": p.wh[1][3].numpy().copy(), "phi_m": p.phi_m_prev.numpy().copy(),
          "t": float(p.t.value)}
    for nm in ("n", "m", "h"):
        f = getattr(p, nm, None)
        st[nm] = f.numpy().copy() if f is not None and hasattr(f, "numpy") else None
    return st


def run_with_snapshots(solver, steps):
    """``SolverKNPEMI.solve()`` -- the reference's loop, nothing changed -- recording after every step in ``steps`` the state
    (``snapshot_state``), the solution vector, the potential norms and phi_m: ``snaps[i]``."""
    s, p = solver, solver.problem
    snaps = {}
    s.setup_solver()
    be = s.backend
    orig_unpack = be.unpack
    count = {"i": 0}

    def unpack():
        orig_unpack()
        count["i"] += 1
        if count["i"] in steps:
            snaps[count["i"]] = {"state": snapshot_state(p), "norms": s.potential_norms(), "phi_m": p.phi_m_prev.numpy().copy(),
                                 "x": be.x.cpu().numpy().copy()}
    be.unpack = unpack
    s.setup_solver = lambda: None
    s.solve()
    return snaps
