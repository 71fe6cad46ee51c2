#!/usr/bin/env python3
"""BASELINE configs[3] at its stated size on ONE GPU (tissue surrogate, ~5e7 unknowns): run a few implicit steps through the
drop-in entry point and check what can be checked at that size without a second solver:

  * every solve converged; the gauge (sum of the potential unknowns) is conserved; A ns = 0 (KNPEMIx_solver.py:327);
  * TRUE residual on a sampled block: the oracle (test infrastructure) assembles A and b of the sampled step on the sub-mesh of a
    corner box from the GPU's state before that step; on the rows of the box's interior vertices -- whose element patches lie
    inside the sub-mesh -- b - A x with the GPU's solution must vanish to the solver tolerance.  Independent of the preconditioner,
    of the Krylov method and of the rest of the mesh.

usage: python tools/tissue_fullsize_check.py [workload=tissue3d_189_47_w1] [steps=4] [box_voxels=16]      (prints one JSON object)"""
import json, os, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("knp-emi-cgx_amd", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import torch
from cgx_hip.configs import make_problem, tissue_config
from CGx.KNPEMI.KNPEMIx_solver import SolverKNPEMI


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def run_check(w="tissue3d_189_47_w1", steps=4, box=16):
    mt = re.fullmatch(r"tissue3d_(\d+)_(\d+)_w(\d+)", w)
    N, m, width = int(mt.group(1)), int(mt.group(2)), int(mt.group(3))
    t0 = time.perf_counter()
    cfg = tissue_config(3, N, m, steps=steps, rtol=1e-9, pc="btcc", stimulus=True, width=width)
    p = make_problem(cfg, "ci")
    s = SolverKNPEMI(p, solver_config=p.solver_config)
    s.prepare()
    be = s.backend
    log(f"[{time.perf_counter() - t0:6.1f} s] problem + preconditioner ready: {be.n_dof_global} unknowns, {be.nnz_global} stored entries")
    lm = p.local_mesh
    X = lm.coords / lm.coords.max()
    hbox = (box + 0.5) / N
    vin = np.all(X <= hbox, axis=1)                                     # vertices of the corner box
    cin = vin[lm.cells].all(axis=1)                                     # cells inside it
    sub_v = np.nonzero(vin)[0]
    renum = np.full(len(X), -1, dtype=np.int64)
    renum[sub_v] = np.arange(sub_v.size)
    sub_cells = renum[lm.cells[cin]].astype(np.int32)
    interior = np.all(X[sub_v] <= (box - 0.5) / N, axis=1)              # not on the cut faces: their element patches are complete

    def restricted_state():
        st = {"k_i": [p.wh[0][j].numpy()[sub_v].copy() for j in range(3)], "k_e": [p.wh[1][j].numpy()[sub_v].copy() for j in range(3)],
              "phi_i": p.wh[0][3].numpy()[sub_v].copy(), "phi_e": p.wh[1][3].numpy()[sub_v].copy(), "phi_m": p.phi_m_prev.numpy()[sub_v].copy(),
              "t": float(p.t.value)}
        for nm in ("n", "m", "h"):
            st[nm] = getattr(p, nm).numpy()[sub_v].copy()
        return st
    state = None
    for i in range(1, steps + 1):
        if i == steps:
            state = restricted_state()
        s.step(i)
        log(f"[{time.perf_counter() - t0:6.1f} s] step {i}: {s.iterations[-1]} iterations, reason {s.reasons[-1]}")
    s.finish()
    x = be.x.cpu().numpy()
    n_intra = int((be.node_i >= 0).sum())
    out = {"workload": w, "n_dof": int(be.n_dof_global), "nnz": int(be.nnz_global), "steps": steps, "iterations": list(map(int, s.iterations)),
           "converged_all": bool(all(r > 0 for r in s.reasons)),
           "gauge_sum_rel_drift": float(abs(x[3::4].sum() - (-0.07 * n_intra)) / (0.07 * n_intra)),
           "A_ns_over_max_A": float(be.nullspace_test() / be.matrix_max_abs()),
           "solve_time_s": [float(v) for v in s.solve_time], "assembly_time_s": [float(v) for v in s.assembly_time], "setup_s": dict(s.setup_breakdown)}
    # ---- sampled block: oracle on the sub-mesh of the corner box ----
    import knpemi_oracle as K
    tags = np.where(np.asarray(p.cell_side)[cin] == 0, 1, 2)                    # intra / extra
    lo, hi = cfg["stimulus_region"]["range"]
    o = K.OracleKNPEMI(lm.coords[sub_v], sub_cells, tags, intra_tags=(1,), extra_tag=2, models=K.CI_MODELS(), mesh_conversion_factor=1.0,
                       stimulus_tags=(4,), stimulus_region=(0, lo * 1e-6, hi * 1e-6))
    o.stimulus_area = float(p.stimulus_area)                                    # a global integral: taken from the full problem
    o.load_state(state)
    A, b = o.step_system()
    xs = np.zeros(o.n_dof)
    for side, nodes_sub, nodes_full in ((0, o.lay.node_i, be.node_i), (1, o.lay.node_e, be.node_e)):
        v = np.nonzero(nodes_sub >= 0)[0]
        gf = nodes_full[sub_v[v]]
        assert (gf >= 0).all()
        for f in range(4):
            xs[4 * nodes_sub[v] + f] = x[4 * gf + f]
    r = b - A @ xs
    ax = abs(A) @ np.abs(xs)
    blocks = {}
    names = ("Na", "K", "Cl", "phi")
    for side, sn, nodes_sub in ((0, "i", o.lay.node_i), (1, "e", o.lay.node_e)):
        v = np.nonzero((nodes_sub >= 0) & interior)[0]
        for f in range(4):
            d = 4 * nodes_sub[v] + f
            nb = float(np.linalg.norm(b[d]))
            blocks[f"{names[f]}_{sn}"] = {"rows": int(d.size), "rel_to_b": float(np.linalg.norm(r[d]) / nb) if nb > 0 else None,
                                          "backward": float(np.linalg.norm(r[d]) / (np.linalg.norm(ax[d]) + nb))}
    out["sampled_block"] = {"box_voxels": box, "sub_mesh_vertices": int(sub_v.size), "sub_mesh_cells": int(cin.sum()),
                            "interior_vertices": int(interior.sum()), "membrane_vertices_in_box": int(((o.lay.node_i >= 0) & (o.lay.node_e >= 0)).sum()),
                            "blocks": blocks, "max_backward": max(v["backward"] for v in blocks.values())}
    out["ok"] = bool(out["converged_all"] and out["gauge_sum_rel_drift"] <= 1e-8 and out["A_ns_over_max_A"] <= 1e-10 and
                     out["sampled_block"]["max_backward"] <= 1e-8)
    out["wall_s"] = time.perf_counter() - t0
    return out


if __name__ == "__main__":
    res = run_check(sys.argv[1] if len(sys.argv) > 1 else "tissue3d_189_47_w1", int(sys.argv[2]) if len(sys.argv) > 2 else 4,
                    int(sys.argv[3]) if len(sys.argv) > 3 else 16)
    print(json.dumps(res), flush=True)
    sys.exit(0 if res["ok"] else 3)
