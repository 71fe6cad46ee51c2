#!/usr/bin/env python3
"""bench.py -- MDoF/s per implicit KNP-EMI timestep (assembly + GMRES) on N MI355X GPUs of one node.

One "step" = one implicit timestep of the reference's loop (src/CGx/KNPEMI/KNPEMIx_solver.py:365-468):
Hodgkin-Huxley gating update, assembly of A and b, GMRES(30) solve with the AMG-on-block-diagonal-P
preconditioner (the native counterpart of the reference's BoomerAMG-on-P), unpack + phi_m update.

Workload (synthetic, deterministic, no RNG): BASELINE.json configs[1] -- the 512x512 unit square with the
inner square [0.25,0.75]^2 as one cell, 3 ions, the CI physics of the reference's test YAML
(HH + ATP pump + neuronal cotransporters), mesh scaled to micrometres, dt 25 us, rtol 1e-9.  With N GPUs the
domain is N such unit squares stacked along y (one per rank, weak scaling, fixed work per GPU) with
ghost-layer halo exchange and Krylov all-reduces over RCCL.  ``--workload cubeM`` selects the 3D analogue.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the byte model of `roofline`).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "knp-emi-cgx_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", type=str, default="square512",
                    help="square<N> | cube<N> (per-GPU mesh, weak scaling) | tissue<dim>d_<N>_<m> (lattice of m^dim cells, one tag "
                         "per cell: the tissue surrogate of SURVEY 8d; the global mesh is partitioned over the ranks)")
    ap.add_argument("--pc", type=str, default="auto", help="auto (hypre-form AMG in 2D, btcc in 3D) | hypre | btcc | vbjacobi | none")
    ap.add_argument("--rtol", type=float, default=1e-9)
    ap.add_argument("--models", type=str, default="ci", help="ci (HH+ATP+cotransporters) | passive")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--profile-all", action="store_true", help="time every kernel class with HIP events (adds overhead)")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                    help="override a ksp_settings entry (e.g. --set amg_cheby_degree=2); tuning runs only")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("KNP_DIST_BACKEND", "nccl")      # "gloo": rehearsal with several ranks on one GPU
        dev_index = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import re
    from parity_utils import ci_config, make_problem, tissue_config
    from cgx_hip import _lib
    from cgx_hip.parallel import stacked_cubes_local_mesh, stacked_squares_local_mesh
    from CGx.KNPEMI.KNPEMIx_solver import SolverKNPEMI

    total_steps = args.warmup + args.steps
    mt = re.fullmatch(r"tissue(\d)d_(\d+)_(\d+)", args.workload)
    m = re.fullmatch(r"(square|cube)(\d+)", args.workload)
    assert m or mt, "workload must be square<N>, cube<N> or tissue<dim>d_<N>_<m>"
    if mt:
        tdim, N, ncell = int(mt.group(1)), int(mt.group(2)), int(mt.group(3))
        kind = "square" if tdim == 2 else "cube"
        if args.pc == "auto":
            args.pc = "hypre" if tdim == 2 else "btcc"
        cfg = tissue_config(tdim, N, ncell, steps=total_steps, rtol=args.rtol, pc=args.pc, stimulus=(args.models == "ci"))
        problem = make_problem(cfg, models=args.models)          # the problem partitions the global mesh itself
        what = f"tissue surrogate: unit {kind} {N}^{tdim} with {ncell}^{tdim} cells (one tag each), over {world} GPU(s)"
    else:
        kind, N = m.group(1), int(m.group(2))
        if args.pc == "auto":
            args.pc = "hypre" if kind == "square" else "btcc"
        gen = stacked_squares_local_mesh if kind == "square" else stacked_cubes_local_mesh
        lm = gen(N, world, rank, scale=1e-6)
        cfg = ci_config(N=N, steps=total_steps, rtol=args.rtol, pc=args.pc, kind=kind)
        problem = make_problem(cfg, models=args.models, local_mesh=lm)
        what = f"{world} x unit {kind} {N}^{2 if kind == 'square' else 3} (BASELINE configs[{1 if kind == 'square' else 2}])"
    problem.solver_config["view_ksp"] = False
    for kv in args.set:
        k, v = kv.split("=", 1)
        if v.lower() in ("true", "false"):
            val = v.lower() == "true"
        else:
            try:
                val = int(v)
            except ValueError:
                try:
                    val = float(v)
                except ValueError:
                    val = v
        problem.solver_config["ksp_settings"][k] = val
    solver = SolverKNPEMI(problem, solver_config=problem.solver_config)

    # ---- run the reference loop, but split into warmup and timed parts -------------------------
    from cgx_hip.ionic_models import HodgkinHuxley
    solver.setup_solver()
    be = solver.backend
    if solver._pc_kind in (_lib.PC_AMG, _lib.PC_AMG_BT):
        problem.setup_preconditioner(solver.use_block_Jacobi)
        solver.assemble_preconditioner()
    be.pc_setup(solver._pc_kind)

    def one_step(i):
        problem.t.value += float(problem.dt.value)
        if problem.gating_variables:
            for model in problem.ionic_models:
                if isinstance(model, HodgkinHuxley):
                    model.update_t_mod()
                    model.update_gating_variables()
        be.assemble_rhs()
        if i > 1:
            be.gmres_prepare()          # ||B b|| on the side stream while the matrix is assembled (as SolverKNPEMI.assemble does)
        be.assemble_matrix()
        if i == 1:
            solver.create_and_set_nullspace()
        its, rnorm, reason = be.gmres(solver._rtol, 1e-50, solver.ksp_max_it, solver.gmres_restart)
        be.unpack()
        return its, reason

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step_no = 0
    its_all, reasons = [], []
    for _ in range(args.warmup):
        step_no += 1
        it, rs = one_step(step_no)
    if args.warmup == 0:
        pass
    be.profile_reset()
    be.profile_enable(0x1f if args.profile_all else 0x1)       # class 0 = SpMV on A (dominant kernel)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_no += 1
        it, rs = one_step(step_no)
        its_all.append(it)
        reasons.append(rs)
    fence()
    elapsed = time.perf_counter() - t0
    prof = be.profile_get()
    be.profile_enable(0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    n_dof = be.n_dof_global
    ms_per_step = 1e3 * elapsed / max(args.steps, 1)
    value = n_dof * args.steps / elapsed / 1e6

    # ---- roofline of the dominant kernel (SpMV on the system matrix A), per launch, this rank -----------
    # Bytes the kernel must move: the node-structured kernel reads the CSR value array
    # (8 B/nnz) but only a 4-B neighbour index per node pair instead of a 4-B column index per entry.
    spmv_ms, spmv_n = prof["spmv"]
    n_own, n_loc = be.n_dof_owned, be.n_dof_local
    b_csr = 12.0 * be.nnz + 4.0 * (n_own + 1) + 8.0 * n_own + 8.0 * n_loc          # SURVEY 8(d) CSR figure
    node_kernel = os.environ.get("KNP_SPMV", "") != "csr"
    b_node = 8.0 * be.nnz + 4.0 * be.n_pairs + 4.0 * (n_own + 1) + 4.0 * (be.n_nodes_owned + 1) + 8.0 * n_own + 8.0 * n_loc
    b_alg = b_node if node_kernel else b_csr
    roof = None
    if spmv_n > 0 and spmv_ms > 0:
        avg_s = spmv_ms * 1e-3 / spmv_n
        ach = b_alg / avg_s / 1e9
        roof = {"bound": "hbm", "kernel": "k_spmv_node (SpMV on A)" if node_kernel else "k_spmv<L,*,1> (CSR SpMV on A)",
                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                "bytes_per_launch": b_alg, "csr_equivalent_bytes": b_csr, "csr_equivalent_GBs": b_csr / avg_s / 1e9,
                "avg_launch_us": avg_s * 1e6, "launches": int(spmv_n),
                "note": "working set of the 512^2 case (~150 MB matrix) largely stays in the 256 MB Infinity Cache"
                        if be.nnz * 8 < 200e6 else "matrix exceeds the Infinity Cache"}
        # HBM traffic per launch from the PMC passes committed under profiles/ (FETCH_SIZE doubled per the gfx950
        # 16-B-load correction + WRITE_SIZE; collected by `rocprofv3 --pmc` in separate runs of this command)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            ent = pmc.get(args.workload, {})
            hit = [v for k, v in ent.items() if k.startswith("void k_spmv_node") and k.endswith(", 0>")]
            if hit and node_kernel and world == 1:
                roof["traffic"] = hit[0]["hbm_bytes_corrected"]
        except Exception:      # noqa: BLE001
            pass

    norms = solver.potential_norms()

    # ---- CPU baseline: the oracle (NumPy/SciPy restatement, 1 core) on a bounded sample --------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not mt:
        cpu = cpu_baseline(kind, N, args.models, args.rtol, args.cpu_steps, args.pc, solver)

    if rank == 0:
        out = {
            "metric": "MDoF/s per implicit timestep (assembly+GMRES)",
            "value": value, "unit": "MDoF/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if (mt and world > 1) else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{what}, "
                                   f"3 ions, {'HH+ATP+cotransporters' if args.models == 'ci' else 'passive membrane'}, GMRES(30)+{'AMG on block-diagonal P' if args.pc in ('hypre', 'amg') else args.pc}, rtol {args.rtol:g}",
                       "n_dof": int(n_dof), "nnz": int(be.nnz_global), "mechanisms": args.models, "pc": args.pc,
                       "parallelism": f"dd{world}", "gmres_its_per_step": float(sum(its_all)) / max(len(its_all), 1),
                       "converged_all": bool(all(r > 0 for r in reasons)),
                       "phi_norms": [norms[0], norms[1]]},
            "roofline": roof,
            "cpu_baseline": cpu,
            "kernel_classes_ms": {k: {"ms": v[0], "launches": v[1]} for k, v in prof.items()} if args.profile_all else None,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(kind, N, models, rtol, steps, pc, solver):
    """Oracle timed on the host: same mesh, same physics, same algorithm (GMRES(30), left PC, the same
    AMG construction and cycle parameters, applied by the NumPy V-cycle), single thread."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import knpemi_oracle as K
    from cgx_hip import amg
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    mk = K.make_square if kind == "square" else K.make_cube
    mdl = K.CI_MODELS() if models == "ci" else [K.Model("passive", (4,))]
    o = mk(N, models=mdl)
    pre, post, deg = solver.amg_pre, solver.amg_post, solver.amg_cheby_degree

    def fac(P):
        if pc == "btcc":
            hk = amg.build_hierarchy(amg.restrict_to_fields(P, (0, 1, 2)), theta=solver.amg_theta, coarse_size=solver.amg_coarse_size)
            hp = amg.build_hierarchy(amg.restrict_to_fields(P, (3,)), theta=solver.amg_theta, coarse_size=solver.amg_coarse_size)
            return K.pc_btcc(o, hk, hp, pre, post, deg)
        h = amg.build_hierarchy(P, theta=solver.amg_theta, coarse_size=solver.amg_coarse_size)
        return K.pc_amg_vcycle(h.levels, h.coarse_inv, pre, post, deg)
    times = []

    def log(step, oo, x):
        times.append(time.perf_counter())
    t_start = time.perf_counter()
    _, its = o.run(steps + 1, solver="gmres", pc=fac, rtol=rtol, log=log)
    per = [(times[i] - times[i - 1]) for i in range(1, len(times))]      # step 1 (null-space check, setup) excluded
    sec = sum(per) / len(per)
    return {"value": o.n_dof / sec / 1e6, "unit": "MDoF/s", "cores": 1, "kind": "port",
            "sample": f"{steps} implicit steps of the same {kind}{N} workload (NumPy/SciPy oracle: vectorised assembly, "
                      f"GMRES(30)+{pc} with the NumPy V-cycle, {sum(its[1:]) / max(len(its) - 1, 1):.1f} its/step), "
                      f"{sec:.2f} s/step; total {time.perf_counter() - t_start:.1f} s incl. setup"}


if __name__ == "__main__":
    main()
