#!/usr/bin/env python3
"""bench.py -- MDoF/s per implicit KNP-EMI timestep (assembly + GMRES) on N MI355X GPUs of one node.

One "step" = one implicit timestep of the reference's loop (src/CGx/KNPEMI/KNPEMIx_solver.py:365-468):
Hodgkin-Huxley gating update, assembly of A and b, GMRES(30) solve with the AMG-on-block-diagonal-P
preconditioner (the native counterpart of the reference's BoomerAMG-on-P), unpack + phi_m update.

Workload (synthetic, deterministic, no RNG): BASELINE.json configs[1] -- the 512x512 unit square with the
inner square [0.25,0.75]^2 as one cell, 3 ions, the CI physics of the reference's test YAML
(HH + ATP pump + neuronal cotransporters), mesh scaled to micrometres, dt 25 us, rtol 1e-9.  With N GPUs the
domain is N such unit squares stacked along y (one per rank, weak scaling, fixed work per GPU) with
ghost-layer halo exchange and Krylov all-reduces over xGMI.  ``--workload cubeM`` selects the 3D analogue,
``--workload tissue...`` the tissue surrogates.

``python bench.py --gpus N`` started WITHOUT a launcher (no WORLD_SIZE in the environment) starts its N ranks
itself as child processes before anything touches the GPU; under ``python -m torch.distributed.run`` it is one rank.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the byte model of `roofline`).  The timed region
is ``--steps`` timesteps between barriers; when that is shorter than ~0.5 s the bracket is repeated (the simulation
simply continues) and the MEDIAN repetition is reported -- every repetition is listed under ``timing``.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "knp-emi-cgx_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MIN_TIMED_S = 0.5         # repeat the K-step bracket until this much has been timed
MAX_REPS = 40
PARITY_TOL = 1e-6         # north_star: potentials match the reference solve to rtol 1e-6
# True residual of the sampled step with the ORACLE's A and b.  The solve stops on the PRECONDITIONED residual at rtol 1e-9 (reference:
# ksp_norm_type preconditioned), which leaves 2e-6 of ||b|| overall on 512^2 (measured; the oracle's own GMRES leaves the same) and
# 1e-11 (512^2) to 1e-9 (64^3 with btcc) normwise backward error ||r|| / (|| |A||x| || + ||b||) in every field block -- the phi-rows have an almost empty right-hand side
# (membrane terms only), so only the backward error is a meaningful gate there.
TRUE_RES_TOL = 1e-5
BACKWARD_TOL = 1e-8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", type=str, default="square512",
                    help="square<N> | cube<N> (per-GPU mesh, weak scaling) | tissue<dim>d_<N>_<m>[_g<gap>|_w<width>] (lattice of m^dim "
                         "cells, one tag per cell: the tissue surrogate of SURVEY 8d; the global mesh is partitioned over the ranks)")
    ap.add_argument("--pc", type=str, default="auto", help="auto (hypre-form AMG in 2D, btcc in 3D) | hypre | btcc | vbjacobi | none")
    ap.add_argument("--rtol", type=float, default=1e-9)
    ap.add_argument("--models", type=str, default="ci", help="ci (HH+ATP+cotransporters) | passive")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--no-parity-lu", action="store_true", help="parity leg: true residual only, skip the oracle's sparse direct solve "
                    "(nested-dissection LU of the sampled step: ~1.5 min and 6 GB at 512^2; 2D workloads up to 1.2 M unknowns only)")
    ap.add_argument("--large", type=str, default="cube136", help="out-of-cache workload of the roofline_large block ('' or 'none' disables)")
    ap.add_argument("--large-steps", type=int, default=4)
    ap.add_argument("--no-repeat", action="store_true", help="time the K steps once, whatever their duration")
    ap.add_argument("--class-steps", type=int, default=5, help="steps of the per-class roofline pass after the timed region (0 disables)")
    ap.add_argument("--profile-all", action="store_true", help="time every kernel class with HIP events in the MAIN run (adds overhead)")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                    help="override a ksp_settings entry (e.g. --set amg_cheby_degree=2); tuning runs only")
    return ap.parse_args()


# --------------------------------------------------------------------------------------------------------------
# launcher: N ranks as child processes (no GPU call, no torch import in this parent)
# --------------------------------------------------------------------------------------------------------------
def launch_ranks(n):
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr, flush=True)
                for q in alive:
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


# --------------------------------------------------------------------------------------------------------------
def build_case(workload, args, world, rank, steps_total):
    from cgx_hip.configs import ci_config, make_problem, tissue_config
    from cgx_hip.parallel import stacked_cubes_local_mesh, stacked_squares_local_mesh
    from CGx.KNPEMI.KNPEMIx_solver import SolverKNPEMI
    mt = re.fullmatch(r"tissue(\d)d_(\d+)_(\d+)(?:_g(\d+))?(?:_w(\d+))?", workload)
    m = re.fullmatch(r"(square|cube)(\d+)", workload)
    assert m or mt, "workload must be square<N>, cube<N> or tissue<dim>d_<N>_<m>[_g<gap>|_w<width>]"
    pc = args.pc
    if mt:
        tdim, N, ncell = int(mt.group(1)), int(mt.group(2)), int(mt.group(3))
        kind = "square" if tdim == 2 else "cube"
        if pc == "auto":
            pc = "hypre" if tdim == 2 else "btcc"
        cfg = tissue_config(tdim, N, ncell, steps=steps_total, rtol=args.rtol, pc=pc, stimulus=(args.models == "ci"),
                            gap=int(mt.group(4)) if mt.group(4) else None, width=int(mt.group(5)) if mt.group(5) else None)
        problem = make_problem(cfg, models=args.models)          # the problem partitions the global mesh itself
        what = f"tissue surrogate {workload}: unit {kind} {N}^{tdim} with {ncell}^{tdim} cells (one tag each), over {world} GPU(s)"
    else:
        kind, N = m.group(1), int(m.group(2))
        if pc == "auto":
            pc = "hypre" if kind == "square" else "btcc"
        gen = stacked_squares_local_mesh if kind == "square" else stacked_cubes_local_mesh
        lm = gen(N, world, rank, scale=1e-6)
        cfg = ci_config(N=N, steps=steps_total, rtol=args.rtol, pc=pc, kind=kind)
        problem = make_problem(cfg, models=args.models, local_mesh=lm)
        what = f"{world} x unit {kind} {N}^{2 if kind == 'square' else 3} (BASELINE configs[{1 if kind == 'square' else 2}])" if N in (512, 64) \
            else f"{world} x unit {kind} {N}^{2 if kind == 'square' else 3}"
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
    return {"problem": problem, "solver": solver, "kind": kind, "N": N, "pc": pc, "what": what, "tissue": bool(mt)}


class Stepper:
    """Drives the drop-in entry point itself: ``SolverKNPEMI.prepare()`` once, then ``SolverKNPEMI.step(i)`` per timestep -- the very
    loop body ``SolverKNPEMI.solve()`` runs (reference KNPEMIx_solver.py:365-468); nothing is re-implemented here."""

    def __init__(self, case):
        self.solver, self.problem = case["solver"], case["problem"]
        t0 = time.perf_counter()
        self.solver.prepare()
        self.be = self.solver.backend
        self.prepare_s = time.perf_counter() - t0
        self.i = 0

    def step(self):
        self.i += 1
        s = self.solver
        s.step(self.i)
        return s.iterations[-1], s.reasons[-1]


def comm_report(be, world, torch, dist):
    """Which exchange path the ranks ended up on: the native peer-to-peer kernels (every plan passed its self-test against the
    torch.distributed exchange on every rank) or the torch.distributed hooks; never measured on more than one device before the
    driver's own multi-GPU run, so the line says what ran."""
    if world == 1:
        return {"path": "none (one GPU)", "device_count": torch.cuda.device_count()}
    p2p = bool(getattr(be, "p2p_on", False))
    plans = getattr(be, "_p2p_plans", {})
    return {"path": "p2p (native kernels over peer-mapped mailboxes)" if p2p else f"hooks-{dist.get_backend()} (torch.distributed)",
            "p2p_selftest": "passed on all ranks" if p2p else ("disabled (KNP_COMM)" if os.environ.get("KNP_COMM", "p2p") != "p2p" else "failed or unavailable: fell back on every rank"),
            "p2p_plans": int(sum(1 for v in plans.values() if v is not None)), "backend": dist.get_backend(),
            "device_count": torch.cuda.device_count(), "p2p_timeout_s": float(os.environ.get("KNP_P2P_TIMEOUT", "30")),
            "halo_overlap": {"spmv_interior_rows": os.environ.get("KNP_SPMV_SPLIT", "1") != "0" and p2p,
                             "pc_level0_interior_rows": os.environ.get("KNP_PC_SPLIT", "1") != "0" and p2p}}


def spmv_bytes(be):
    """Bytes one SpMV on A must move: the node-structured kernel reads the matrix values (8 B per entry,
    pair-major) and a 4-B neighbour index per node PAIR instead of a 4-B column index per entry, the pair pointer, the
    membrane index and side of every node, a 4-B column per membrane coupling, x and y.  Returns (node kernel, CSR)."""
    n_own, n_loc = be.n_dof_owned, be.n_dof_local
    b_csr = 12.0 * be.nnz + 4.0 * (n_own + 1) + 8.0 * n_own + 8.0 * n_loc          # SURVEY 8(d) CSR figure
    # the library states what its kernel reads (knp_get_traffic_model): per pair 32 B of entries that depend on the previous solution,
    # 16 B {mass, stiffness} from which the six time-invariant entries are recomputed (48 B if they are read), the neighbour index
    b_node = be.traffic_model()["spmv"]
    return b_node, b_csr


def roofline_block(be, prof, workload, world):
    spmv_ms, spmv_n = prof["spmv"]
    b_node, b_csr = spmv_bytes(be)
    node_kernel = True
    b_alg = b_node
    if not (spmv_n > 0 and spmv_ms > 0):
        return None
    avg_s = spmv_ms * 1e-3 / spmv_n
    ach = b_alg / avg_s / 1e9
    in_cache = be.nnz * 8 < 200e6
    roof = {"bound": "hbm", "kernel": "k_spmv_node (SpMV on A)",
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
            "traffic_source": None, "bytes_per_launch": b_alg, "csr_equivalent_bytes": b_csr, "csr_equivalent_GBs": b_csr / avg_s / 1e9,
            "avg_launch_us": avg_s * 1e6, "launches": int(spmv_n), "working_set": "infinity-cache resident" if in_cache else "exceeds the infinity cache",
            "note": ("the matrix of this case (%.0f MB) stays in the 256 MB Infinity Cache: the fraction is memory-system, not pure HBM, "
                     "bandwidth -- see roofline_large for the out-of-cache figure" % (be.nnz * 8 / 1e6)) if in_cache
                    else "matrix exceeds the Infinity Cache: HBM bandwidth"}
    # HBM traffic per launch is a PMC quantity (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE doubled per the
    # gfx950 16-B-load correction): it cannot be collected inside this process, so it is read from the committed profile of this command
    for fn in ("r03_pmc_traffic.json", "r02_pmc_traffic.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", fn)))
            ent = pmc.get(workload, {})
            hit = [v for k, v in ent.items() if re.search(r"k_spmv_node<\d+, 0[,>]", k)]      # MODE 0 = the product y = A x
            if hit and node_kernel and world == 1:
                roof["traffic"] = hit[0]["hbm_bytes_corrected"]
                roof["traffic_source"] = ("from_committed_profile: profiles/" + fn + " (2 x FETCH_SIZE + WRITE_SIZE: an upper bound where the "
                                          "doubled fetch counter also covers the gathered x)")
                break
        except Exception:      # noqa: BLE001
            pass
    return roof


def timed_run(case, args, world, dist, torch, steps, warmup, snapshot_at=None, allow_repeat=True, profile_mask=0x1):
    """warmup untimed steps, then `steps` steps between barriers, repeated while the timed total is below MIN_TIMED_S."""
    st = Stepper(case)
    be = st.be
    solver = case["solver"]

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    snap = None
    state_before = None
    t_first = None
    for _ in range(warmup):
        if snapshot_at is not None and st.i == snapshot_at - 1 and st.i >= 1:
            state_before = problem_state(case["problem"])      # the state the snapshot step starts from (single-step checker)
        t0 = time.perf_counter()
        st.step()
        if st.i == 1:
            torch.cuda.synchronize()
            t_first = time.perf_counter() - t0
        if snapshot_at is not None and st.i == snapshot_at:
            ni, ne = solver.potential_norms()
            snap = {"step": st.i, "phi_i": ni, "phi_e": ne, "phi_m": case["problem"].phi_m_prev.numpy().copy(),
                    "x": be.x.cpu().numpy().copy(), "state_before": state_before}
    be.profile_reset()
    be.profile_enable(int(os.environ.get("KNP_BENCH_PROFILE_MASK", profile_mask)))      # (developer knob: 0 = no events at all)
    reps, its_all, reasons = [], [], []
    total = 0.0
    while True:
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            it, rs = st.step()
            its_all.append(it)
            reasons.append(rs)
        fence()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        reps.append(el)
        total += el
        if not allow_repeat or total >= MIN_TIMED_S or len(reps) >= MAX_REPS:
            break
    prof = be.profile_get()
    stats = be.stats()
    be.profile_enable(0)
    srt = sorted(reps)
    med = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    setup = dict(solver.setup_breakdown)
    setup["amg_hierarchy_s"] = float(getattr(solver, "amg_setup_time", 0.0))      # part of preconditioner_setup_s
    setup["amg_phases"] = {k: round(v, 3) for k, v in getattr(solver, "amg_setup_phases", {}).items()}
    setup["first_step_s"] = t_first
    setup["total_before_first_step_s"] = st.prepare_s
    return {"stepper": st, "elapsed": med, "reps": reps, "its": its_all, "reasons": reasons, "prof": prof, "stats": stats, "snap": snap,
            "setup_s": setup}


def problem_state(p):
    """host copy of the time-dependent state (what oracle.load_state takes): concentrations, potentials, phi_m, gating, t"""
    st = {"k_i": [p.wh[0][j].numpy().copy() for j in range(3)], "k_e": [p.wh[1][j].numpy().copy() for j in range(3)],
          "phi_i": p.wh[0][3].numpy().copy(), "phi_e": p.wh[1][3].numpy().copy(), "phi_m": p.phi_m_prev.numpy().copy(), "t": float(p.t.value)}
    for nm in ("n", "m", "h"):
        f = getattr(p, nm, None)
        st[nm] = f.numpy().copy() if f is not None and hasattr(f, "numpy") else None
    return st


def main_case(args, world, rank, dist, torch):
    """The headline measurement; returns (json dict, failure text or None, what the CPU baseline / parity legs need or None)."""
    # the parity snapshot is taken during warmup (never inside the timed region), at the last step the oracle sample covers
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    snap_at = min(args.warmup, args.cpu_steps + 1) if want_cpu and args.warmup >= 1 else None

    case = build_case(args.workload, args, world, rank, args.warmup + args.steps * MAX_REPS)
    run = timed_run(case, args, world, dist, torch, args.steps, args.warmup, snapshot_at=snap_at,
                    allow_repeat=not args.no_repeat, profile_mask=0x1f if args.profile_all else 0x1)
    be, solver = run["stepper"].be, case["solver"]
    elapsed = run["elapsed"]
    n_dof = be.n_dof_global
    ms_per_step = 1e3 * elapsed / max(args.steps, 1)
    value = n_dof * args.steps / elapsed / 1e6
    roof = roofline_block(be, run["prof"], args.workload, world)
    classes = class_pass(run["stepper"], args, torch) if (world == 1 and args.class_steps > 0) else None
    norms = solver.potential_norms()
    n_steps_timed = len(run["its"])
    per_it = {k: run["stats"][k] / max(sum(run["its"]), 1) for k in ("allreduces", "halos", "readbacks", "norm_fallbacks")}
    out = {
        "metric": "MDoF/s per implicit timestep (assembly+GMRES)",
        "value": value, "unit": "MDoF/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if (case["tissue"] and world > 1) else "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{case['what']}, 3 ions, {'HH+ATP+cotransporters' if args.models == 'ci' else 'passive membrane'}, "
                               f"GMRES(30)+{'AMG on block-diagonal P' if case['pc'] in ('hypre', 'amg') else case['pc']}, rtol {args.rtol:g}",
                   "n_dof": int(n_dof), "nnz": int(be.nnz_global), "mechanisms": args.models, "pc": case["pc"],
                   "amg_agg_distance": solver.ion_agg_distance(), "vertex_order": os.environ.get("KNP_VERTEX_ORDER", "native"),
                   "parallelism": f"dd{world}", "gmres_its_per_step": float(sum(run["its"])) / max(n_steps_timed, 1),
                   "converged_all": bool(all(r > 0 for r in run["reasons"])),
                   "phi_norms": [norms[0], norms[1]],
                   "exchanges_per_gmres_iteration": per_it,
                   "comm": comm_report(be, world, torch, dist)},
        "timing": {"reps": len(run["reps"]), "steps_per_rep": args.steps, "ms_per_step_each_rep": [1e3 * r / max(args.steps, 1) for r in run["reps"]],
                   "stat": "median of the repetitions" if len(run["reps"]) > 1 else "single bracket",
                   "timed_total_s": sum(run["reps"])},
        "roofline": roof,
        "cpu_baseline": None, "parity": None, "roofline_large": None,
        "kernel_classes_ms": ({k: {"ms": v["ms_per_step"], "launches": v["launches_per_step"]} for k, v in classes["classes"].items()} if classes else
                              ({k: {"ms": v[0], "launches": v[1]} for k, v in run["prof"].items()} if args.profile_all else None)),
        "kernel_classes": classes,
        "setup_s": run["setup_s"],
        "entry_point": "SolverKNPEMI.prepare() + SolverKNPEMI.step(i): the loop body of SolverKNPEMI.solve()",
    }
    fail = None
    if not out["config"]["converged_all"]:
        fail = "GMRES did not converge in every timed step"

    # what the CPU legs need of this run (they run after the out-of-cache block, while the GPU would otherwise idle and clock down
    # in front of it): solver parameters, the evaluation order of the cycle, the snapshot
    info = None
    if want_cpu and not case["tissue"]:
        class _P:      # plain record: the device objects of this case can be freed
            pass
        sp_ = _P()
        for k in ("amg_pre", "amg_post", "amg_cheby_degree", "amg_theta", "amg_coarse_size", "amg_fp32", "amg_node_sync"):
            setattr(sp_, k, getattr(solver, k))
        sp_.fused = bool(be.stats()["fused"])
        sp_.coupled_phi = bool(getattr(solver, "_coupled_phi", False))
        sp_.ion_dist, sp_.phi_dist = solver.ion_agg_distance(), solver.phi_agg_distance()
        info = {"case": {"kind": case["kind"], "N": case["N"], "pc": case["pc"]}, "solver": sp_, "snap": run["snap"]}
    return out, fail, info


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("KNP_DIST_BACKEND", "nccl")      # "gloo": rehearsal with several ranks on one GPU
        n_dev = torch.cuda.device_count()
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        if backend == "nccl" and n_dev < local_world:
            # RCCL cannot put two ranks on one device: say so and stop instead of waiting in the rendezvous
            print(f"bench.py: --gpus {world} needs {local_world} visible GPUs on this node, found {n_dev} "
                  f"(KNP_DIST_BACKEND=gloo rehearses several ranks on one GPU)", file=sys.stderr, flush=True)
            sys.exit(4)
        dev_index = local_rank % max(n_dev, 1)
        torch.cuda.set_device(dev_index)
        tmo = datetime.timedelta(seconds=float(os.environ.get("KNP_RENDEZVOUS_TIMEOUT", "180")))      # a missing rank ends the run, it does not hang it
        try:
            if backend == "nccl":
                dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index), timeout=tmo)
            else:
                dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=tmo)
        except Exception as exc:      # noqa: BLE001
            print(f"bench.py: rank {rank}: rendezvous failed ({type(exc).__name__}: {exc})", file=sys.stderr, flush=True)
            sys.exit(5)
    else:
        torch.cuda.set_device(0)

    out, fail, info = main_case(args, world, rank, dist, torch)

    # ---- out-of-cache roofline: the 10^7-DoF-per-GPU point, same process, every kernel class timed with HIP events ----
    large = (args.large or "").lower()
    if rank == 0 and world == 1 and large not in ("", "none") and large != args.workload:
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        out["roofline_large"] = large_block(args, torch, dist)

    # ---- CPU baseline (oracle, 1 core; C/OpenMP twin at 1 core and all usable cores) + parity of the GPU run against it ----
    if info is not None:
        cpu, par = cpu_baseline(info["case"], args, info["solver"], info["snap"])
        out["cpu_baseline"], out["parity"] = cpu, par
        if par is not None and not par["ok"]:
            fail = f"GPU solution differs from the oracle beyond {PARITY_TOL:g}: {par}"

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if fail:
        print("bench.py: " + fail, file=sys.stderr, flush=True)
        sys.exit(3)


def class_pass(st, args, torch, n_steps=None):
    """Per-class roofline: a few MORE steps of the same run with every kernel class bracketed by HIP events.  The events
    serialise what the timed region overlaps (matrix assembly next to the right-hand side, ||B b|| next to the first residual), so
    the class times add up to more than ms_per_step: they are per-class costs, not a decomposition of the headline number.  Bytes
    are the library's own statement of what each application reads and writes (knp_get_traffic_model); the orthogonalisation of
    iteration j moves (2 (j + 1) + 3) vectors (multi-dot + update)."""
    be = st.be
    n_steps = n_steps or args.class_steps
    be.profile_reset()
    be.profile_enable(0x1f)
    its = []
    for _ in range(n_steps):
        it, _ = st.step()
        its.append(it)
    torch.cuda.synchronize()
    prof = be.profile_get()
    be.profile_enable(0)
    tm = be.traffic_model()
    orth_vectors = sum(sum(2 * (j + 1) + 3 for j in range(k)) + 6 for k in its)       # + the two norms and the scaling of a solve
    by = {"spmv": tm["spmv"] * prof["spmv"][1], "orthogonalisation": tm["vector"] * orth_vectors, "pc": tm["pc"] * prof["pc"][1],
          "assembly": (tm["assembly_matrix"] + tm["assembly_rhs"]) * n_steps}
    out = {}
    for k in ("spmv", "orthogonalisation", "pc", "assembly"):
        ms, n = prof[k]
        gbs = by[k] / (ms * 1e-3) / 1e9 if ms > 0 else None
        out[k] = {"ms_per_step": ms / n_steps, "launches_per_step": n / n_steps, "bytes_per_step": by[k] / n_steps, "GBs": gbs,
                  "frac_of_hbm_peak": gbs / HBM_PEAK_GBS if gbs else None}
    out["other"] = {"ms_per_step": prof["other"][0] / n_steps, "launches_per_step": prof["other"][1] / n_steps, "bytes_per_step": None, "GBs": None,
                    "frac_of_hbm_peak": None}
    return {"steps": n_steps, "gmres_its": its, "classes": out,
            "bytes_model": {"spmv_per_launch": tm["spmv"], "pc_per_application": tm["pc"], "assembly_matrix_per_step": tm["assembly_matrix"],
                            "assembly_rhs_per_step": tm["assembly_rhs"], "vector": tm["vector"]},
            "note": "event-bracketed classes run serialised (no stream overlap): per-class costs, their sum exceeds ms_per_step; 'launches' of pc = "
                    "preconditioner applications, of assembly = assembly calls; in-cache workloads report memory-system, not HBM, bandwidth"}


def large_block(args, torch, dist):
    """SpMV roofline and per-class times on a working set far beyond the 256 MB Infinity Cache (cube136: 10.4 M DoF,
    nnz 3.9e8 = 3.1 GB of matrix values)."""
    t0 = time.perf_counter()
    a2 = argparse.Namespace(**vars(args))
    a2.pc = "auto"
    # six untimed steps first: the iteration count of the first steps after the initial condition (12-17) is a transient, the
    # per-step cost that is quoted is the one of the settled run (9-10), like the headline case after its warmup
    lw = 6
    case = build_case(args.large, a2, 1, 0, lw + args.large_steps + 8)
    run = timed_run(case, a2, 1, dist, torch, args.large_steps, lw, allow_repeat=False, profile_mask=0x1)
    be = run["stepper"].be
    roof = roofline_block(be, run["prof"], args.large, 1)
    classes = class_pass(run["stepper"], a2, torch, n_steps=max(2, min(args.large_steps, 4)))
    ms = 1e3 * run["elapsed"] / max(args.large_steps, 1)
    n_steps = max(args.large_steps, 1)
    blk = {"workload": case["what"] + f", {case['pc']}", "n_dof": int(be.n_dof_global), "nnz": int(be.nnz_global), "steps": args.large_steps, "warmup": lw,
           "ms_per_step": ms, "MDoF_per_s": be.n_dof_global / ms / 1e3, "gmres_its_per_step": float(sum(run["its"])) / n_steps,
           "converged_all": bool(all(r > 0 for r in run["reasons"])),
           "spmv": roof,
           "kernel_classes_ms_per_step": {k: {"ms": v["ms_per_step"], "launches": v["launches_per_step"]} for k, v in classes["classes"].items()},
           "kernel_classes": classes,
           "setup_s": run["setup_s"], "wall_s_incl_setup": None}
    blk["wall_s_incl_setup"] = time.perf_counter() - t0
    return blk


def cpu_baseline(case, args, solver, snap):
    """CPU legs, timed on this box's host cores on a bounded sample of the same workload:
       "port"      the NumPy/SciPy oracle (same mesh, physics and algorithm: GMRES(30), left PC, the same AMG construction and cycle
                   parameters applied by the NumPy V-cycle), single thread -- also the parity checker of the GPU run;
       "port-omp"  the C/OpenMP twin of the per-step kernels (oracle/knpemi_cpu.c), at 1 thread and at all cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import knpemi_oracle as K
    from cgx_hip import amg
    kind, N, models, rtol, pc = case["kind"], case["N"], args.models, args.rtol, case["pc"]
    steps = args.cpu_steps
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    mk = K.make_square if kind == "square" else K.make_cube
    mdl = K.CI_MODELS() if models == "ci" else [K.Model("passive", (4,))]
    o = mk(N, models=mdl)
    pre, post, deg = solver.amg_pre, solver.amg_post, solver.amg_cheby_degree

    fused = solver.fused                                             # the order in which the library's cycle evaluates its operators
    rnd = amg.fp32_stored if solver.amg_fp32 else (lambda h, **k: h)    # ... and their values as it stores them

    built = {}

    def fac(P, wrap=lambda h: h):      # hierarchies are built once (host setup, like ksp.setUp()) and shared by both CPU legs
        if not built:
            if pc == "btcc":
                built["k"] = rnd(amg.build_hierarchy(amg.restrict_to_fields(P, (0, 1, 2)), theta=solver.amg_theta, coarse_size=solver.amg_coarse_size,
                                                     node_fields=(4, (0, 1, 2)) if solver.amg_node_sync else None, agg_distance=solver.ion_dist), coarse=fused)
                built["p"] = rnd(amg.build_hierarchy(o.potential_block_of_A() if solver.coupled_phi else amg.restrict_to_fields(P, (3,)),
                                                     theta=solver.amg_theta, coarse_size=solver.amg_coarse_size, agg_distance=solver.phi_dist),
                                 level0_uploaded=solver.coupled_phi)
            else:
                built["h"] = rnd(amg.build_hierarchy(P, theta=solver.amg_theta, coarse_size=solver.amg_coarse_size,
                                                     node_fields=(4, (0, 1, 2, 3)) if solver.amg_node_sync else None, agg_distance=solver.ion_dist))
        if pc == "btcc":
            return K.pc_btcc(o, wrap(built["k"]), wrap(built["p"]), pre, post, deg, fused=fused)
        h = wrap(built["h"])
        return K.pc_amg_vcycle(h.levels, h.coarse_inv, pre, post, deg, fused=fused)
    times, osnap = [], {}

    def log(step, oo, x):
        times.append(time.perf_counter())
        if snap is not None and step == snap["step"]:
            ni, ne = oo.potential_norms()
            osnap.update(phi_i=ni, phi_e=ne, phi_m=np.array(oo.phi_m, copy=True))
    t_start = time.perf_counter()
    _, its = o.run(steps + 1, solver="gmres", pc=fac, rtol=rtol, log=log)
    per = [(times[i] - times[i - 1]) for i in range(1, len(times))]      # step 1 (null-space check, setup) excluded
    sec = sum(per) / len(per)
    cpu = {"value": o.n_dof / sec / 1e6, "unit": "MDoF/s", "cores": 1, "kind": "port", "host_cores": os.cpu_count(),
           "sample": f"{steps} implicit steps (steps 2..{steps + 1}) of the same {kind}{N} workload (NumPy/SciPy oracle: vectorised assembly, "
                     f"GMRES(30)+{pc} with the NumPy V-cycle, {sum(its[1:]) / max(len(its) - 1, 1):.1f} its/step), "
                     f"{sec:.2f} s/step; total {time.perf_counter() - t_start:.1f} s incl. setup",
           "omp_twin": None}
    try:
        import knpemi_cpu_twin as T
        cpu["omp_twin"] = T.time_kernels(o, lambda wrap: fac(None, wrap), rtol=rtol, budget_s=8.0)
    except Exception as exc:      # noqa: BLE001  (the twin is an extra leg; its absence must not take the bench line down)
        cpu["omp_twin"] = {"error": f"{type(exc).__name__}: {exc}"}
    par = None
    if snap is not None and osnap:
        s = float(np.abs(osnap["phi_m"]).max())
        # (1) same-algorithm trajectory: the oracle's GMRES with the same preconditioner construction, steps 1..snap
        par = {"step": snap["step"], "checker": "oracle GMRES with the same preconditioner algorithm, same rtol",
               "rel_err_phi_i_L2": abs(snap["phi_i"] - osnap["phi_i"]) / osnap["phi_i"],
               "rel_err_phi_e_L2": abs(snap["phi_e"] - osnap["phi_e"]) / osnap["phi_e"],
               "abs_err_phi_e_L2_over_phi_i_L2": abs(snap["phi_e"] - osnap["phi_e"]) / osnap["phi_i"],
               "rel_err_phi_m_max": float(np.abs(snap["phi_m"] - osnap["phi_m"]).max()) / s,
               "tol": PARITY_TOL,
               "phi_e_statement": "||phi_e||_L2 is matched to tol * ||phi_i||_L2 (1e-6 of the potential scale): phi_e is 100-2000x smaller than "
                                  "phi_i here, so the truncation of an rtol-%g solve is amplified in its own relative error (reported, not gated)" % rtol}
        ok = par["rel_err_phi_i_L2"] <= PARITY_TOL and par["rel_err_phi_m_max"] <= PARITY_TOL and par["abs_err_phi_e_L2_over_phi_i_L2"] <= PARITY_TOL
        # (2) preconditioner-independent: the oracle redoes the sampled step from the GPU's own state before it -- its A and b applied
        # to the GPU's solution (true residual per field block) and, in 2D, its sparse direct solve of that step
        if snap.get("state_before") is not None:
            want_lu = (not args.no_parity_lu) and kind == "square" and o.n_dof <= 1_200_000
            chk = K.single_step_check(mk(N, models=mdl), snap["state_before"], snap["x"], lu=want_lu)
            par["true_residual"] = {"rel_to_b": chk["rel_residual"], "max_block_backward_error": chk["max_backward"],
                                    "blocks": chk["blocks"], "gauge_drift": chk["gauge_drift"], "tol_rel_to_b": TRUE_RES_TOL,
                                    "tol_backward": BACKWARD_TOL,
                                    "what": "||b - A x_gpu|| with the ORACLE's A, b of the sampled step (assembled from the GPU's previous state)"}
            ok = ok and chk["rel_residual"] <= TRUE_RES_TOL and chk["max_backward"] <= BACKWARD_TOL
            if want_lu:
                par["direct_solve"] = {"checker": "oracle sparse LU (nested dissection) of the same step, same gauge", "lu_s": chk["lu_s"],
                                       "field_max_rel_diff": chk["lu_field_diff"], "rel_err_phi_i_L2": chk["rel_err_phi_i_L2"],
                                       "rel_err_phi_e_L2": chk["rel_err_phi_e_L2"], "abs_err_phi_e_L2_over_phi_i_L2": chk["abs_err_phi_e_over_phi_i"],
                                       "rel_err_phi_m_max": chk["rel_err_phi_m_max"], "lu_rel_residual": chk["lu_rel_residual"]}
                ok = ok and chk["rel_err_phi_i_L2"] <= PARITY_TOL and chk["rel_err_phi_m_max"] <= PARITY_TOL and \
                    chk["abs_err_phi_e_over_phi_i"] <= PARITY_TOL and max(chk["lu_field_diff"]) <= PARITY_TOL
        par["ok"] = bool(ok)
    return cpu, par


if __name__ == "__main__":
    main()
