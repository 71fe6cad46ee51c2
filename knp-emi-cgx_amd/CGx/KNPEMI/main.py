"""Command-line driver with the reference's interface (src/CGx/KNPEMI/main.py:12-106):

    python -m CGx.KNPEMI.main --config <file.yaml> [--view 1]

Same construction order (problem -> membrane mechanisms chosen from the config name / glia flag -> initial conditions ->
init_ionic_models -> setup_variational_form -> SolverKNPEMI(problem, solver_config).solve()), same printed lines
("Variational form setup in ...", "L2 norm phi_i = ...", "L2 norm phi_e = ...", "Total script time: ...").
One process per GPU: start it under ``python -m torch.distributed.run --nproc-per-node N`` for N GPUs.
"""
from __future__ import annotations

import argparse
import os
import time
from pathlib import Path


def _init_distributed():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("KNP_DIST_BACKEND", "nccl")
    dev = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend=backend, rank=rank, world_size=world)


def main_yaml(yaml_file: str = "config.yaml", view_ksp: bool = False):
    """Main for running scripts with a yaml/yml configuration file (reference main.py:12-87)."""
    from cgx_hip.configs import default_ionic_models
    from CGx.KNPEMI.KNPEMIx_problem import ProblemKNPEMI
    from CGx.KNPEMI.KNPEMIx_solver import SolverKNPEMI

    problem = ProblemKNPEMI(yaml_file)
    ionic_models = default_ionic_models(problem, yaml_file)           # reference main.py:27-45
    problem.set_initial_conditions()
    problem.init_ionic_models(ionic_models)

    tic = time.perf_counter()
    problem.setup_variational_form()
    t_form = problem.comm.allreduce_max(time.perf_counter() - tic)
    problem.print(f"Variational form setup in {t_form:0.4f} seconds")

    solver_config = problem.solver_config
    solver_config["view_ksp"] = view_ksp
    solver = SolverKNPEMI(problem, solver_config=solver_config)
    solver.solve()

    phi_i_L2, phi_e_L2 = solver.potential_norms()                     # reference main.py:65-87
    problem.print(f"L2 norm phi_i = {phi_i_L2}")
    problem.print(f"L2 norm phi_e = {phi_e_L2}")
    return solver


def main(argv=None):
    parser = argparse.ArgumentParser(description="KNP-EMI on MI355X: the reference's main.py interface")
    parser.add_argument("--config", dest="config_file", type=Path, required=True, help="Configuration file")
    parser.add_argument("--view", dest="view_ksp", default=0, type=int, help="Verbose KSP object log")
    args = parser.parse_args(argv)
    _init_distributed()
    tic = time.perf_counter()
    solver = main_yaml(yaml_file=str(args.config_file), view_ksp=bool(args.view_ksp))
    script_time = solver.comm.allreduce_max(time.perf_counter() - tic)
    solver.print(f"Total script time: {script_time:0.4f} seconds")
    return solver


if __name__ == "__main__":
    main()
