"""Developer tool: time the distributed AMG setup (cgx_hip/dist_amg.build_distributed_hierarchy) and a few timesteps with
several ranks sharing one GPU (gloo rendezvous, native peer-to-peer exchange).
usage: python tools/dist_setup_timing.py <ranks> <workload, e.g. square512 | cube64> [steps]"""
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker():
    sys.path[:0] = [os.path.join(ROOT, "knp-emi-cgx_amd")]
    import re
    import torch
    import torch.distributed as dist
    rank, size = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    from cgx_hip.configs import ci_config, make_problem
    from cgx_hip.parallel import stacked_cubes_local_mesh, stacked_squares_local_mesh
    from CGx.KNPEMI.KNPEMIx_solver import SolverKNPEMI
    kind, N = re.fullmatch(r"(square|cube)(\d+)", sys.argv[2]).groups()
    N = int(N)
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    gen = stacked_squares_local_mesh if kind == "square" else stacked_cubes_local_mesh
    t0 = time.perf_counter()
    lm = gen(N, size, rank, scale=1e-6)
    p = make_problem(ci_config(N=N, steps=steps, rtol=1e-9, pc="hypre" if kind == "square" else "btcc", kind=kind), local_mesh=lm)
    s = SolverKNPEMI(p, solver_config=p.solver_config)
    t1 = time.perf_counter()
    s.solve()
    t2 = time.perf_counter()
    if rank == 0:
        print(f"ranks {size} x {kind}{N}: problem {t1 - t0:.2f} s, solve() {t2 - t1:.2f} s of which AMG setup {s.amg_setup_time:.2f} s, "
              f"setup total {s.setup_time:.2f} s; its {s.iterations}; solve times {[round(t, 4) for t in s.solve_time]}; "
              f"native exchange {getattr(s.backend, 'p2p_on', False)}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    if "RANK" in os.environ:
        worker()
    else:
        n = int(sys.argv[1])
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                  env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                                           HSA_ENABLE_IPC_MODE_LEGACY="0")) for r in range(n)]
        sys.exit(max(p.wait() for p in procs))
