"""Developer tool (GPU box): wall time of back-to-back preconditioner applications (one sync at the end) vs the number of
kernel launches -- tells whether the host's launch rate or the GPU limits the V-cycle."""
import sys, time, ctypes as C; sys.path[:0] = ['tests', 'oracle', 'knp-emi-cgx_amd']
import conftest  # noqa
import torch
from parity_utils import ci_config, make_problem
from CGx.KNPEMI.KNPEMIx_solver import SolverKNPEMI
kind, N, pc = sys.argv[1], int(sys.argv[2]), sys.argv[3]
p = make_problem(ci_config(N=N, steps=2, kind=kind, pc=pc)); p.solver_config["view_ksp"] = False
s = SolverKNPEMI(p, solver_config=p.solver_config); s.solve()
be = s.backend
r = torch.randn_like(be.x); z = torch.zeros_like(be.x)
for reps in (1, 50, 200):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        be.lib.knp_pc_apply(be.ctx, C.c_void_p(r.data_ptr()), C.c_void_p(z.data_ptr()))
    t_enq = time.perf_counter() - t
    torch.cuda.synchronize(); t_all = time.perf_counter() - t
    print(f"{kind}{N} {pc}: reps {reps:4d}  enqueue {1e6 * t_enq / reps:8.1f} us/apply   total {1e6 * t_all / reps:8.1f} us/apply", flush=True)
