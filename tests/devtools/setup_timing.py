"""Developer tool (GPU box): where the one-off setup time goes for a cube<N> workload."""
import sys, time; sys.path[:0] = ['tests', 'oracle', 'knp-emi-cgx_amd']
import conftest  # noqa
import torch
from parity_utils import ci_config, make_problem
from cgx_hip.parallel import stacked_cubes_local_mesh
from cgx_hip import amg, amg_gpu, _lib
from CGx.KNPEMI.KNPEMIx_solver import SolverKNPEMI
N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
T = {}
def tick(name, t0):
    torch.cuda.synchronize(); T[name] = time.perf_counter() - t0; print(f"{name:28s} {T[name]:7.2f} s", flush=True)
t = time.perf_counter(); lm = stacked_cubes_local_mesh(N, 1, 0, scale=1e-6); tick("mesh generation", t)
t = time.perf_counter(); p = make_problem(ci_config(N=N, steps=1, kind="cube", pc="btcc"), local_mesh=lm); tick("problem + programs", t)
p.solver_config["view_ksp"] = False
s = SolverKNPEMI(p, solver_config=p.solver_config)
t = time.perf_counter(); s.setup_solver(); tick("backend (graph build, upload)", t)
be = s.backend
t = time.perf_counter(); be.assemble_precond(); P = be.precond_csr(); tick("assemble P + fetch CSR", t)
t = time.perf_counter(); Pk = amg.restrict_to_fields(P, (0, 1, 2)); Pp = amg.restrict_to_fields(P, (3,)); tick("field restriction", t)
mode = sys.argv[2] if len(sys.argv) > 2 else "gpu"
bh = (lambda M, nf=None: amg_gpu.build_hierarchy(M, theta=s.amg_theta, coarse_size=s.amg_coarse_size, device=be.device, node_fields=nf)) if mode == "gpu" else (lambda M, nf=None: amg.build_hierarchy(M, theta=s.amg_theta, coarse_size=s.amg_coarse_size, node_fields=nf))
DIST = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2, 1]
bh_ion = (lambda M, nf: amg_gpu.build_hierarchy(M, theta=s.amg_theta, coarse_size=s.amg_coarse_size, device=be.device, node_fields=nf, agg_distance=DIST)) if mode == "gpu" else (lambda M, nf: amg.build_hierarchy(M, theta=s.amg_theta, coarse_size=s.amg_coarse_size, node_fields=nf, agg_distance=DIST))
t = time.perf_counter(); hk = bh_ion(Pk, (4, (0, 1, 2))); tick(f"hierarchy ions ({mode}, distance {DIST})", t)
t = time.perf_counter(); be.set_coupled_potential(True); be.assemble_precond(); Pc = be.precond_phi_csr(); tick("coupled potential block: assemble + fetch", t)
t = time.perf_counter(); hp = bh(Pc); tick(f"hierarchy potential ({mode})", t)
print(hk.describe(), hp.describe())
class TimedLib:      # per-entry-point time of the upload
    def __init__(self, lib): self.lib, self.t = lib, {}
    def __getattr__(self, name):
        f = getattr(self.lib, name)
        def w(*a):
            t0 = time.perf_counter(); r = f(*a); self.t[name] = self.t.get(name, 0.0) + time.perf_counter() - t0; return r
        return w
tl = TimedLib(be.lib)
t = time.perf_counter(); amg.upload(tl, be.ctx, be.check, hk, 1, 1, 1, index=0, level0_native=True); amg.upload(tl, be.ctx, be.check, hp, 1, 1, 1, index=1); tick("upload", t)
print("  inside the library:", {k: round(v, 3) for k, v in tl.t.items()})
be.check(be.lib.knp_amg_use_native_level0(be.ctx, 0, 2)); be.check(be.lib.knp_amg_use_native_level0(be.ctx, 1, 4))
t = time.perf_counter(); be.check(be.lib.knp_pc_setup(be.ctx, _lib.PC_AMG_BT)); tick("knp_pc_setup", t)
print("n_dof", be.n_dof_owned, "total", sum(T.values()))
