"""Developer tool (GPU box): where the HOST time of a timestep goes (cProfile over N steps of the drop-in loop body)."""
import cProfile, pstats, sys, os, io, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "knp-emi-cgx_amd"))
import torch
from cgx_hip.configs import ci_config, make_problem
from CGx.KNPEMI.KNPEMIx_solver import SolverKNPEMI
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
p = make_problem(ci_config(N=N, steps=steps + 10, rtol=1e-9))
p.solver_config["view_ksp"] = False
s = SolverKNPEMI(p, solver_config=p.solver_config)
s.prepare()
for i in range(1, 11):
    s.step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(11, 11 + steps):
    s.step(i)
pr.disable()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"{el / steps * 1e6:.1f} us per step under cProfile")
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(22)
print(out.getvalue())
