"""Developer check (GPU box): btcc with the coupled / uncoupled potential hierarchy on a workload: iterations, ms per step, hierarchies.
usage: python tests/devtools/chk_coupled.py <workload> [0|1]"""
import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); import conftest  # noqa
import re, torch
from cgx_hip.configs import ci_config, make_problem, tissue_config
from CGx.KNPEMI.KNPEMIx_solver import SolverKNPEMI
w = sys.argv[1]; coupled = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mt = re.fullmatch(r"tissue(\d)d_(\d+)_(\d+)_w(\d+)", w)
if mt:
    cfg = tissue_config(int(mt.group(1)), int(mt.group(2)), int(mt.group(3)), steps=12, rtol=1e-9, pc="btcc", stimulus=True, width=int(mt.group(4)))
else:
    m = re.fullmatch(r"(square|cube)(\d+)", w)
    cfg = ci_config(N=int(m.group(2)), steps=12, rtol=1e-9, kind=m.group(1), pc="btcc")
cfg["solver"]["ksp_settings"]["btcc_coupled_phi"] = bool(coupled)
p = make_problem(cfg); s = SolverKNPEMI(p, solver_config=p.solver_config)
s.prepare()
for i in range(1, 5): s.step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(5, 13): s.step(i)
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 8
print(w, "coupled", s._coupled_phi, "ms/step %.3f" % (el * 1e3), "its", s.iterations, "fused", s.backend.stats()["fused"])
for h in s.hierarchies:
    d = h.describe(); print("   rows", d["rows"], "nnz", d["nnz"], "opc %.2f" % d["operator_complexity"])
