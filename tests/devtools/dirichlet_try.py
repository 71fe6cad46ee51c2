"""Developer tool (GPU box): Dirichlet run with solver settings from the command line: python tests/devtools/dirichlet_try.py cube 6 btcc amg_split_decoupled=False"""
import sys; sys.path.insert(0, 'tests'); import conftest  # noqa
from parity_utils import ci_config, run_native
kind, N, pc = sys.argv[1], int(sys.argv[2]), sys.argv[3]
cfg = ci_config(N=N, steps=2, rtol=1e-11, kind=kind, pc=pc)
cfg["dirichlet_bcs"] = True
cfg["initial_conditions"].update({"Na_i": 10, "Na_e": 145, "K_i": 130, "K_e": 3, "Cl_i": 5, "Cl_e": 134})
cfg["solver"]["ksp_settings"]["amg_fp32"] = False
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    cfg["solver"]["ksp_settings"][k] = {"True": True, "False": False}.get(v, v)
s = run_native(cfg)
print(sys.argv[4:], "its", s.iterations, "reasons", s.reasons, s.backend.stats()["fused"], [h.describe()["rows"] for h in s.hierarchies])
