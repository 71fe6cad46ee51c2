"""Developer tool: MMS run with solver settings from the command line: python tests/devtools/mms_try.py 2 32 amg_setup=host ..."""
import sys; sys.path.insert(0, 'tests'); import conftest  # noqa
import numpy as np
from parity_utils import mms_config
from CGx.KNPEMI.KNPEMIx_ionic_model import PassiveModel
from CGx.KNPEMI.KNPEMIx_problem import ProblemKNPEMI
from CGx.KNPEMI.KNPEMIx_solver import SolverKNPEMI
dim, N = int(sys.argv[1]), int(sys.argv[2])
cfg = mms_config(dim, N)
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    cfg["solver"]["ksp_settings"][k] = {"True": True, "False": False}.get(v, v)
p = ProblemKNPEMI(cfg)
p.set_initial_conditions(); p.init_ionic_models([PassiveModel(p)]); p.setup_variational_form()
p.solver_config["view_ksp"] = False
s = SolverKNPEMI(p, solver_config=p.solver_config)
s.solve()
print(sys.argv[3:], "its", s.iterations, "reasons", s.reasons, "stats", s.backend.stats(), "nf", [getattr(h, "node_fields", None) for h in s.hierarchies], [h.describe()["rows"] for h in s.hierarchies])
