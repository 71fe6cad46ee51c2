"""Developer tool (not part of the product or the tests); run on a GPU box from the repo root."""
import sys, time; sys.path.insert(0,'tests'); import conftest
import numpy as np
from parity_utils import *
N=int(sys.argv[1]); steps=int(sys.argv[2])
extra = dict(a.split("=") for a in sys.argv[3:])
cfg = ci_config(N=N, steps=steps, rtol=1e-9, kind="cube", pc="btcc")
for k,v in extra.items(): cfg["solver"]["ksp_settings"][k] = float(v) if "." in v else int(v)
s = run_native(cfg)
print("N",N, extra, "its", s.iterations, "solve ms", [round(t*1e3,1) for t in s.solve_time], "amg setup s", round(s.amg_setup_time,1))
for h in s.hierarchies: print("  ", h.describe())
