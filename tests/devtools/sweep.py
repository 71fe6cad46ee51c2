"""Developer tool (not part of the product or the tests); run on a GPU box from the repo root."""
import sys, time; sys.path.insert(0,'tests'); import conftest
import numpy as np, torch
from parity_utils import *
kind=sys.argv[1]; N=int(sys.argv[2]); pc=sys.argv[3]
for extra in [dict(), dict(amg_cheby_degree=3), dict(amg_cheby_degree=1), dict(amg_pre=2, amg_post=2), dict(amg_pre=0, amg_post=1), dict(amg_theta=0.04), dict(amg_theta=0.16), dict(gmres_restart=50)]:
    cfg = ci_config(N=N, steps=6, rtol=1e-9, kind=kind, pc=pc)
    cfg["solver"]["ksp_settings"].update(extra)
    s = run_native(cfg)
    st = np.array(s.solve_time[2:])*1e3
    print(kind, N, pc, extra, "its", s.iterations, "solve ms", round(st.mean(),3), flush=True)
    del s; torch.cuda.empty_cache()
