"""Developer tool (not part of the product or the tests); run on a GPU box from the repo root."""
import sys; sys.path.insert(0,'tests'); import conftest
import numpy as np
from parity_utils import *
for rtol in (1e-9, 1e-11, 1e-13):
    cfg = ci_config(N=16, steps=2, rtol=rtol); cfg["solver"]["ksp_settings"]["ksp_max_it"]=2000
    s = run_native(cfg)
    o = run_oracle(N=16, steps=2)
    vi = o.lay.node_i>=0
    d = [np.max(np.abs(s.problem.wh[0][j].numpy()[vi]-o.k[0][j][vi])/o.k[0][j][vi]) for j in range(3)]
    ni,ne = s.potential_norms(); oi,oe=o.potential_norms()
    print(rtol, "its", s.iterations, "reasons", s.reasons, "rnorm", s.ksp.rnorm, "k relerr", d, "phi", (ni-oi)/oi, (ne-oe)/oe, flush=True)
    print(s.hierarchy.describe())
