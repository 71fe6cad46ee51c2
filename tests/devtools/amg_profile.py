"""Developer tool: cProfile of the host AMG setup on a cube<N> ion block (no GPU needed)."""
import sys, time, cProfile, pstats; sys.path[:0] = ['knp-emi-cgx_amd', 'oracle', 'tests']
import numpy as np
import knpemi_oracle as K
from cgx_hip import amg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
o = K.make_cube(N, models=K.CI_MODELS())
P = o.assemble_P()
Pk = amg.restrict_to_fields(P, (0, 1, 2))
print("n", Pk.shape[0], "nnz", Pk.nnz, flush=True)
pr = cProfile.Profile(); pr.enable()
t = time.time(); h = amg.build_hierarchy(Pk, theta=0.08, coarse_size=2500); print("build", time.time() - t, h.describe())
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(16)
