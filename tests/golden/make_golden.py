"""Generates the golden vectors under tests/golden/ from the CPU oracle.

The reference itself cannot run here (DOLFINx / multiphenicsx / PETSc are not installed), so
the fixtures come from this repo's oracle *after* it reproduced the reference's own known answers
(tests/test_oracle_pins.py).  Fixtures are data only: inputs are fully described by (kind, N, steps) and the
CI physics; outputs are the assembled system of step 1, the right-hand side, the solution after every step
and the potential norms.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import knpemi_oracle as K  # noqa: E402


def make(kind, N, steps, models):
    mk = K.make_square if kind == "square" else K.make_cube
    o = mk(N, models=K.CI_MODELS() if models == "ci" else [K.Model("passive", (4,))])
    out = {"kind": kind, "N": N, "steps": steps, "models": models}
    # step-1 system
    o2 = mk(N, models=K.CI_MODELS() if models == "ci" else [K.Model("passive", (4,))])
    o2.t += o2.p.dt
    for m in o2.models:
        if m.kind == "hh":
            o2.update_t_mod()
            o2.update_gating(m)
    A = o2.assemble_A()
    P = o2.assemble_P()
    out.update(A_indptr=A.indptr, A_indices=A.indices, A_data=A.data, b=o2.assemble_b(),
               P_indptr=P.indptr, P_indices=P.indices, P_data=P.data, n_gate=o2.n, m_gate=o2.m, h_gate=o2.h)
    xs, norms = [], []

    def log(step, oo, x):
        xs.append(x.copy())
        norms.append(oo.potential_norms())
    o.run(steps, solver="lu_gauge", log=log)
    out.update(x=np.array(xs), norms=np.array(norms), phi_m=o.phi_m)
    np.savez_compressed(os.path.join(HERE, f"{kind}{N}_{models}.npz"), **out)
    print(kind, N, models, "n_dof", o.n_dof, "norms", norms[-1])


if __name__ == "__main__":
    make("square", 8, 3, "ci")
    make("square", 8, 3, "passive")
    make("cube", 4, 2, "ci")
