"""Developer tool: compare the matrices written by the assembly kernel variants (GPU box, repo root)."""
import os, sys; sys.path.insert(0, 'tests'); import conftest  # noqa
import numpy as np
from test_gpu_parity import _setup
N, kind = int(sys.argv[1]), sys.argv[2]
def build():
    p, be, o = _setup(N, kind)
    be.assemble_matrix(); be.assemble_precond()
    return be.csr(), be.precond_csr()
A1, P1 = build()
A1b, P1b = build()
print("repeat same variant:", np.abs(A1.data - A1b.data).max(), np.abs(P1.data - P1b.data).max())
os.environ["KNP_ASM_TRANSPOSED"] = "0"
A2, P2 = build()
os.environ["KNP_ASM_STAGE"] = "0"
A3, P3 = build()
for nm, A, P in (("staged", A2, P2), ("plain", A3, P3)):
    dA = np.abs(A.data - A1.data); dP = np.abs(P.data - P1.data)
    print(nm, "A max diff", dA.max(), "rel", dA.max() / np.abs(A1.data).max(), "n", int((dA > 0).sum()), "| P", dP.max(), int((dP > 0).sum()))
d23 = np.abs(A2.data - A3.data); print("staged vs plain", d23.max(), int((d23 > 0).sum()))
