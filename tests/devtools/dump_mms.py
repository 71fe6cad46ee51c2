"""Developer tool: dump A, P, b of the MMS problem (run on a GPU box from the repo root): python tests/devtools/dump_mms.py 2 32"""
import sys; sys.path.insert(0, 'tests'); import conftest  # noqa
import numpy as np, scipy.sparse as sp
from parity_utils import mms_config
from CGx.KNPEMI.KNPEMIx_ionic_model import PassiveModel
from CGx.KNPEMI.KNPEMIx_problem import ProblemKNPEMI
from CGx.KNPEMI.KNPEMIx_solver import SolverKNPEMI
dim, N = int(sys.argv[1]), int(sys.argv[2])
p = ProblemKNPEMI(mms_config(dim, N))
p.set_initial_conditions(); p.init_ionic_models([PassiveModel(p)]); p.setup_variational_form()
s = SolverKNPEMI(p, solver_config=p.solver_config)
s.setup_solver()
be = s.backend
p.setup_preconditioner(s.use_block_Jacobi)
be.assemble_precond()
P = be.precond_csr()
be.assemble_rhs(); be.assemble_matrix()
A = be.csr()
sp.save_npz("gpurun_out/mms_P.npz", sp.csr_matrix(P)); sp.save_npz("gpurun_out/mms_A.npz", sp.csr_matrix(A))
np.save("gpurun_out/mms_b.npy", be.b.cpu().numpy())
