"""Developer tool (GPU box, repo root): library preconditioner vs the oracle's restatement on the same hierarchies, Dirichlet config.
usage: python tests/devtools/pc_compare.py cube 6 btcc"""
import sys; sys.path.insert(0, 'tests'); import conftest  # noqa
import numpy as np, torch
import knpemi_oracle as K
from parity_utils import ci_config, make_problem
from CGx.KNPEMI.KNPEMIx_solver import SolverKNPEMI
kind, N, pc = sys.argv[1], int(sys.argv[2]), sys.argv[3]
cfg = ci_config(N=N, steps=2, rtol=1e-11, kind=kind, pc=pc)
cfg["dirichlet_bcs"] = True
cfg["initial_conditions"].update({"Na_i": 10, "Na_e": 145, "K_i": 130, "K_e": 3, "Cl_i": 5, "Cl_e": 134})
cfg["solver"]["ksp_settings"]["amg_fp32"] = False
p = make_problem(cfg)
s = SolverKNPEMI(p, solver_config=p.solver_config)
s.setup_solver(); be = s.backend
p.setup_preconditioner(s.use_block_Jacobi); s.assemble_preconditioner()
be.assemble_rhs(); be.assemble_matrix(); be.pc_setup(s._pc_kind)
print("stats", be.stats(), [h.describe()["rows"] for h in s.hierarchies], [h.node_fields for h in s.hierarchies])
params = K.Params(ki_init=K.OracleKNPEMI.REF_DEFAULT_KI, ke_init=K.OracleKNPEMI.REF_DEFAULT_KE)
o = (K.make_square if kind == "square" else K.make_cube)(N, models=K.CI_MODELS(), params=params)
x = o.coords / o.coords.max()
bv = np.nonzero(np.any((np.abs(x) < 1e-12) | (np.abs(x - 1.0) < 1e-12), axis=1))[0]
dofs, vals = o.dirichlet_initial_values(bv)
fused = bool(be.stats()["fused"])
if pc == "btcc":
    hk, hp = s.hierarchies
    M = K.pc_btcc(o, hk, hp, 1, 1, 1, bc_dofs=dofs, fused=fused)
else:
    h = s.hierarchy; M = K.pc_amg_vcycle(h.levels, h.coarse_inv, 1, 1, 1, fused=fused)
rng = np.random.default_rng(0)
r = rng.standard_normal(be.n_dof_owned)
bc = np.zeros(len(r), bool); bc[dofs] = True
if "--zero-bc" in sys.argv:
    r[bc] = 0.0
z = torch.zeros(be.n_dof_owned, dtype=torch.float64, device=be.device)
prev = None
for rep in range(3):
    z.fill_(float(rep) * 1e30)                  # stale output entries would show
    be.pc_apply(torch.as_tensor(r, device=be.device), z)
    cur = z.cpu().numpy().copy()
    if prev is not None:
        print("call", rep, "vs previous: max abs diff", np.abs(cur - prev).max(), "entries differing", int((cur != prev).sum()))
    prev = cur
zg = z.cpu().numpy(); zo = M(r)
for f in range(4):
    for name, m in (("bc", bc[f::4]), ("free", ~bc[f::4])):
        d = np.abs(zg[f::4] - zo[f::4])[m]
        print(f, name, "max diff", d.max() if d.size else 0, "scale", np.abs(zo[f::4][m]).max() if d.size else 0)

# GMRES of the oracle driven by the LIBRARY's operators (SpMV and preconditioner): separates operator bugs from solver-loop bugs
if "--gmres" in sys.argv:
    bt = be.b.cpu().numpy().copy()
    def Aop(v):
        y = torch.zeros_like(z); be.spmv(torch.as_tensor(v, device=be.device), y); return y.cpu().numpy()
    def Mop(v):
        y = torch.zeros_like(z); be.pc_apply(torch.as_tensor(v, device=be.device), y); return y.cpu().numpy()
    class Op:
        def __matmul__(self, v): return Aop(v)
    x0 = be.x.cpu().numpy().copy()
    xg, its, res = K.gmres_left(Op(), bt, x0, Mop, rtol=1e-11, max_it=200)
    print("oracle GMRES on library operators: its", its, "res", res)
    xo2, its2, res2 = K.gmres_left(Op(), bt, x0, M, rtol=1e-11, max_it=200)
    print("oracle GMRES, library A, oracle PC: its", its2, "res", res2)
    itn, rn, reason = be.gmres(1e-11, 1e-50, 200, 30)
    print("library GMRES: its", itn, "res", rn, "reason", reason)
