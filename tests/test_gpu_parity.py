"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Tolerances (fp64 everywhere): assembled operators and vectors 1e-12 relative to the largest entry;
solutions after time stepping 1e-6 relative on the potential norms / membrane potential
(BASELINE.json: "matching reference potentials to rtol 1e-6").
"""
import numpy as np
import pytest
import torch

from parity_utils import ci_config, fp32_stored, make_oracle, make_problem, run_native, run_oracle

pytestmark = pytest.mark.gpu

PIN_ITERATIVE = (3.510994056704844e-08, 6.369472309249516e-11)   # reference tests/KNPEMI/electric_potential_norms_iterative_solver.py:58-59


def _rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def _setup(N, kind, models="ci", perturb=True):
    cfg = ci_config(N=N, steps=1, kind=kind)
    p = make_problem(cfg, models=models)
    be = p.create_backend()
    o = make_oracle(N, kind, models)
    if perturb:
        # non-uniform but smooth previous state so that every term of the forms is exercised
        X = o.coords / o.coords.max()
        s = 1.0 + 0.05 * np.sin(3.0 * X[:, 0] + 1.0) * np.cos(2.0 * X[:, 1] + 0.5)
        for side in range(2):
            for j in range(3):
                o.k[side][j] = o.k[side][j] * (s if (side + j) % 2 == 0 else 2.0 - s)
                p.wh[side][j].x.array[:] = torch.as_tensor(o.k[side][j], device=p.mesh.device)
        o.phi_m = o.phi_m * (2.0 - s)
        p.phi_m_prev.x.array[:] = torch.as_tensor(o.phi_m, device=p.mesh.device)
        for name in ("n", "m", "h"):
            if hasattr(p, name):
                setattr(o, name, getattr(o, name) * s)
                getattr(p, name).x.array[:] = torch.as_tensor(getattr(o, name), device=p.mesh.device)
    return p, be, o


@pytest.mark.parametrize("N,kind", [(8, "square"), (12, "square"), (4, "cube")])
def test_layout_and_pattern(N, kind):
    p, be, o = _setup(N, kind, perturb=False)
    assert be.n_dof_owned == o.n_dof
    assert np.array_equal(be.node_i, o.lay.node_i.astype(np.int32))
    assert np.array_equal(be.node_e, o.lay.node_e.astype(np.int32))


@pytest.mark.parametrize("N,kind", [(8, "square"), (16, "square"), (4, "cube"), (8, "cube")])
def test_matrix_rhs_precond_match_oracle(N, kind):
    p, be, o = _setup(N, kind)
    o.t = o.p.dt
    o.update_t_mod()
    p.t.value = o.p.dt
    for m in p.ionic_models:
        if hasattr(m, "update_t_mod"):
            m.update_t_mod()
    be.assemble_matrix()
    A = be.csr()
    Ao = o.assemble_A()
    assert A.shape == Ao.shape
    D = (A - Ao).tocoo()
    assert (np.abs(D.data).max() if D.nnz else 0.0) <= 1e-12 * np.abs(Ao.data).max()
    # structural pattern identical (explicit zeros kept on both sides are allowed to differ)
    be.assemble_rhs()
    b = be.b.cpu().numpy()
    bo = o.assemble_b()
    assert _rel(b, bo) <= 1e-12
    # each block row separately (potential rows are 1e6 smaller than concentration rows)
    for f in range(4):
        assert _rel(b[f::4], bo[f::4]) <= 1e-10, f
    be.assemble_precond()
    P = be.precond_csr()
    Po = o.assemble_P()
    D = (P - Po).tocoo()
    assert (np.abs(D.data).max() if D.nnz else 0.0) <= 1e-12 * np.abs(Po.data).max()


@pytest.mark.parametrize("N,kind", [(16, "square"), (6, "cube")])
def test_spmv_and_nullspace(N, kind):
    p, be, o = _setup(N, kind)
    be.assemble_matrix()
    Ao = o.assemble_A()
    rng = np.random.default_rng(0)
    x = rng.standard_normal(o.n_dof)
    xt = torch.as_tensor(x, device=be.device)
    yt = torch.empty_like(xt)
    be.spmv(xt, yt)
    y = yt.cpu().numpy()
    yo = Ao @ x
    assert _rel(y, yo) <= 1e-13
    assert be.nullspace_test() <= 1e-10 * np.abs(Ao.data).max()
    # projection removes the mean of the potential entries only
    v = torch.as_tensor(x.copy(), device=be.device)
    be.project_nullspace(v)
    ns = o.nullspace()
    vo = x - ns * (ns @ x)
    assert _rel(v.cpu().numpy(), vo) <= 1e-13


def test_hh_gating_update_matches_oracle():
    import knpemi_oracle as K
    p, be, o = _setup(8, "square")
    mdl = [m for m in o.models if m.kind == "hh"][0]
    o.update_gating(mdl)
    hh = [m for m in p.ionic_models if hasattr(m, "update_gating_variables")][0]
    hh.update_gating_variables()
    for name in ("n", "m", "h"):
        assert _rel(getattr(p, name).numpy(), getattr(o, name)) <= 1e-12, name
    # forward-Euler variant
    hh.use_Rush_Larsen = False
    mdl.use_rush_larsen = False
    o.update_gating(mdl)
    hh.update_gating_variables()
    for name in ("n", "m", "h"):
        assert _rel(getattr(p, name).numpy(), getattr(o, name)) <= 1e-12, name


def test_pc_apply_vbjacobi_inverts_vertex_blocks():
    p, be, o = _setup(8, "square")
    be.assemble_matrix()
    be.pc_setup(1)
    A = be.csr().tocsr()
    rng = np.random.default_rng(1)
    r = rng.standard_normal(o.n_dof)
    rt = torch.as_tensor(r, device=be.device)
    zt = torch.zeros_like(rt)
    be.pc_apply(rt, zt)
    z = zt.cpu().numpy()
    grp = o.lay.node_vertex
    starts = np.nonzero(np.r_[True, grp[1:] != grp[:-1]])[0]
    sizes = np.diff(np.r_[starts, o.lay.n_nodes])
    for s, sz in zip(starts, sizes):
        idx = np.arange(4 * s, 4 * (s + sz))
        blk = A[idx][:, idx].toarray()
        assert np.allclose(blk @ z[idx], r[idx], rtol=1e-9, atol=1e-12 * np.abs(r).max())


@pytest.mark.parametrize("pc", ["hypre", "vbjacobi"])
def test_two_steps_match_oracle(pc):
    cfg = ci_config(N=16, steps=2, rtol=1e-13 if pc == "hypre" else 1e-14, pc=pc)
    cfg["solver"]["ksp_settings"]["ksp_max_it"] = 20000
    s = run_native(cfg)
    o = run_oracle(N=16, steps=2)
    ni, ne = s.potential_norms()
    oi, oe = o.potential_norms()
    # Vertex-block Jacobi has no coarse space: the near-null "each side's potential floats" modes
    # converge slowly in the Jacobi-preconditioned norm, so its potentials are only good to ~1e-4
    # even at rtol 1e-14 ("Preconditioners").  AMG on P meets the 1e-6 bar.
    tol = 1e-6 if pc == "hypre" else 1e-4
    assert abs(ni - oi) <= tol * oi
    phim = s.problem.phi_m_prev.numpy()
    gam = (o.lay.node_i >= 0) & (o.lay.node_e >= 0)
    assert np.allclose(phim[gam], o.phi_m[gam], rtol=tol)
    for j in range(3):
        vi = o.lay.node_i >= 0
        ve = o.lay.node_e >= 0
        # the preconditioned-residual stopping rule (reference: ksp_norm_type preconditioned) bounds the
        # concentration error only up to the non-normality of P^-1 A (||P_kk^-1 A_kphi|| ~ c/psi ~ 4e3)
        assert np.allclose(s.problem.wh[0][j].numpy()[vi], o.k[0][j][vi], rtol=1e-7)
        assert np.allclose(s.problem.wh[1][j].numpy()[ve], o.k[1][j][ve], rtol=1e-7)
    if pc == "hypre":
        assert abs(ne - oe) <= 1e-5 * oe          # rtol 1e-13 against sparse LU (smoke at rtol 1e-11 reaches 2.6e-6)


@pytest.mark.parametrize("fp32", [True, False])
@pytest.mark.parametrize("N,kind,steps", [(32, "square", 3), (8, "cube", 2)])
def test_gmres_amg_iterates_match_oracle_gmres(N, kind, steps, fp32):
    """Same algorithm on both sides: the oracle's PETSc-style GMRES(30) with the NumPy V-cycle applied to the
    *same* hierarchy data must take the same number of iterations and land on the same iterate.  With the
    default mixed-precision storage the oracle gets the same fp32-rounded operator values."""
    import knpemi_oracle as K
    cfg = ci_config(N=N, steps=steps, rtol=1e-9, kind=kind)
    cfg["solver"]["ksp_settings"]["amg_coarse_size"] = 200      # force a real multilevel cycle on this small mesh
    cfg["solver"]["ksp_settings"]["amg_fp32"] = fp32
    s = run_native(cfg)
    h = fp32_stored(s.hierarchy) if fp32 else s.hierarchy
    assert len(h.levels) >= 2
    o = make_oracle(N, kind)
    xo, its = o.run(steps, solver="gmres", rtol=1e-9,
                    pc=lambda P: K.pc_amg_vcycle(h.levels, h.coarse_inv, s.amg_pre, s.amg_post, s.amg_cheby_degree,
                                                 fused=bool(s.backend.stats()["fused"])))
    assert its == list(s.iterations)
    x = s.backend.x.cpu().numpy()
    for f in range(4):
        assert np.max(np.abs(x[f::4] - xo[f::4])) <= 1e-9 * np.max(np.abs(xo[f::4])), f


@pytest.mark.parametrize("fp32", [True, False])
@pytest.mark.parametrize("N,kind,steps", [(32, "square", 3), (8, "cube", 2)])
def test_btcc_iterates_match_oracle(N, kind, steps, fp32):
    """Block-triangular preconditioner (pc_type btcc): same algorithm restated in NumPy on the same two
    hierarchies -> same iteration counts, same iterates; and it converges to the LU solution."""
    import knpemi_oracle as K
    cfg = ci_config(N=N, steps=steps, rtol=1e-9, kind=kind, pc="btcc")
    cfg["solver"]["ksp_settings"]["amg_coarse_size"] = 150
    cfg["solver"]["ksp_settings"]["amg_fp32"] = fp32
    s = run_native(cfg)
    hk, hp = s.hierarchies
    if fp32:
        hk, hp = fp32_stored(hk, coarse=bool(s.backend.stats()["fused"])), fp32_stored(hp, level0_uploaded=s._coupled_phi)
    assert len(hk.levels) >= 2 and len(hp.levels) >= 2
    o = make_oracle(N, kind)
    xo, its = o.run(steps, solver="gmres", rtol=1e-9,
                    pc=lambda P: K.pc_btcc(o, hk, hp, s.amg_pre, s.amg_post, s.amg_cheby_degree, fused=bool(s.backend.stats()["fused"])))
    assert its == list(s.iterations)
    assert max(its) <= (6 if kind == "square" else 20)
    x = s.backend.x.cpu().numpy()
    for f in range(4):
        assert np.max(np.abs(x[f::4] - xo[f::4])) <= 1e-9 * np.max(np.abs(xo[f::4])), f
    ol = run_oracle(N=N, steps=steps, kind=kind)
    ni, ne = s.potential_norms()
    oi, oe = ol.potential_norms()
    assert abs(ni - oi) <= 2e-6 * oi


def test_ci_problem_against_reference_pins():
    """The reference's own CI problem (32x32, 10 steps, GMRES rtol 1e-9, AMG in place of BoomerAMG)."""
    s = run_native(ci_config(N=32, steps=10, rtol=1e-9))
    ni, ne = s.potential_norms()
    # phi_i: within the north-star tolerance of the reference's saved value
    assert abs(ni - PIN_ITERATIVE[0]) <= 1e-6 * PIN_ITERATIVE[0]
    # phi_e is 1000x smaller and only determined up to the linear-solver truncation error of the
    # reference run itself (see tests/test_oracle_pins.py::test_iterative_pin_noise_floor)
    assert abs(ne - PIN_ITERATIVE[1]) <= 3e-4 * PIN_ITERATIVE[1]
    # iteration count comparable to the reference's 3.0 (hypre) -- informational bound
    assert np.mean(s.iterations) <= 6.0
    # gauge: sum of potential dofs conserved
    x = s.backend.x.cpu().numpy()
    assert abs(x[3::4].sum() - (-0.07 * 289)) <= 1e-9 * 0.07 * 289


def test_cube_run_matches_oracle():
    cfg = ci_config(N=8, steps=2, rtol=1e-13, kind="cube")
    s = run_native(cfg)
    o = run_oracle(N=8, steps=2, kind="cube")
    ni, ne = s.potential_norms()
    oi, oe = o.potential_norms()
    assert abs(ni - oi) <= 1e-6 * oi
    gam = (o.lay.node_i >= 0) & (o.lay.node_e >= 0)
    assert np.allclose(s.problem.phi_m_prev.numpy()[gam], o.phi_m[gam], rtol=1e-6)


def test_abi_argument_validation():
    """Error convention of the C ABI: negative return code + message, never a crash (include/knpemi_hip.h)."""
    import ctypes as C
    from cgx_hip import _lib
    p, be, o = _setup(8, "square", perturb=False)
    lib = be.lib
    # bad parameters
    z = (C.c_double * 3)(1, 1, 0)
    d = (C.c_double * 3)(1, 1, 1)
    assert lib.knp_set_params(be.ctx, 1e-5, 1.0, 1.0, 1.0, 3, z, d, d) == -1          # zero valence
    assert b"valence" in lib.knp_last_error(be.ctx)
    assert lib.knp_set_params(be.ctx, -1.0, 1.0, 1.0, 1.0, 3, d, d, d) == -1
    assert lib.knp_set_params(be.ctx, 1e-5, 1.0, 1.0, 1.0, 2, d, d, d) == -1          # n_ions != 3
    be.set_params()                                                                    # restore
    # invalid bytecode: register out of range, unknown constant
    bad = (C.c_int32 * 4)(6, 99, 0, 0)
    assert lib.knp_set_program(be.ctx, 0, 1, bad, 0, None) == -1
    bad = (C.c_int32 * 4)(0, 0, 5, 0)
    assert lib.knp_set_program(be.ctx, 0, 1, bad, 1, (C.c_double * 1)(1.0)) == -1
    be.upload_programs()
    # solve before assembly -> state error, not a crash
    p2, be2, _ = _setup(8, "square", perturb=False)
    its, rn, reason = C.c_int32(), C.c_double(), C.c_int32()
    rc = lib.knp_gmres_solve(be2.ctx, be2.b.data_ptr(), be2.x.data_ptr(), 1e-9, 1e-50, 10, 30, C.byref(its), C.byref(rn), C.byref(reason))
    assert rc == -3 and b"not assembled" in lib.knp_last_error(be2.ctx)
    # restart out of range
    be.assemble_matrix()
    rc = lib.knp_gmres_solve(be.ctx, be.b.data_ptr(), be.x.data_ptr(), 1e-9, 1e-50, 10, 500, C.byref(its), C.byref(rn), C.byref(reason))
    assert rc == -1
    # ... the slot layout carries restart + 2 values per reduction in 57 slots: 55 is the largest legal restart
    rc = lib.knp_gmres_solve(be.ctx, be.b.data_ptr(), be.x.data_ptr(), 1e-9, 1e-50, 10, 56, C.byref(its), C.byref(rn), C.byref(reason))
    assert rc == -1 and b"[1,55]" in lib.knp_last_error(be.ctx)
    # entry points added in round 3
    out1 = (C.c_double * 1)()
    k = C.c_int32()
    for _ in range(3):
        assert lib.knp_timer_mark(be.ctx, 0) == 0
    assert lib.knp_timer_pending(be.ctx) == 3
    assert lib.knp_timer_read(be.ctx, 1, out1, C.byref(k)) == -1 and b"capacity" in lib.knp_last_error(be.ctx)      # 3 marks need 2 slots
    out2 = (C.c_double * 2)()
    assert lib.knp_timer_read(be.ctx, 2, out2, C.byref(k)) == 0 and k.value == 2 and out2[0] >= 0.0 and lib.knp_timer_pending(be.ctx) == 0
    assert lib.knp_assemble_matrix_async(be.ctx, None) == -1
    assert lib.knp_get_traffic_model(be.ctx, None) == -1
    rp0 = (C.c_int32 * 2)(0, 0)
    ci0 = (C.c_int32 * 1)(0)
    v0 = (C.c_double * 1)(0.0)
    assert lib.knp_amg_set_level_coarse_fused(be.ctx, 0, 0, 1, rp0, ci0, v0, 1, rp0, ci0, v0) == -1      # level 0 is never an intermediate level
    assert lib.knp_get_precond_phi_csr(be2.ctx, rp0, ci0, v0) == -3 and b"not assembled" in lib.knp_last_error(be2.ctx)
    # Dirichlet dof out of range
    bad = (C.c_int32 * 1)(10 ** 8)
    assert lib.knp_set_dirichlet(be.ctx, 1, bad) == -1
    # AMG selected without a hierarchy
    assert lib.knp_pc_setup(be2.ctx, 2) == -3
    # non-convergence is a reason code, not an error
    be.assemble_rhs()
    rc = lib.knp_gmres_solve(be.ctx, be.b.data_ptr(), be.x.data_ptr(), 1e-30, 1e-300, 2, 30, C.byref(its), C.byref(rn), C.byref(reason))
    assert rc == 0 and reason.value == -3 and its.value == 2


def test_mesh_validation_in_knp_create():
    import ctypes as C
    from cgx_hip import _lib
    from cgx_hip._lib import MeshDesc
    lib = _lib.load()
    coords = np.array([[0, 0], [1, 0], [0, 1]], dtype=np.float64)
    cells = np.array([[0, 1, 5]], dtype=np.int32)                       # vertex index out of range
    side = np.zeros(1, dtype=np.uint8)
    q = np.array([[0.5, 0.5]]); w = np.array([1.0])
    d = MeshDesc(dim=2, n_vertices=3, n_vertices_owned=3, n_cells=1, n_cells_owned=1,
                 cells=cells.ctypes.data_as(_lib.i32p), coords=coords.ctypes.data_as(_lib.f64p),
                 cell_side=side.ctypes.data_as(_lib.u8p), n_gamma=0, gamma=None, gamma_prog=None,
                 n_q=1, q_pts=q.ctypes.data_as(_lib.f64p), q_w=w.ctypes.data_as(_lib.f64p))
    ctx = C.c_void_p()
    rc = lib.knp_create(C.byref(ctx), C.byref(d))
    assert rc == -5 and b"out of range" in lib.knp_last_error(ctx)
    lib.knp_destroy(ctx)
    cells[0, 2] = 2
    coords[2] = [2, 0]                                                  # degenerate (collinear) cell
    ctx = C.c_void_p()
    rc = lib.knp_create(C.byref(ctx), C.byref(d))
    assert rc == -5 and b"degenerate" in lib.knp_last_error(ctx)
    lib.knp_destroy(ctx)


@pytest.mark.parametrize("kind,N,pc", [("square", 16, "btcc"), ("square", 16, "hypre"), ("cube", 6, "btcc")])
def test_dirichlet_bcs_without_mms(kind, N, pc):
    """``dirichlet_bcs: True`` outside MMS runs (KNPEMIx_problem.py:135-160): all fields pinned on the exterior boundary,
    no null space.  The reference fills the BC functions at construction time, i.e. with the class-default initial
    values; the config below uses those defaults as initial conditions too, so the data are consistent.

    Checked against the oracle running the SAME algorithm (GMRES(30), preconditioned-norm stopping rule, the same
    hierarchies): with every extracellular field pinned, "charge the membrane from the intracellular bulk" is a mode
    on which the block-diagonal P (the reference's preconditioner form) is ~1e7 times weaker than A^-1, so that
    stopping rule accepts iterates that are still 1e-4 away from the LU solution -- with hypre as with this AMG."""
    import knpemi_oracle as K
    cfg = ci_config(N=N, steps=2, rtol=1e-11, kind=kind, pc=pc)
    cfg["dirichlet_bcs"] = True
    cfg["initial_conditions"].update({"Na_i": 10, "Na_e": 145, "K_i": 130, "K_e": 3, "Cl_i": 5, "Cl_e": 134})
    cfg["solver"]["ksp_settings"]["amg_fp32"] = False
    s = run_native(cfg)
    assert all(r > 0 for r in s.reasons)
    params = K.Params(ki_init=K.OracleKNPEMI.REF_DEFAULT_KI, ke_init=K.OracleKNPEMI.REF_DEFAULT_KE)
    o = (K.make_square if kind == "square" else K.make_cube)(N, models=K.CI_MODELS(), params=params)
    x = o.coords / o.coords.max()
    bv = np.nonzero(np.any((np.abs(x) < 1e-12) | (np.abs(x - 1.0) < 1e-12), axis=1))[0]
    dofs, vals = o.dirichlet_initial_values(bv)
    assert len(dofs) == 4 * len(bv)                      # the intracellular box does not touch the boundary
    assert set(dofs.tolist()) == set(s.backend.bc_dofs.cpu().numpy().tolist())

    def fac(P):                       # the hierarchies are data: the ones the host setup built for the library
        if pc == "btcc":
            hk, hp = s.hierarchies
            return K.pc_btcc(o, hk, hp, s.amg_pre, s.amg_post, s.amg_cheby_degree, bc_dofs=dofs, fused=bool(s.backend.stats()["fused"]))
        h = s.hierarchy
        return K.pc_amg_vcycle(h.levels, h.coarse_inv, s.amg_pre, s.amg_post, s.amg_cheby_degree, fused=bool(s.backend.stats()["fused"]))
    xo = o.run_dirichlet(2, dofs, vals, solver="gmres", pc=fac, rtol=1e-11)
    assert o.dirichlet_iterations == list(s.iterations)
    xn = s.backend.x.cpu().numpy()
    for f in range(4):
        assert np.max(np.abs(xn[f::4] - xo[f::4])) <= 1e-9 * np.max(np.abs(xo[f::4])), f
    # boundary values really are the pinned ones
    p = s.problem
    assert np.allclose(p.wh[1][0].numpy()[bv], 145.0) and np.allclose(p.wh[1][3].numpy()[bv], 0.0)


@pytest.mark.parametrize("kind,N", [("square", 16), ("cube", 6)])
def test_runtime_compiled_membrane_programs_match_interpreter(kind, N, monkeypatch):
    """The membrane currents run as native code compiled with hiprtc from the uploaded bytecode (csrc/knp_jit.cpp);
    the bytecode interpreter is the fallback.  Same right-hand side from both, and the native one is the default."""
    def rhs(jit):
        monkeypatch.setenv("KNP_JIT", "1" if jit else "0")
        p = make_problem(ci_config(N=N, steps=1, kind=kind))
        be = p.create_backend()
        p.t.value = float(p.dt.value)
        for m in p.ionic_models:
            if hasattr(m, "update_t_mod"):
                m.update_t_mod()
        be.assemble_matrix()
        be.assemble_rhs()
        status = be.lib.knp_jit_status(be.ctx).decode()
        return be.b.cpu().numpy().copy(), status
    b_native, st_native = rhs(True)
    b_interp, st_interp = rhs(False)
    assert st_native == "native", st_native
    assert "KNP_JIT=0" in st_interp
    for f in range(4):
        scale = np.max(np.abs(b_interp[f::4]))
        assert np.max(np.abs(b_native[f::4] - b_interp[f::4])) <= 1e-13 * scale, f


def test_pin_ecs_potential(monkeypatch):
    """``pin_ecs_potential`` (class switch of the reference, KNPEMIx_problem.py:163-196, :997): phi_e = 0 at one
    non-membrane vertex instead of the null-space gauge.  Gauge-invariant quantities equal the oracle's."""
    from CGx.KNPEMI.KNPEMIx_problem import ProblemKNPEMI
    monkeypatch.setattr(ProblemKNPEMI, "pin_ecs_potential", True)
    s = run_native(ci_config(N=16, steps=2, rtol=1e-13, pc="btcc"))
    assert all(r > 0 for r in s.reasons)
    p = s.problem
    assert len(p.bc_vertices) == 1 and abs(p.wh[1][3].numpy()[p.bc_vertices[0]]) <= 1e-14
    o = run_oracle(N=16, steps=2)
    gam = (o.lay.node_i >= 0) & (o.lay.node_e >= 0)
    assert np.allclose(p.phi_m_prev.numpy()[gam], o.phi_m[gam], rtol=1e-6)
    ve = o.lay.node_e >= 0
    shift = (p.wh[1][3].numpy()[ve] - o.phi[1][ve])
    assert np.ptp(shift) <= 1e-6 * np.abs(o.phi[0]).max()          # potentials differ by one constant only
    for j in range(3):
        assert np.allclose(p.wh[1][j].numpy()[ve], o.k[1][j][ve], rtol=1e-7)


def _run_in_subprocess(env_extra):
    """three CI steps in a fresh interpreter (the library reads its environment switches once per process)"""
    import json
    import os
    import subprocess
    import sys
    code = ("import sys; sys.path[:0]=['tests','oracle','knp-emi-cgx_amd']; import conftest, json, numpy as np\n"
            "from parity_utils import ci_config, run_native\n"
            "s = run_native(ci_config(N=24, steps=3, rtol=1e-10, pc='btcc', kind='square'))\n"
            "x = s.backend.x.cpu().numpy()\n"
            "print('RESULT' + json.dumps({'its': list(s.iterations), 'sum': float(x.sum()), 'l1': float(np.abs(x).sum())}))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=300)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")]
    assert line, out.stdout[-2000:] + out.stderr[-2000:]
    return json.loads(line[0][6:])


def test_side_stream_prenorm_changes_nothing():
    """knp_gmres_prepare computes ||B b|| on a side stream while the matrix is assembled: same iteration counts and the
    same solution, bit for bit, as the in-line computation (KNP_NO_PREPARE=1)."""
    a = _run_in_subprocess({})
    b = _run_in_subprocess({"KNP_NO_PREPARE": "1"})
    assert a["its"] == b["its"]
    assert a["sum"] == b["sum"] and a["l1"] == b["l1"]


@pytest.mark.parametrize("fp32", [False, True])
@pytest.mark.parametrize("kind,N,pc,flags", [("square", 32, "hypre", 1), ("cube", 8, "btcc", 3), ("square", 24, "btcc", 3)])
def test_fused_cycle_is_the_same_operator_as_the_unfused_cycle(kind, N, pc, flags, fp32, monkeypatch):
    """The fused V(1,1) cycle (pre-smoothing + residual as one gather with P Dinv; prolongation + post-smoothing as one
    gather with S = (I - c2 Dinv A) P; potential hierarchy on node-indexed vectors) applies the same linear operator as
    the level-by-level cycle: identical up to rounding with fp64 storage, up to fp32 rounding of S vs (A, P) otherwise."""
    from CGx.KNPEMI.KNPEMIx_solver import SolverKNPEMI
    cfg = ci_config(N=N, steps=1, rtol=1e-9, kind=kind, pc=pc)
    cfg["solver"]["ksp_settings"]["amg_coarse_size"] = 150
    cfg["solver"]["ksp_settings"]["amg_fp32"] = fp32
    p = make_problem(cfg)
    s = SolverKNPEMI(p, solver_config=p.solver_config)
    s.setup_solver()
    be = s.backend
    p.setup_preconditioner(s.use_block_Jacobi)
    s.assemble_preconditioner()
    be.assemble_rhs()
    be.assemble_matrix()
    be.pc_setup(s._pc_kind)
    assert be.stats()["fused"] == flags
    rng = np.random.default_rng(3)
    r = torch.as_tensor(rng.standard_normal(be.n_dof_owned), device=be.device)
    z1 = torch.zeros_like(r)
    be.pc_apply(r, z1)
    monkeypatch.setenv("KNP_FUSED", "0")
    be.pc_setup(s._pc_kind)
    assert be.stats()["fused"] == 0
    z0 = torch.zeros_like(r)
    be.pc_apply(r, z0)
    z0, z1 = z0.cpu().numpy(), z1.cpu().numpy()
    tol = 2e-6 if fp32 else 1e-12
    for f in range(4):
        assert np.max(np.abs(z1[f::4] - z0[f::4])) <= tol * np.max(np.abs(z0[f::4])), f


@pytest.mark.parametrize("kind,N,pc,nf", [("cube", 10, "btcc", 3), ("square", 40, "btcc", 3), ("square", 48, "hypre", 4), ("cube", 8, "hypre", 4)])
def test_node_blocked_cycle_is_the_same_operator_as_the_scalar_cycle(kind, N, pc, nf, monkeypatch):
    """Hierarchies built with node-synchronised aggregation (cgx_hip/amg.py ``node_fields``: the ion hierarchy of the
    block-triangular form, all four fields in the ``hypre`` form): the library keeps node-blocked copies of R, the coarse
    operators and S (one column node per entry, nf values behind it; union pattern with zeros where the potential alone
    couples the two sides of a membrane vertex) and the fused cycle runs on them.  Same linear operator as the scalar-row
    kernels on the same fp32-stored hierarchy, to summation order."""
    from CGx.KNPEMI.KNPEMIx_solver import SolverKNPEMI
    cfg = ci_config(N=N, steps=1, rtol=1e-9, kind=kind, pc=pc)
    cfg["solver"]["ksp_settings"]["amg_coarse_size"] = 100
    p = make_problem(cfg)
    s = SolverKNPEMI(p, solver_config=p.solver_config)
    s.setup_solver()
    be = s.backend
    p.setup_preconditioner(s.use_block_Jacobi)
    s.assemble_preconditioner()
    assert s.hierarchies[0].node_fields == nf and len(s.hierarchies[0].levels) >= 3
    be.assemble_rhs()
    be.assemble_matrix()
    be.pc_setup(s._pc_kind)
    st = be.stats()
    fused = 3 if pc == "btcc" else 1
    assert st["fused"] == fused and st["blocked"] == 1
    rng = np.random.default_rng(5)
    r = torch.as_tensor(rng.standard_normal(be.n_dof_owned), device=be.device)
    z1 = torch.zeros_like(r)
    be.pc_apply(r, z1)
    monkeypatch.setenv("KNP_BLOCKED", "0")
    be.pc_setup(s._pc_kind)
    st = be.stats()
    assert st["fused"] == fused and st["blocked"] == 0
    z0 = torch.zeros_like(r)
    be.pc_apply(r, z0)
    z0, z1 = z0.cpu().numpy(), z1.cpu().numpy()
    for f in range(4):
        assert np.max(np.abs(z1[f::4] - z0[f::4])) <= 1e-12 * np.max(np.abs(z0[f::4])), f


@pytest.mark.parametrize("N,kind", [(24, "square"), (8, "cube")])
def test_non_block_jacobi_form_of_P(N, kind, monkeypatch):
    """``use_block_Jacobi = False`` (class switch of the reference's solver, KNPEMIx_solver.py:37; form KNPEMIx_problem.py:720-722):
    P keeps the (phi,k) blocks of A and is applied as a block forward substitution.  Same iterations and iterates as the
    oracle's restatement, and the converged answer is the sparse-LU one."""
    import knpemi_oracle as K
    from CGx.KNPEMI.KNPEMIx_solver import SolverKNPEMI
    monkeypatch.setattr(SolverKNPEMI, "use_block_Jacobi", False)
    cfg = ci_config(N=N, steps=2, rtol=1e-10, kind=kind)
    cfg["solver"]["ksp_settings"]["amg_coarse_size"] = 150
    cfg["solver"]["ksp_settings"]["ksp_max_it"] = 500
    s = run_native(cfg)
    assert all(r > 0 for r in s.reasons)
    from cgx_hip import _lib
    assert s._pc_kind == _lib.PC_AMG_LT and not s.problem.P_block_jacobi
    hk, hp = s.hierarchies
    fused = bool(s.backend.stats()["fused"])
    hk, hp = fp32_stored(hk, coarse=fused), fp32_stored(hp, level0_uploaded=getattr(s, "_coupled_phi", False))
    o = make_oracle(N, kind)
    xo, its = o.run(2, solver="gmres", rtol=1e-10, pc=lambda P: K.pc_block_lower(o, hk, hp, s.amg_pre, s.amg_post, s.amg_cheby_degree, fused=fused))
    assert its == list(s.iterations)
    x = s.backend.x.cpu().numpy()
    for f in range(4):
        assert np.max(np.abs(x[f::4] - xo[f::4])) <= 1e-8 * np.max(np.abs(xo[f::4])), f
    ol = run_oracle(N=N, steps=2, kind=kind)
    ni, _ = s.potential_norms()
    oi, _ = ol.potential_norms()
    assert abs(ni - oi) <= 1e-5 * oi


def test_direct_solver_config_against_reference_pins():
    """The reference's direct-solver CI test (configs/tests/electric_potential_norms_direct_solver.yaml: ``direct: True``, 10 steps):
    natively GMRES+AMG at rtol 1e-13 with PETSc's gauge for a preonly/LU solve with an attached null space (solution projected:
    zero-mean potentials).  Both saved norms (tests/KNPEMI/electric_potential_norms_direct_solver.py:55-56) within 1e-6."""
    pin_i, pin_e = 2.6337161145147203e-08, 1.5258564901943312e-08
    s = run_native(ci_config(N=32, steps=10, direct=True))
    assert all(r > 0 for r in s.reasons)
    ni, ne = s.potential_norms()
    assert abs(ni - pin_i) <= 1e-6 * pin_i, (ni, pin_i)
    assert abs(ne - pin_e) <= 1e-6 * pin_e, (ne, pin_e)
    x = s.backend.x.cpu().numpy()
    assert abs(x[3::4].sum()) <= 1e-12 * np.abs(x[3::4]).sum()


@pytest.mark.parametrize("pc", ["hypre", "btcc"])
def test_reassemble_P_rebuilds_the_preconditioner_every_step(pc):
    """``ksp_settings: {reassemble_P: True}`` (reference KNPEMIx_solver.py:34-35, 405-406: re-assemble P with the current
    concentrations every ``reassemble_N`` steps): the preconditioner matrix and its hierarchies are rebuilt inside the time loop,
    the converged solution is that of the direct solve, and P really follows the fields."""
    from CGx.KNPEMI.KNPEMIx_solver import SolverKNPEMI
    from parity_utils import run_oracle
    cfg = ci_config(N=16, steps=4, rtol=1e-11, pc=pc)
    cfg["solver"]["ksp_settings"]["reassemble_P"] = True
    p = make_problem(cfg)
    s = SolverKNPEMI(p, solver_config=p.solver_config)
    assert s.reassemble_P
    seen = []
    orig = s.assemble_preconditioner

    def spy():
        orig()
        seen.append((s.hierarchy, s.backend.precond_csr().data.copy()))    # the object itself: an id may be reused once it is collected
    s.assemble_preconditioner = spy
    s.solve()
    assert len(seen) == 4 and all(r > 0 for r in s.reasons)          # once before the loop + steps 2, 3, 4
    assert len({id(h) for h, _ in seen}) == 4                         # a new hierarchy every time
    assert np.abs(seen[-1][1] - seen[0][1]).max() > 0                 # P moved with the concentrations
    o = run_oracle(N=16, steps=4)
    gam = (o.lay.node_i >= 0) & (o.lay.node_e >= 0)
    assert np.allclose(p.phi_m_prev.numpy()[gam], o.phi_m[gam], rtol=1e-6, atol=0)
    oi, oe = o.potential_norms()
    ni, ne = s.potential_norms()
    assert abs(ni - oi) <= 1e-6 * oi and abs(ne - oe) <= 1e-5 * oe


@pytest.mark.gpu
@pytest.mark.parametrize("kind,N,pc", [("cube", 8, "btcc"), ("square", 24, "hypre")])
def test_morton_vertex_order_gives_the_same_solution(kind, N, pc):
    """``vertex_order: morton`` (config key; cgx_hip/parallel.py reorder_local_mesh): vertices, cells and membrane facets renumbered along
    a space-filling curve.  Same mesh, same physics: the fields agree vertex by vertex through ``l2g`` with the run in the native order,
    and so do the potential norms (the hierarchies differ -- aggregation follows the numbering -- so the agreement is that of two
    converged solves, not bit for bit)."""
    from parity_utils import run_native
    cfg = ci_config(N=N, steps=2, rtol=1e-12, kind=kind, pc=pc)
    s0 = run_native(cfg)
    cfg1 = ci_config(N=N, steps=2, rtol=1e-12, kind=kind, pc=pc)
    cfg1["vertex_order"] = "morton"
    s1 = run_native(cfg1)
    p0, p1 = s0.problem, s1.problem
    assert "Morton" in p1.mesh_description and not np.array_equal(p0.local_mesh.l2g, p1.local_mesh.l2g)
    assert all(r > 0 for r in s1.reasons)
    o0, o1 = np.argsort(p0.local_mesh.l2g), np.argsort(p1.local_mesh.l2g)
    assert np.allclose(p0.local_mesh.coords[o0], p1.local_mesh.coords[o1])
    pot_scale = np.abs(p0.wh[0][3].numpy()).max()      # potentials are compared on the potential scale (phi_e is 100-2000x smaller than phi_i)
    for side in (0, 1):
        for f in range(4):
            a, b = p0.wh[side][f].numpy()[o0], p1.wh[side][f].numpy()[o1]
            tol = 1e-6 * pot_scale if f == 3 else 1e-7 * max(np.abs(a).max(), 1e-300)
            assert np.abs(a - b).max() <= tol, (side, f, np.abs(a - b).max(), tol)
    pm0, pm1 = p0.phi_m_prev.numpy()[o0], p1.phi_m_prev.numpy()[o1]
    assert np.abs(pm0 - pm1).max() <= 1e-6 * np.abs(pm0).max()
    n0, n1 = s0.potential_norms(), s1.potential_norms()
    assert abs(n0[0] - n1[0]) <= 1e-6 * n0[0] and abs(n0[1] - n1[1]) <= 1e-6 * n0[0]

