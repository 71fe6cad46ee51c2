"""``SolverKNPEMI``: the reference's time-loop / linear-solver surface on top of libknpemi_hip.

Mirrors reference src/CGx/KNPEMI/KNPEMIx_solver.py: same constructor, same class-level defaults
(:25-51), same per-step order (:365-468): advance t -> HH gating update -> assemble A, b ->
(step 1) null space -> solve -> unpack -> phi_m_prev = phi_i - phi_e; same timing lists
(``assembly_time``, ``solve_time``, ``iterations``) and ``print_info`` fields (:504-548).

Linear solver mapping (reference :211-214, 269-280):
  ksp_type gmres           -> knp_gmres_solve: GMRES(30), left preconditioning, classical
                              Gram-Schmidt, preconditioned norm, non-zero initial guess
  pc_type  hypre           -> smoothed-aggregation AMG V-cycle on the block-diagonal P (cgx_hip/amg.py
                              + HIP V-cycle), the native stand-in for BoomerAMG
  pc_type  btcc            -> block lower-triangular variant: AMG on the ion blocks of P, then AMG on the
                              potential block applied to r_phi - A_{phi,k} z_k plus a Cahouet-Chabard Schur
                              term; keeps iteration counts mesh independent in 3D where the block-Jacobi
                              form does not ("Preconditioners")
  pc_type  bjacobi|vbjacobi-> per-vertex 4x4 / 8x8 block Jacobi of A (HIP)
  pc_type  none            -> unpreconditioned
  direct: True             -> MUMPS has no native counterpart; emulated by GMRES+AMG driven to
                              rtol 1e-13, with the gauge of the reference's preonly/LU solve: zero mean over
                              the potential unknowns (the solution is projected, as PETSc does with the
                              attached null space; see tests/test_oracle_pins.py).
Unlike the reference (which never checks ``ksp.getConvergedReason()``), non-convergence is recorded
in ``self.reasons`` and raises when ``strict`` is set.
"""
from __future__ import annotations

import time

import numpy as np
import torch

import os

from . import _lib, amg, dist_amg
from .ionic_models import HodgkinHuxley
from .problem import ProblemKNPEMI


class _KSPInfo:
    """What callers read off ``solver.ksp`` in the reference."""

    def __init__(self):
        self.its = 0
        self.reason = 0
        self.rnorm = 0.0

    def getIterationNumber(self):
        return self.its

    def getConvergedReason(self):
        return self.reason

    def getResidualNorm(self):
        return self.rnorm


class SolverKNPEMI:
    ksp_type = "gmres"
    pc_type = "hypre"
    ksp_rtol = 1e-8
    ksp_max_it = 5000
    use_P_mat = True
    reassemble_P = False
    reassemble_N = 1
    verbose = False
    use_block_Jacobi = True
    nonzero_init_guess = True
    norm_type = "preconditioned"
    max_amg_iter = 1
    strong_threshold = 0.5
    save_interval = 20
    tot_its = 0.0
    tot_assembly_time = 0.0
    tot_solver_time = 0.0
    gmres_restart = 30
    strict = False
    # native AMG parameters
    amg_theta = 0.08
    amg_cheby_degree = 1
    amg_pre = 1
    amg_post = 1
    amg_coarse_size = 2500
    amg_replicate_below = 300000
    amg_fp32 = True        # mixed-precision preconditioner storage (operators fp32, vectors/Krylov fp64)
    amg_node_sync = True   # ion hierarchy: aggregate NODES once, all three ion fields share aggregates and sparsity patterns
    amg_split_decoupled = True   # unknowns without off-diagonal entries on a level are solved by its smoother, not coarsened further
    # aggregation distance per level (last entry repeated: "2" | "2,1" | "1") of the first hierarchy (ions / all fields) and of the
    # potential hierarchy of btcc; "auto": see ion_agg_distance
    amg_agg_distance = "auto"
    amg_agg_distance_phi = "2"
    _b_is_final = False
    btcc_coupled_phi = True  # btcc on one GPU: potential hierarchy on the potential block of A (both sides + membrane coupling), not on P's
    amg_setup = "gpu"      # where the hierarchy is built: "gpu" (torch sparse products, cgx_hip/amg_gpu.py) | "host" (SciPy)

    def __init__(self, problem: ProblemKNPEMI, solver_config: dict):
        self.problem = problem
        self.comm = problem.comm
        self.time_steps = problem.time_steps
        out = solver_config.get("output", {}) if isinstance(solver_config.get("output", {}), dict) else {}
        self.save_xdmfs = out.get("save_xdmf", False)
        self.save_pngs = out.get("save_pngs", False)
        self.save_cpoints = out.get("save_cpoints", False)
        self.save_dat = out.get("save_dat", False)
        self.save_mat = out.get("save_mat", False)
        if "save_interval" in out:
            self.save_interval = out["save_interval"]
        self.out_file_prefix = problem.output_dir
        self.direct_solver = bool(solver_config["direct"])
        self.view_input = solver_config.get("view_ksp", False)
        if "ksp_settings" in solver_config:
            ks = solver_config["ksp_settings"]
            if "ksp_type" in ks: self.ksp_type = ks["ksp_type"]
            if "pc_type" in ks: self.pc_type = ks["pc_type"]
            if "ksp_rtol" in ks: self.ksp_rtol = float(ks["ksp_rtol"])
            if "norm_type" in ks: self.norm_type = ks["norm_type"]
            if "strong_threshold" in ks: self.strong_threshold = float(ks["strong_threshold"])
            if "reassemble_P" in ks: self.reassemble_P = bool(ks["reassemble_P"])
            if "non_zero_init_guess" in ks: self.nonzero_init_guess = bool(ks["non_zero_init_guess"])
            if "ksp_max_it" in ks: self.ksp_max_it = int(ks["ksp_max_it"])
            if "gmres_restart" in ks: self.gmres_restart = int(ks["gmres_restart"])
            if "strict" in ks: self.strict = bool(ks["strict"])
            for k in ("amg_theta", "amg_cheby_degree", "amg_pre", "amg_post", "amg_coarse_size", "amg_replicate_below", "amg_fp32", "amg_setup", "amg_node_sync", "amg_split_decoupled",
                      "btcc_coupled_phi", "amg_agg_distance", "amg_agg_distance_phi"):
                if k in ks: setattr(self, k, type(getattr(self, k))(ks[k]))
        if self.ksp_type != "gmres":
            raise NotImplementedError(f"ksp_type '{self.ksp_type}': only 'gmres' is implemented natively.")
        if self.pc_type not in ("hypre", "amg", "btcc", "bjacobi", "vbjacobi", "none"):
            raise NotImplementedError(f"pc_type '{self.pc_type}' has no native counterpart (hypre|amg|btcc|bjacobi|vbjacobi|none).")
        if self.norm_type != "preconditioned":
            raise NotImplementedError("only norm_type 'preconditioned' is implemented (reference default).")
        if self.save_mat:
            self.time_steps = 1
        self.ksp = _KSPInfo()
        self.setup_time = 0.0
        self.reasons = []
        self.ode_time = []

    def print(self, *a, **k):
        self.problem.print(*a, **k)

    # ---- reference :104-116
    def assemble(self):
        self.print("Assembling linear system ...")
        be = self.backend
        p = self.problem
        # A and b are independent (the reference assembles A then b, :114-116): the matrix is assembled on the library's own stream
        # while the right-hand side chain runs on this one; the preconditioned norm of b (the first thing the solve needs) then
        # starts on a third stream and overlaps both, the solve's first SpMV and its first preconditioner application
        async_matrix = self._b_is_final and not p.MMS_test
        if async_matrix:
            be.assemble_matrix_async()
        be.assemble_rhs()
        if p.MMS_test:
            # extra terms of L for the manufactured solution (KNPEMIx_problem.py:616-651): host integrals of the
            # analytic sources, added to the device vector; then the Dirichlet values (bcs= of assemble_vector_block)
            if not hasattr(p, "_mms_asm"):
                from .mms import MMSAssembler
                p._mms_asm = MMSAssembler(p)
            extra = p._mms_asm.rhs_vector(be.node_i, be.node_e, be.n_dof_local)
            be.b += torch.as_tensor(extra, dtype=torch.float64, device=be.device)
        be.apply_dirichlet_rhs()
        if self._b_is_final:
            be.gmres_prepare()
        if not async_matrix:
            be.assemble_matrix()

    # ---- reference :118-135
    def assemble_preconditioner(self):
        self.print("Assembling preconditioner ...")
        be = self.backend
        # btcc (native extension) on one GPU: the potential hierarchy is built on the potential block of A as assembled now -- both
        # sides and their membrane coupling -- instead of P's uncoupled block with the reference's minus sign (24 -> 17
        # iterations per step on the membrane-dominated lattices, never more on the cubes)
        self._coupled_phi = bool(self.btcc_coupled_phi and self._pc_kind == _lib.PC_AMG_BT and self.comm.size == 1 and not self.problem.MMS_test
                                 and not self.problem.dirichlet_bcs and not self.problem.pin_ecs_potential and getattr(self.problem, "P_block_jacobi", True))
        be.set_coupled_potential(self._coupled_phi)
        be.assemble_precond()
        if self._pc_kind in (_lib.PC_AMG, _lib.PC_AMG_BT) and not getattr(self.problem, "P_block_jacobi", True):
            # the reference's non-block-Jacobi form of P: same diagonal blocks + the (phi,k) coupling, applied as a block forward
            # substitution with the ion and potential hierarchies
            self._pc_kind = _lib.PC_AMG_LT
        if self._pc_kind in (_lib.PC_AMG, _lib.PC_AMG_BT, _lib.PC_AMG_LT):
            tic = time.perf_counter()
            phases = self.amg_setup_phases = {}
            last = [tic]

            def lap(name):      # where the one-off setup goes (bench.py: setup_s.amg_phases)
                now = time.perf_counter()
                phases[name] = phases.get(name, 0.0) + now - last[0]
                last[0] = now
            be.check(be.lib.knp_amg_set_precision(be.ctx, 1 if self.amg_fp32 else 0))
            up_lib = be.lib
            if os.environ.get("KNP_AMG_TIMING"):      # developer aid: time per entry point of the hierarchy hand-over
                class _Timed:
                    def __init__(self, lib):
                        self.lib, self.t = lib, {}

                    def __getattr__(self, name):
                        f = getattr(self.lib, name)

                        def w(*a):
                            t0 = time.perf_counter()
                            r = f(*a)
                            self.t[name] = self.t.get(name, 0.0) + time.perf_counter() - t0
                            return r
                        return w
                up_lib = _Timed(be.lib)
            P = be.precond_csr()
            lap("fetch_P_csr")
            if self.comm.size > 1 and os.environ.get("KNP_DIST_PC", "global") != "bj":
                self._assemble_distributed_amg(P)
                self.P_ = "device CSR (see Backend.precond_csr)"
                return
            own_block = lambda M: M if M.shape[1] == be.n_dof_owned else M[:, :be.n_dof_owned].tocsr()
            P = own_block(P)                           # per-rank block (block-Jacobi across GPUs)
            # aggregation distance: the first hierarchy (the one built with node fields) may use smaller aggregates on its finest levels
            dist_of = lambda nf: self.ion_agg_distance() if nf is not None else self.phi_agg_distance()
            # ``fields``: the hierarchy of a field class of P (ions (0, 1, 2), potential (3,)); the device builder restricts P itself
            host_build = lambda M, nf=None, fields=None: amg.build_hierarchy(M if fields is None else amg.restrict_to_fields(M, fields), theta=self.amg_theta,
                                                                             coarse_size=self.amg_coarse_size, node_fields=nf,
                                                                             split_decoupled=self.amg_split_decoupled, smoother_degree=self.amg_cheby_degree,
                                                                             agg_distance=dist_of(nf))
            if str(self.amg_setup) == "gpu":
                from . import amg_gpu

                def build(M, nf=None, fields=None):
                    # the setup is host logic either way (the V-cycle always runs in the library): if torch's sparse
                    # products are not usable on this installation, build the same hierarchy with SciPy
                    try:
                        return amg_gpu.build_hierarchy(M, theta=self.amg_theta, coarse_size=self.amg_coarse_size, device=be.device, node_fields=nf,
                                                       split_decoupled=self.amg_split_decoupled, smoother_degree=self.amg_cheby_degree,
                                                       agg_distance=dist_of(nf), fields=fields)
                    except (RuntimeError, NotImplementedError) as exc:
                        self.print(f"device-side AMG setup unavailable ({type(exc).__name__}: {exc}); using the host setup")
                        return host_build(M, nf, fields)
            else:
                build = host_build
            if self._pc_kind == _lib.PC_AMG:
                self.hierarchy = build(P, self.all_node_fields())
                lap("build_hierarchy")
                amg.upload(up_lib, be.ctx, be.check, self.hierarchy, self.amg_pre, self.amg_post, self.amg_cheby_degree, index=0, level0_native=True)
                be.check(be.lib.knp_amg_use_native_level0(be.ctx, 0, 1))   # level 0 is the library's own P
                lap("upload")
                self.hierarchies = [self.hierarchy]
            else:
                hk = build(P, self.ion_node_fields(), (0, 1, 2))
                lap("build_ion_hierarchy")
                if self._coupled_phi:
                    Pphi = own_block(be.precond_phi_csr())
                    lap("potential_block")
                    hp = build(Pphi)
                else:
                    hp = build(P, None, (3,))
                lap("build_potential_hierarchy")
                amg.upload(up_lib, be.ctx, be.check, hk, self.amg_pre, self.amg_post, self.amg_cheby_degree, index=0, level0_native=True)
                amg.upload(up_lib, be.ctx, be.check, hp, self.amg_pre, self.amg_post, self.amg_cheby_degree, index=1, level0_native=not self._coupled_phi)
                be.check(be.lib.knp_amg_use_native_level0(be.ctx, 0, 2))   # ion fields of P
                be.check(be.lib.knp_amg_use_native_level0(be.ctx, 1, 4 if self._coupled_phi else 3))   # potential: uploaded coupled block | P's
                lap("upload")
                self.hierarchies = [hk, hp]
                self.hierarchy = hk
            if up_lib is not be.lib:
                import sys
                print("[amg upload] inside the library:", {k: round(v, 3) for k, v in up_lib.t.items()}, file=sys.stderr, flush=True)
            self.amg_setup_time = time.perf_counter() - tic
            self.print(f"AMG hierarchies: {[h.describe() for h in self.hierarchies]} (host setup {self.amg_setup_time:0.3f} s)")
        self.P_ = "device CSR (see Backend.precond_csr)"

    def ion_agg_distance(self):
        """Aggregation distance per level of the first hierarchy (``amg.build_hierarchy(agg_distance=...)``, last entry repeated).
        ``auto``: distance 2 (a root and everything within two strong couplings) on the finest level, and in 3D distance 1 on every
        level below it.  The Galerkin operators of smoothed aggregation on tetrahedral meshes have 40-100 entries per row from level 1 on:
        distance-2 aggregates there swallow hundreds of fine nodes and the coarse correction degrades, while the level-0 aggregates keep
        level 1 small (7-9 % of level 0), so the finer coarsening below costs little.  Measured on MI355X, btcc, ms per step (GMRES
        iterations): cube 136^3 21.1 (9.5) -> 17.0 (6.7); cube 64^3 1.40 (4.75) -> 1.34 (4.26); tissue surrogate 97^3 with 13 824 cells
        32.5 (19.9) -> 23.2 (13.8).  Distance 1 on all levels: 28.5 (12.6) on the tissue surrogate, 22.7 (5.8) / 3.07 (5.4) on the cubes (level 1
        four times larger); "1,2": no gain.  In 2D (7-point graphs, coarse rows of 12-16 entries) distance 2 everywhere stays: 512^2 0.555 vs
        0.561 ms with "2,1"."""
        v = str(self.amg_agg_distance).replace(" ", "")
        if v == "auto":
            v = "2,1" if int(self.problem.mesh.geometry.dim) == 3 else "2"
        return [int(t) for t in v.split(",") if t]

    def phi_agg_distance(self):
        return [int(t) for t in str(self.amg_agg_distance_phi).replace(" ", "").split(",") if t]

    def ion_node_fields(self):
        """``node_fields`` of the ion hierarchy (amg.build_hierarchy): the three ion blocks of P have the same graph"""
        return (4, (0, 1, 2)) if self.amg_node_sync else None

    def all_node_fields(self):
        """``node_fields`` of the all-field hierarchy (``hypre`` form): the potential follows the aggregates of the ion graph"""
        return (4, (0, 1, 2, 3)) if self.amg_node_sync else None

    def _assemble_distributed_amg(self, P_loc):
        """Multi-GPU: one global smoothed-aggregation hierarchy (cgx_hip/dist_amg.py) instead of per-GPU blocks."""
        be = self.backend
        tic = time.perf_counter()
        halo0, start, ggid, gown = be.dof_level_halo()
        specs = [((0, 1, 2, 3), 1)] if self._pc_kind == _lib.PC_AMG else [((0, 1, 2), 2), ((3,), 3)]      # BT and LT: two hierarchies
        self.hierarchies = []
        for index, (fields, native_mode) in enumerate(specs):
            Pm = P_loc if len(fields) == 4 else dist_amg.restrict_to_fields_rect(P_loc, fields)
            levels, tail = dist_amg.build_distributed_hierarchy(self.comm, Pm, halo0, start, ggid, gown, theta=self.amg_theta,
                                                                coarse_size=self.amg_coarse_size,
                                                                replicate_below=self.amg_replicate_below, device=be.device,
                                                                agg_distance=self.phi_agg_distance() if fields == (3,) else self.ion_agg_distance())
            dist_amg.upload(be.lib, be.ctx, be.check, levels, tail, self.amg_pre, self.amg_post, self.amg_cheby_degree, index=index)
            be.check(be.lib.knp_amg_use_native_level0(be.ctx, index, native_mode))
            for l, L in enumerate(levels):
                be.level_halos[(index, l)] = L.halo
                if L.replicated:
                    be.level_repl[(index, l)] = int(L.repl_n)
                be.p2p_attach_level(index, l, L.halo, int(L.repl_n) if L.replicated else 0)
            self.hierarchies.append((levels, tail))
            self.print(f"distributed AMG hierarchy {index}: {dist_amg.describe(levels, tail, self.comm)}")
        self.hierarchy = self.hierarchies[0]
        be.check(be.lib.knp_set_deflation(be.ctx, 0, None, None))     # the global coarse levels carry those modes
        self.amg_setup_time = time.perf_counter() - tic

    def reassemble_preconditioner(self):
        self.print("Re-assembling preconditioner ...")
        self.assemble_preconditioner()
        self.backend.pc_setup(self._pc_kind)

    # ---- reference :152-295
    def setup_solver(self):
        p = self.problem
        self.backend = p.create_backend()
        be = self.backend
        self.A = be                                   # matrix handle: be.csr(), be.spmv()
        self.b = be.b
        self.x = be.x
        if self.direct_solver:
            self.print("Direct solver requested: emulated natively by GMRES+AMG at rtol 1e-13, zero-mean gauge (no MUMPS on the GPU).")
            self._pc_kind = _lib.PC_AMG
            self._rtol = 1e-13
        else:
            self.print("Setting up iterative solver ...")
            self._pc_kind = {"hypre": _lib.PC_AMG, "amg": _lib.PC_AMG, "btcc": _lib.PC_AMG_BT, "bjacobi": _lib.PC_VBJACOBI,
                             "vbjacobi": _lib.PC_VBJACOBI, "none": _lib.PC_NONE}[self.pc_type]
            self._rtol = self.ksp_rtol
        # initial conditions as initial guess (reference :177-209).  In MMS runs the initial data are fields
        # (already interpolated by set_initial_conditions); the reference would need its direct solver there.
        for idx, ion in enumerate(p.ion_list if not p.MMS_test else []):
            if not p.glia_flag:
                p.wh[0][idx].x.array[:] = ion["ki_init"].value
                p.wh[1][idx].x.array[:] = ion["ke_init"].value
            else:
                p.wh[0][idx].x.array[p.neuron_dofs] = ion["ki_init_n"].value
                p.wh[0][idx].x.array[p.glia_dofs] = ion["ki_init_g"].value
                p.wh[1][idx].x.array[:] = ion["ke_init"].value
        if p.MMS_test:
            pass
        elif not p.glia_flag:
            p.wh[0][p.N_ions].x.array[:] = p.phi_m_init.value
            p.wh[1][p.N_ions].x.array[:] = 0.0
        else:
            p.wh[0][p.N_ions].x.array[p.neuron_dofs] = p.phi_m_n_init.value
            p.wh[0][p.N_ions].x.array[p.glia_dofs] = p.phi_m_g_init.value
            p.wh[1][p.N_ions].x.array[:] = 0.0
        if self.nonzero_init_guess or self.direct_solver:
            be.pack()
        else:
            be.x.zero_()
        self.iterations = []
        self.solve_time = []
        self.assembly_time = []
        # traces, probe points, checkpoints (reference :96-99: init_png_savefile / init_checkpoint_file / init_data)
        from .output import RunOutput
        self.output = RunOutput(self) if (self.save_pngs or self.save_dat or self.save_cpoints or self.save_xdmfs or p.point_evaluation) else None
        if self.output is not None:
            self.output.record(0)

    # ---- reference :297-335
    def create_and_set_nullspace(self):
        self.print("Creating and setting null space ...")
        be = self.backend
        nrm = be.nullspace_test()
        a_scale = max(1e-300, float(self._matrix_scale()))
        assert nrm <= 1e-10 * a_scale, f"constant potential is not in the null space of A (||A ns|| = {nrm:.3e})"
        be.set_nullspace(True)
        be.project_nullspace(be.b)
        self.print("Null space set.")

    def _matrix_scale(self):
        return self.comm.allreduce_max(self.backend.matrix_max_abs()) + 1e-300

    def _sync(self):
        torch.cuda.synchronize()

    # ---- reference :337-501
    # The loop body never waits for the device: the timers the reference takes with perf_counter + allreduce(MAX) around assembly
    # and solve (:402-413, :434-449) are HIP events recorded on the stream and read once, at the end (or every step when the
    # problem is not ``quiet``, because the reference prints them per step).  ``prepare()`` / ``step(i)`` / ``finish()`` are the
    # three parts of ``solve()``; bench.py drives exactly these.
    def prepare(self):
        """Everything before the time loop (reference :351-362): solver setup, preconditioner matrix and hierarchy."""
        p = self.problem
        self.setup_breakdown = {}
        tic = time.perf_counter()
        self.setup_solver()
        self._sync()
        self.setup_breakdown["solver_setup_s"] = self.comm.allreduce_max(time.perf_counter() - tic)     # mesh graph + device upload
        self._setup_timer = self.setup_breakdown["solver_setup_s"]
        if self.use_P_mat and self._pc_kind in (_lib.PC_AMG, _lib.PC_AMG_BT, _lib.PC_AMG_LT):
            tic = time.perf_counter()
            p.setup_preconditioner(self.use_block_Jacobi)
            self.assemble_preconditioner()
            self._sync()
            self.setup_breakdown["preconditioner_setup_s"] = self.comm.allreduce_max(time.perf_counter() - tic)
            self._setup_timer += self.setup_breakdown["preconditioner_setup_s"]
        self._n_marked = 0
        self._prepared = True

    def _resolve_timers(self):
        """Read the step timers not yet accounted for (one synchronisation: ``knp_timer_read``) and append them to ``ode_time``,
        ``assembly_time`` and ``solve_time`` -- MAX over the ranks, as the reference's allreduce does.  Four marks per step:
        start | gating done | assembly done (both streams joined) | solve done."""
        if not getattr(self, "_n_marked", 0):
            return
        dts = self.backend.timer_read()
        n_steps, self._n_marked = self._n_marked, 0
        if dts.size < 4 * n_steps - 1:
            return
        arr = np.zeros((n_steps, 3))
        for k in range(n_steps):
            arr[k] = dts[4 * k:4 * k + 3]            # (the fourth interval of a step is unpack + output + the host's next-step work)
        if self.comm.size > 1:      # MAX over the ranks (device tensor under nccl, host tensor under gloo: same rule as Comm._reduce)
            tt = torch.as_tensor(arr, device=self.backend.device if self.comm.backend == "nccl" else "cpu")
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            arr = tt.cpu().numpy()
        for ode, asm, sol in arr:
            if self.problem.gating_variables:
                self.ode_time.append(float(ode))
            self.assembly_time.append(float(asm))
            self.solve_time.append(float(sol))
            self.tot_assembly_time += float(asm)
            self.tot_solver_time += float(sol)

    def step(self, i):
        """One pass of the reference's loop body for time step ``i`` (KNPEMIx_solver.py:365-474).  Returns False when the run ends
        here (``save_mat``)."""
        p = self.problem
        be = self.backend
        p.t.value += float(p.dt.value)
        self.print("\nTime step ", i)
        self.print("t (ms) = ", 1000 * float(p.t.value))
        if i == 1:
            tic = time.perf_counter()
            be.pc_setup(self._pc_kind)            # ksp.setOperators + ksp.setUp
            self._sync()
            self.setup_breakdown["pc_setup_s"] = self.comm.allreduce_max(time.perf_counter() - tic)
            self._setup_timer += self.setup_breakdown["pc_setup_s"]
            if self.view_input:
                self.view()
            tic = time.perf_counter()

        be.timer_mark()
        if p.gating_variables:
            for model in p.ionic_models:
                if isinstance(model, HodgkinHuxley):
                    model.update_t_mod()
                    model.update_gating_variables()
        be.timer_mark()
        self._b_is_final = i > 1          # step 1: the null-space projection still modifies b after the assembly
        self.assemble()
        if i > 1 and self.reassemble_P and (i % self.reassemble_N == 0) and self.use_P_mat and self._pc_kind in (_lib.PC_AMG, _lib.PC_AMG_BT, _lib.PC_AMG_LT):
            self.reassemble_preconditioner()
        be.timer_mark(join_assembly=True)
        if i == 1:
            self._sync()                      # first assembly: run-time compilation of the membrane programs, first launches
            self.setup_breakdown["first_assembly_s"] = self.comm.allreduce_max(time.perf_counter() - tic)

        if i == 1 and not p.dirichlet_bcs and not p.pin_ecs_potential:
            tic = time.perf_counter()
            self.create_and_set_nullspace()
            self._sync()
            self.setup_breakdown["nullspace_s"] = self.comm.allreduce_max(time.perf_counter() - tic)
            self._setup_timer += self.setup_breakdown["nullspace_s"]

        if self.save_mat:
            A = be.csr().tocoo()
            np.save(self.out_file_prefix + "Amat", np.c_[A.row, A.col, A.data])
            return False

        its, rnorm, reason = be.gmres(self._rtol, 1e-50, self.ksp_max_it, self.gmres_restart)
        self.ksp.its, self.ksp.rnorm, self.ksp.reason = its, rnorm, reason
        self.tot_its += its
        if self.direct_solver and not p.dirichlet_bcs and not p.pin_ecs_potential:
            # preonly + LU with the null space attached (reference :167-172, :331-333): PETSc removes the null-space
            # component from the SOLUTION, i.e. the potentials come back with zero mean over all potential unknowns --
            # the gauge of the reference's direct-solver pins (tests/test_oracle_pins.py::test_direct_solver_pin_without_any_fit)
            be.project_nullspace(be.x)
        be.timer_mark()
        self._n_marked += 1
        self.iterations.append(its)
        self.reasons.append(reason)
        if not getattr(p, "quiet", False):        # the reference prints both timers at every step: read them now
            self._resolve_timers()
            self.print(f"Time dependent assembly in {self.assembly_time[-1]:0.4f} seconds")
            self.print(f"Solved in {self.solve_time[-1]:0.4f} seconds ({its} its, {_lib.REASONS.get(reason, reason)}, |r| = {rnorm:.3e})")
        elif self._n_marked >= 512:
            self._resolve_timers()
        if reason < 0 and self.strict:
            raise RuntimeError(f"GMRES did not converge at step {i}: {_lib.REASONS.get(reason, reason)}")

        be.unpack()                                # x -> wh, phi_m_prev = phi_i - phi_e (reference :452-468)
        if self.output is not None:
            self.output.record(i)                  # reference :471-474 (checkpoint / trace / point evaluation)
        if p.MMS_test:
            p.print_errors()                       # reference :500-501
        return True

    def finish(self):
        """After the last step (reference :476-498): totals, ``print_info``, figures and exports."""
        self._resolve_timers()
        self.setup_time = self._setup_timer
        self.print("\nTotal setup time:", self._setup_timer)
        self.print("Total assembly time:", sum(self.assembly_time))
        self.print("Total solve time:", sum(self.solve_time))
        self.print_info()
        if self.save_pngs and self.output is not None:
            self.output.figures()
        if self.save_dat:
            self.export_data()

    def solve(self):
        self.prepare()
        try:
            for i in range(1, self.time_steps + 1):
                if self.step(i) is False:
                    return
            self.finish()
        finally:
            # a run that raises (strict non-convergence, NaN, a failed exchange) or is interrupted must still leave a readable
            # solution.h5: its metadata and superblock are written at close (reference: XDMFFile flushes per write, :482-484)
            try:
                self._resolve_timers()
            except Exception:      # noqa: BLE001  (a failed device must not mask the error that brought us here)
                pass
            if self.save_xdmfs and getattr(self, "output", None) is not None:
                self.output.close_xdmf()

    def potential_norms(self):
        """L2 norms of phi_i over Omega_i and phi_e over Omega_e (reference main.py:70-84)."""
        a, b = self.backend.l2_norms_sq()
        return float(np.sqrt(a)), float(np.sqrt(b))

    def export_data(self):
        """``.npy`` artefacts with the reference's file names (KNPEMIx_solver.py:833-866): membrane trace at the measurement
        vertex, gating variables there, probe-point values, timings and iteration counts."""
        if self.output is not None:
            self.output.export()
        elif self.comm.rank == 0:
            np.save(self.out_file_prefix + "assembly_time.npy", np.array(self.assembly_time))
            np.save(self.out_file_prefix + "solve_time.npy", np.array(self.solve_time))
            np.save(self.out_file_prefix + "iterations.npy", np.array(self.iterations))

    def view(self):
        """``--view 1`` / ``view_ksp``: what PETSc's -ksp_view reports for the reference (KNPEMIx_solver.py:285-288), for this solver."""
        pr = self.print
        be = self.backend
        pr("KSP Object: type gmres (native), restart", self.gmres_restart, ", classical Gram-Schmidt, left preconditioning, "
           "preconditioned-residual norm, one reduction per iteration")
        pr(f"  tolerances: relative={self._rtol:g}, absolute=1e-50, divergence=1e5, maximum iterations={self.ksp_max_it}")
        pr("  initial guess nonzero:", bool(self.nonzero_init_guess or self.direct_solver))
        st = be.stats()
        pr(f"PC Object: type {self.pc_type} -> native kind {self._pc_kind}", "(fused V(1,1) cycle)" if st["fused"] else "")
        for k, h in enumerate(getattr(self, "hierarchies", []) or []):
            if hasattr(h, "describe"):
                nf = getattr(h, "node_fields", 0)
                pr(f"  hierarchy {k}:", h.describe(),
                   (f"node-synchronised aggregation, {nf} fields per node" + (", node-blocked operators" if (st["blocked"] >> k) & 1 else "")) if nf else "")
        pr(f"  linear system: {be.n_dof_global} unknowns, {be.nnz_global} stored entries, membrane programs: {be.lib.knp_jit_status(be.ctx).decode()}")

    # ---- reference :504-548
    def print_info(self):
        p = self.problem
        be = self.backend
        pr = self.print
        pr("\n#------------ PROBLEM -------------#\n")
        pr("MPI Size = ", self.comm.size)
        pr("Input mesh = ", p.mesh_description)
        pr("Global # mesh cells = ", p.local_mesh.n_cells_global)
        pr("System size (global # dofs) = ", be.n_dof_global)
        pr("FEM order = ", p.fem_order)
        pr("# Time steps = ", self.time_steps)
        pr("dt = ", float(p.dt.value))
        pr("Using Dirichlet BCs." if p.dirichlet_bcs else "Using Neumann BCs.")
        pr("\n#------------ SOLVER -------------#\n")
        if self.direct_solver:
            pr("Direct solve emulated by GMRES+AMG (rtol 1e-13).")
        else:
            pr("Solver type: [" + self.ksp_type + "+" + self.pc_type + "]")
            pr(f"Tolerance: {self.ksp_rtol:.2e}")
            pr(f"Norm type: {self.norm_type}")
            pr(f"None-zero initial guess: {self.nonzero_init_guess}")
            if self.use_P_mat: pr("Preconditioner matrix P enabled.")
            if self.use_block_Jacobi: pr("Using block-Jacobi preconditioner form.")
            if self.reassemble_P: pr(f"Re-assembling preconditioner every {self.reassemble_N} timesteps.")
        if self.iterations:
            pr("Average iterations: " + str(sum(self.iterations) / len(self.iterations)))
