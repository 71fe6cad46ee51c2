"""Per-rank handle on libknpemi_hip: owns the ``knp_ctx`` and the torch tensors (device memory) that
back the block vectors.  Everything numerical goes through the C ABI (include/knpemi_hip.h)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import Fields, FieldsOut, KnpError, MeshDesc
from .parallel import HaloPlan, all_reduce_sum_, guarded


class _CAI:
    """Expose a raw HIP pointer through __cuda_array_interface__ so torch can view it."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def _f64(a):
    return a.ctypes.data_as(_lib.f64p)


def _i32(a):
    return a.ctypes.data_as(_lib.i32p)


class Backend:
    def __init__(self, problem):
        self.p = problem
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise KnpError("No HIP device visible: the KNP-EMI assemble-and-solve path runs on the GPU only "
                           "(there is deliberately no CPU fallback).")
        self.device = torch.device("cuda", torch.cuda.current_device())
        lm = problem.local_mesh
        dim = lm.coords.shape[1]
        self._keep = []
        coords = np.ascontiguousarray(lm.coords, dtype=np.float64)
        cells = np.ascontiguousarray(lm.cells, dtype=np.int32)
        side = np.ascontiguousarray(problem.cell_side, dtype=np.uint8)
        gamma = np.ascontiguousarray(lm.gamma, dtype=np.int32)
        tag_index = {t: k for k, t in enumerate(problem.gamma_tags)}
        self.tag_program = dict(getattr(problem, "tag_program", None) or {k: k for k in tag_index.values()})
        gprog = np.ascontiguousarray([self.tag_program[tag_index[int(t)]] for t in lm.gamma_tags], dtype=np.int32)
        qp = np.ascontiguousarray(problem.q_pts, dtype=np.float64)
        qw = np.ascontiguousarray(problem.q_w, dtype=np.float64)
        self._keep += [coords, cells, side, gamma, gprog, qp, qw]
        desc = MeshDesc()
        desc.dim = dim
        desc.n_vertices = coords.shape[0]
        desc.n_vertices_owned = lm.n_vertices_owned
        desc.n_cells = cells.shape[0]
        desc.n_cells_owned = lm.n_cells_owned
        desc.cells = _i32(cells)
        desc.coords = _f64(coords)
        desc.cell_side = side.ctypes.data_as(_lib.u8p)
        desc.n_gamma = gamma.shape[0]
        desc.gamma = _i32(gamma) if gamma.size else None
        desc.gamma_prog = _i32(gprog) if gprog.size else None
        desc.n_q = qw.shape[0]
        desc.q_pts = _f64(qp)
        desc.q_w = _f64(qw)
        ctx = C.c_void_p()
        rc = self.lib.knp_create(C.byref(ctx), C.byref(desc))
        self.ctx = ctx
        if rc != 0:
            msg = self.lib.knp_last_error(ctx) if ctx else b"allocation failed"
            if ctx:
                self.lib.knp_destroy(ctx)
                self.ctx = None
            raise KnpError(f"knp_create failed ({rc}): {msg.decode()}")
        self.check(self.lib.knp_set_stream(self.ctx, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        sz = (C.c_int64 * _lib.KNP_SZ_COUNT)()
        self.check(self.lib.knp_get_sizes(self.ctx, sz))
        self.sizes = list(sz)
        self.n_nodes = sz[_lib.SZ_N_NODES]
        self.n_nodes_owned = sz[_lib.SZ_N_NODES_OWNED]
        self.n_dof_local = sz[_lib.SZ_N_DOF_LOCAL]
        self.n_dof_owned = sz[_lib.SZ_N_DOF_OWNED]
        self.nnz = sz[_lib.SZ_NNZ]
        self.n_pairs = sz[_lib.SZ_N_PAIRS]
        self.n_contrib = sz[_lib.SZ_N_CONTRIB]
        self.node_i = np.empty(desc.n_vertices, dtype=np.int32)
        self.node_e = np.empty(desc.n_vertices, dtype=np.int32)
        self.check(self.lib.knp_get_layout(self.ctx, _i32(self.node_i), _i32(self.node_e)))
        self.n_dof_global = int(problem.comm.allreduce_sum(self.n_dof_owned))
        self.nnz_global = int(problem.comm.allreduce_sum(self.nnz))
        self.set_params()
        # block vectors
        self.b = torch.zeros(self.n_dof_local, dtype=torch.float64, device=self.device)
        self.x = torch.zeros(self.n_dof_local, dtype=torch.float64, device=self.device)
        # communication hooks
        self.halo = HaloPlan(problem.comm, lm, self.node_i, self.node_e, self.device)
        self._views = {}
        self.level_halos = {}          # (hier, level) -> LevelHalo of a distributed AMG hierarchy
        self.level_repl = {}           # (hier, level) -> replicated coarse size
        if problem.comm.size > 1:
            self._halo_cb = _lib.HALO_FN(guarded(self._halo))
            self._ar_cb = _lib.ALLREDUCE_FN(guarded(self._allreduce))
            self.check(self.lib.knp_set_comm(self.ctx, self._halo_cb, self._ar_cb, None))
            self._lc_cb = _lib.LEVEL_COMM_FN(guarded(self._level_comm))
            self.check(self.lib.knp_set_level_comm(self.ctx, self._lc_cb))
            self.p2p_setup()
        if getattr(problem, "programs", None):
            self.upload_programs()
        self.set_sources()
        self.setup_deflation()
        self.setup_dirichlet()

    def setup_dirichlet(self):
        """Dirichlet rows (MMS runs): extracellular unknowns at the exterior-boundary vertices."""
        p = self.p
        self.bc_dofs = None
        if not getattr(p, "bcs", None):
            return
        dofs, vals = [], []
        for side, f, verts, values in p.bcs:
            nodes = (self.node_e if side == "extra" else self.node_i)[verts]
            ok = (nodes >= 0) & (nodes < self.n_nodes_owned)
            dofs.append(4 * nodes[ok] + f)
            vals.append(np.asarray(values)[ok])
        dofs = np.ascontiguousarray(np.concatenate(dofs), dtype=np.int32)
        self.bc_dofs = torch.as_tensor(dofs.astype(np.int64), device=self.device)
        self.bc_vals = torch.as_tensor(np.concatenate(vals), dtype=torch.float64, device=self.device)
        self.check(self.lib.knp_set_dirichlet(self.ctx, len(dofs), _i32(dofs)))

    def apply_dirichlet_rhs(self):
        if self.bc_dofs is not None:
            self.b.index_copy_(0, self.bc_dofs, self.bc_vals)

    def setup_deflation(self):
        """Multi-GPU only: coarse correction for the floating-potential modes cut by the partition
        (E = Z^T A Z only involves the membrane capacitance terms: KNPEMIx_problem.py:637-638)."""
        p = self.p
        lm = p.local_mesh
        d = getattr(lm, "defl", None)
        if p.comm.size == 1 or not d or getattr(self, "_skip_deflation", False):
            return
        m = int(d["n_modes"])
        nvo = lm.n_vertices_owned
        node_mode = np.full(self.n_nodes_owned, -1, dtype=np.int32)
        vi = np.nonzero(self.node_i[:nvo] >= 0)[0]
        node_mode[self.node_i[vi]] = d["vertex_mode_i"][vi]
        ve = np.nonzero(self.node_e[:nvo] >= 0)[0]
        node_mode[self.node_e[ve]] = d["ecs_mode"]
        total = d["total_area"]
        if total is None:
            total = p.integrate_over_membrane(1.0, p.gamma_tags)
        kappa = float(p.C_M.value) / float(p.F.value)
        E = np.zeros((m, m))
        e = int(d["ecs_mode"])
        for k, a in enumerate(d["areas"]):
            E[k, k] = kappa * a
            E[k, e] = E[e, k] = -kappa * a
        E[e, e] = kappa * total
        Einv = np.ascontiguousarray(np.linalg.pinv(E, rcond=1e-12))
        self.deflation = {"E": E, "node_mode": node_mode}
        self.check(self.lib.knp_set_deflation(self.ctx, m, _i32(node_mode), _f64(Einv)))

    # ------------------------------------------------------------------ helpers
    def check(self, rc):
        _lib.check(self.ctx, rc)

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.lib.knp_destroy(self.ctx)
                self.ctx = None
        except Exception:      # noqa: BLE001
            pass

    def _view(self, ptr, n):
        key = (int(ptr), int(n))
        t = self._views.get(key)
        if t is None:
            t = torch.as_tensor(_CAI(ptr, n), device=self.device)
            if len(self._views) > 256:
                self._views.clear()
            self._views[key] = t
        return t

    def _halo(self, user, xptr):
        self.halo.exchange(self._view(xptr, self.n_dof_local))

    def _allreduce(self, user, ptr, n):
        all_reduce_sum_(self._view(ptr, n), self.p.comm)

    def _level_comm(self, user, hier, level, op, ptr):
        if op == 2:
            all_reduce_sum_(self._view(ptr, self.level_repl[(hier, level)]), self.p.comm)
            return
        h = self.level_halos[(hier, level)]
        x = self._view(ptr, h.n_loc)
        if op == 0:
            h.forward(x)
        else:
            h.reverse_add(x)

    # ---- native peer-to-peer exchange (csrc/knp_p2p.hip): rendezvous over torch.distributed, data path in the library
    P2P_FINE, P2P_LEVEL_HALO, P2P_LEVEL_REPL, P2P_SLOTS = 0, 1, 2, 3
    P2P_SELFTEST_REPS = 8

    def _all_ok(self, ok):
        return all(self.p.comm.all_gather_object(bool(ok)))

    def p2p_setup(self):
        """Enable the in-library exchange for the fine halo and the reduction slots.  Every step is collective and
        ends with a self-test against the torch.distributed path; any failure on any rank leaves ALL ranks on the
        hook path (KNP_COMM=hooks forces that)."""
        comm = self.p.comm
        self.p2p_on = False
        self._p2p_plans = {}
        if comm.size < 2 or comm.size > 16 or torch.device(self.device).type != "cuda":
            return False
        if os.environ.get("KNP_COMM", "p2p") != "p2p":
            return False
        rc = self.lib.knp_p2p_init(self.ctx, comm.rank, comm.size, float(os.environ.get("KNP_P2P_TIMEOUT", "30")))
        if not self._all_ok(rc == 0):
            self.lib.knp_p2p_shutdown(self.ctx)
            return False
        self.p2p_on = True
        self.halo._h.n_own, self.halo._h.n_loc = int(self.n_dof_owned), int(self.n_dof_local)
        slots = self._p2p_allreduce_plan(64)
        fine = self._p2p_halo_plan(self.halo._h) if slots is not None else None
        if slots is None or fine is None:
            self.p2p_on = False
            self.lib.knp_p2p_shutdown(self.ctx)
            if comm.rank == 0:
                print("native p2p exchange unavailable; using the torch.distributed hooks", flush=True)
            return False
        self.check(self.lib.knp_p2p_attach(self.ctx, self.P2P_SLOTS, 0, 0, slots))
        self.check(self.lib.knp_p2p_attach(self.ctx, self.P2P_FINE, 0, 0, fine))
        return True

    def _p2p_allreduce_plan(self, n_max):
        comm = self.p.comm
        handle = (C.c_char * 64)()
        plan = C.c_int32(-1)
        rc = self.lib.knp_p2p_plan_create(self.ctx, 1, int(n_max), 0, C.byref(plan), handle)
        info = comm.all_gather_object((rc, bytes(handle.raw)))
        if any(i[0] != 0 for i in info):
            return None
        handles = b"".join(i[1] for i in info)
        rc = self.lib.knp_p2p_plan_connect(self.ctx, plan.value, handles, 0, None, None, None, None, None, None, None)
        if not self._all_ok(rc == 0):
            return None
        # self-test: sums of rank-dependent vectors, identical bits on every rank; repeated so that both mailbox parities
        # and back-to-back exchanges are exercised (a flaky visibility problem should show up here, not in a solve)
        n = min(int(n_max), 64)
        good = True
        for rep in range(self.P2P_SELFTEST_REPS):
            base = torch.arange(n, dtype=torch.float64, device=self.device) * 0.5 + rep
            v = base + (comm.rank + 1)
            ref = base * comm.size + comm.size * (comm.size + 1) / 2
            # every rank runs every repetition whatever its own rc: leaving the loop alone would desynchronise the
            # collectives (a timed-out wait latches d_err, so later waits return at once instead of stalling)
            rc = self.lib.knp_p2p_test_allreduce(self.ctx, plan.value, C.c_void_p(v.data_ptr()), n)
            good = good and rc == 0 and bool(torch.equal(v, ref))
        return plan.value if self._all_ok(good) else None

    def _p2p_halo_plan(self, h):
        """Plan for one LevelHalo (cached: the same halo object may serve several attachments)."""
        if id(h) in self._p2p_plans:
            return self._p2p_plans[id(h)]
        comm = self.p.comm
        me = comm.rank
        peers = sorted(int(r) for r in h.peers)
        n_send = {r: int(len(h.send_idx[r])) if r in h.send_idx else 0 for r in peers}
        n_recv = {r: int(len(h.recv_idx[r])) if r in h.recv_idx else 0 for r in peers}
        send_ptr = np.concatenate([[0], np.cumsum([n_send[r] for r in peers])]).astype(np.int64)
        recv_ptr = np.concatenate([[0], np.cumsum([n_recv[r] for r in peers])]).astype(np.int64)
        n_fwd, n_rev = int(recv_ptr[-1]), int(send_ptr[-1])
        handle = (C.c_char * 64)()
        plan = C.c_int32(-1)
        rc = self.lib.knp_p2p_plan_create(self.ctx, 0, n_fwd, n_rev, C.byref(plan), handle) if len(peers) <= 16 else -1
        mine = {"rc": rc, "handle": bytes(handle.raw), "n_fwd": n_fwd, "n_rev": n_rev,
                "recv_off": {r: int(recv_ptr[j]) for j, r in enumerate(peers)},
                "send_off": {r: int(send_ptr[j]) for j, r in enumerate(peers)}}
        info = comm.all_gather_object(mine)
        if any(i["rc"] != 0 for i in info):
            self._p2p_plans[id(h)] = None
            return None
        handles = b"".join(i["handle"] for i in info)
        symmetric = all(me in info[r]["recv_off"] for r in peers)
        fwd_off = np.zeros(2 * len(peers), dtype=np.int64)
        rev_off = np.zeros(2 * len(peers), dtype=np.int64)
        if symmetric:
            for j, r in enumerate(peers):
                B = info[r]
                for par in (0, 1):
                    fwd_off[2 * j + par] = par * B["n_fwd"] + B["recv_off"][me]
                    rev_off[2 * j + par] = 2 * B["n_fwd"] + par * B["n_rev"] + B["send_off"][me]
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        send_idx = i32(h.send_all.cpu().numpy())
        recv_idx = i32(h.recv_all.cpu().numpy())
        pr = i32(peers)
        p32, p64 = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        rc = -1
        if symmetric:
            rc = self.lib.knp_p2p_plan_connect(self.ctx, plan.value, handles, len(peers), pr.ctypes.data_as(p32),
                                               send_ptr.ctypes.data_as(p64), send_idx.ctypes.data_as(p32), fwd_off.ctypes.data_as(p64),
                                               recv_ptr.ctypes.data_as(p64), recv_idx.ctypes.data_as(p32), rev_off.ctypes.data_as(p64))
        if not self._all_ok(rc == 0):
            self._p2p_plans[id(h)] = None
            return None
        # self-test against the torch.distributed exchange of the same halo, repeated (both mailbox parities,
        # back-to-back exchanges, forward and reverse interleaved)
        good = True
        for rep in range(self.P2P_SELFTEST_REPS):
            x = torch.zeros(h.n_loc, dtype=torch.float64, device=self.device)
            x[:h.n_own] = torch.arange(h.n_own, dtype=torch.float64, device=self.device) + 1e6 * (me + 1) + rep
            y = x.clone()
            h.forward(y)
            rc = self.lib.knp_p2p_test_halo(self.ctx, plan.value, C.c_void_p(x.data_ptr()), 0)
            good = good and rc == 0 and bool(torch.equal(x, y))
            x = torch.sin(torch.arange(h.n_loc, dtype=torch.float64, device=self.device) * (0.37 + rep) + me)
            y = x.clone()
            h.reverse_add(y)
            rc = self.lib.knp_p2p_test_halo(self.ctx, plan.value, C.c_void_p(x.data_ptr()), 1)
            good = good and rc == 0 and bool(torch.allclose(x, y, rtol=1e-13, atol=1e-13))
        ok = self._all_ok(good)
        self._p2p_plans[id(h)] = plan.value if ok else None
        self._p2p_keep = getattr(self, "_p2p_keep", []) + [send_idx, recv_idx]
        return self._p2p_plans[id(h)]

    def p2p_attach_level(self, hier, level, halo, repl_n=0):
        """Bind the native exchange to one level of a distributed hierarchy (collective; falls back to the hook)."""
        if not getattr(self, "p2p_on", False):
            return
        plan = self._p2p_halo_plan(halo)
        rp = self._p2p_allreduce_plan(int(repl_n)) if (repl_n and plan is not None) else None
        if plan is None or (repl_n and rp is None):
            # a failed self-test may have latched a timeout: drop the native path everywhere (same decision on all ranks)
            self.p2p_on = False
            self.lib.knp_p2p_shutdown(self.ctx)
            if self.p.comm.rank == 0:
                print("native p2p exchange failed its self-test on a hierarchy level; using the torch.distributed hooks", flush=True)
            return
        self.check(self.lib.knp_p2p_attach(self.ctx, self.P2P_LEVEL_HALO, hier, level, plan))
        if repl_n:
            self.check(self.lib.knp_p2p_attach(self.ctx, self.P2P_LEVEL_REPL, hier, level, rp))

    def dof_level_halo(self):
        """LevelHalo of the fine DoF vector layout (level 0 of a distributed hierarchy)."""
        from .dist_amg import LevelHalo
        p = self.p
        comm = p.comm
        counts = comm.all_gather_object(int(self.n_dof_owned))
        offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        start = int(offs[comm.rank])
        ids = torch.zeros(self.n_dof_local, dtype=torch.float64, device=self.device)
        ids[:self.n_dof_owned] = torch.arange(start, start + self.n_dof_owned, dtype=torch.float64, device=self.device)
        self.halo.exchange(ids)
        ghost_gid = np.rint(ids[self.n_dof_owned:].cpu().numpy()).astype(np.int64)
        ghost_owner = (np.searchsorted(offs, ghost_gid, side="right") - 1).astype(np.int64)
        return LevelHalo(comm, self.n_dof_owned, ghost_gid, ghost_owner, start, self.device), start, ghost_gid, ghost_owner

    def set_params(self):
        p = self.p
        z = np.array([float(ion["z"].value) for ion in p.ion_list], dtype=np.float64)
        Di = np.array([float(ion["Di"].value) for ion in p.ion_list], dtype=np.float64)
        De = np.array([float(ion["De"].value) for ion in p.ion_list], dtype=np.float64)
        self.check(self.lib.knp_set_params(self.ctx, float(p.dt.value), float(p.F.value), float(p.C_M.value),
                                           float(p.psi.value), len(p.ion_list), _f64(z), _f64(Di), _f64(De)))

    def set_sources(self):
        """Volume source functions f_i / f_e of the ion table (``source_terms: ion_injection``); Constants are the
        reference's zero placeholders."""
        from .fem import Function
        fi = (C.c_void_p * 3)()
        fe = (C.c_void_p * 3)()
        self._src_keep = []
        for j, ion in enumerate(self.p.ion_list[:3]):
            for arr, key in ((fi, "f_i"), (fe, "f_e")):
                f = ion.get(key)
                if isinstance(f, Function):
                    arr[j] = f.data_ptr()
                    self._src_keep.append(f)
                elif f is not None and float(getattr(f, "value", 0.0)) != 0.0:
                    raise NotImplementedError("constant non-zero volume sources: pass a Function (nodal values) instead")
        self.check(self.lib.knp_set_sources(self.ctx, fi, fe))

    def upload_programs(self):
        for pid, spec in self.p.programs.items():
            code = np.ascontiguousarray(spec.code, dtype=np.int32)
            consts = spec.constants()
            self.check(self.lib.knp_set_program(self.ctx, int(pid), code.shape[0], _i32(code), consts.shape[0],
                                                _f64(consts) if consts.size else None))
        self._programs_uploaded = True

    def refresh_program_constants(self):
        for pid, spec in self.p.programs.items():
            consts = spec.constants()
            if consts.size:
                self.check(self.lib.knp_set_program_constants(self.ctx, int(pid), consts.shape[0], _f64(consts)))

    def fields(self):
        """the nodal input fields as the ABI's struct of device pointers (built once: the tensors behind wh / phi_m / aux never move)"""
        p = self.p
        key = tuple(id(fn) for fn in getattr(p, "aux_functions", []))
        if getattr(self, "_fields_cache", None) is not None and self._fields_key == key:
            return self._fields_cache
        f = Fields()
        for j in range(3):
            f.k_i[j] = p.wh[0][j].data_ptr()
            f.k_e[j] = p.wh[1][j].data_ptr()
        f.phi_m = p.phi_m_prev.data_ptr()
        for k, fn in enumerate(getattr(p, "aux_functions", [])):
            f.aux[k] = fn.data_ptr()
        self._fields_cache, self._fields_key = f, key
        return f

    def fields_out(self):
        p = self.p
        if getattr(self, "_fields_out_cache", None) is not None:
            return self._fields_out_cache
        f = FieldsOut()
        for j in range(3):
            f.k_i[j] = p.wh[0][j].data_ptr()
            f.k_e[j] = p.wh[1][j].data_ptr()
        f.phi_i = p.wh[0][3].data_ptr()
        f.phi_e = p.wh[1][3].data_ptr()
        f.phi_m = p.phi_m_prev.data_ptr()
        self._fields_out_cache = f
        return f

    # ------------------------------------------------------------------ operations
    def assemble_matrix(self):
        f = self.fields()
        self.check(self.lib.knp_assemble_matrix(self.ctx, C.byref(f)))

    def assemble_matrix_async(self):
        """matrix assembly on the library's own stream (next to the right-hand side chain of the same step); joined by the solve"""
        f = self.fields()
        self.check(self.lib.knp_assemble_matrix_async(self.ctx, C.byref(f)))

    def assemble_rhs(self):
        if not getattr(self, "_programs_uploaded", False):
            self.upload_programs()
        self.refresh_program_constants()
        f = self.fields()
        self.check(self.lib.knp_assemble_rhs(self.ctx, C.byref(f), C.c_void_p(self.b.data_ptr())))

    def assemble_precond(self):
        f = self.fields()
        self.check(self.lib.knp_assemble_precond(self.ctx, C.byref(f)))

    def pack(self):
        f = self.fields_out()
        self.check(self.lib.knp_pack(self.ctx, C.byref(f), C.c_void_p(self.x.data_ptr())))

    def unpack(self):
        f = self.fields_out()
        self.check(self.lib.knp_unpack(self.ctx, C.c_void_p(self.x.data_ptr()), C.byref(f)))

    def hh_update(self, phi_m, n, m, h, dt, phi_rest, rush_larsen, substeps):
        self.check(self.lib.knp_hh_update(self.ctx, C.c_void_p(phi_m.data_ptr()), C.c_void_p(n.data_ptr()),
                                          C.c_void_p(m.data_ptr()), C.c_void_p(h.data_ptr()),
                                          int(phi_m.x.array.numel()), dt, phi_rest, int(rush_larsen), int(substeps)))

    def set_nullspace(self, on=True):
        self.check(self.lib.knp_set_nullspace(self.ctx, 1 if on else 0))

    def project_nullspace(self, vec: torch.Tensor):
        self.check(self.lib.knp_project_nullspace(self.ctx, C.c_void_p(vec.data_ptr())))

    def nullspace_test(self) -> float:
        out = C.c_double()
        self.check(self.lib.knp_nullspace_test(self.ctx, C.byref(out)))
        return out.value

    def pc_setup(self, kind):
        self.check(self.lib.knp_pc_setup(self.ctx, int(kind)))

    def pc_apply(self, r: torch.Tensor, z: torch.Tensor):
        self.check(self.lib.knp_pc_apply(self.ctx, C.c_void_p(r.data_ptr()), C.c_void_p(z.data_ptr())))

    def spmv(self, x: torch.Tensor, y: torch.Tensor):
        self.check(self.lib.knp_spmv(self.ctx, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr())))

    def gmres_prepare(self):
        """Start ||B b|| of the next solve on the library's side stream (b must be final); overlaps the matrix assembly."""
        self.check(self.lib.knp_gmres_prepare(self.ctx, C.c_void_p(self.b.data_ptr())))

    def gmres(self, rtol, atol=1e-50, max_it=5000, restart=30):
        its = C.c_int32()
        rn = C.c_double()
        reason = C.c_int32()
        self.check(self.lib.knp_gmres_solve(self.ctx, C.c_void_p(self.b.data_ptr()), C.c_void_p(self.x.data_ptr()),
                                            float(rtol), float(atol), int(max_it), int(restart), C.byref(its),
                                            C.byref(rn), C.byref(reason)))
        return its.value, rn.value, reason.value

    def matrix_max_abs(self) -> float:
        """max |A_ij| over the locally stored entries (device reduction over the pair-major arrays)."""
        out = C.c_double()
        self.check(self.lib.knp_matrix_max_abs(self.ctx, C.byref(out)))
        return out.value

    def l2_norms_sq(self):
        out = (C.c_double * 2)()
        p = self.p
        self.check(self.lib.knp_l2_norms(self.ctx, C.c_void_p(p.wh[0][3].data_ptr()), C.c_void_p(p.wh[1][3].data_ptr()), out))
        return p.comm.allreduce_sum(out[0]), p.comm.allreduce_sum(out[1])

    def total_ion_amounts(self):
        raise NotImplementedError

    # ---- exports (parity hooks) ------------------------------------------------------------
    def csr(self):
        """A as scipy CSR (rows = owned DoFs, cols = local DoFs), columns sorted."""
        import scipy.sparse as sp
        rp = np.empty(self.n_dof_owned + 1, dtype=np.int32)
        ci = np.empty(self.nnz, dtype=np.int32)
        va = np.empty(self.nnz, dtype=np.float64)
        self.check(self.lib.knp_get_csr_pattern(self.ctx, _i32(rp), _i32(ci)))
        self.check(self.lib.knp_get_csr_values(self.ctx, _f64(va)))
        A = sp.csr_matrix((va, ci, rp), shape=(self.n_dof_owned, self.n_dof_local))
        A.sort_indices()
        return A

    def precond_csr(self):
        import scipy.sparse as sp
        nnzp = self.sizes[_lib.SZ_NNZ_P]
        rp = np.empty(self.n_dof_owned + 1, dtype=np.int32)
        ci = np.empty(nnzp, dtype=np.int32)
        va = np.empty(nnzp, dtype=np.float64)
        self.check(self.lib.knp_get_precond_csr(self.ctx, _i32(rp), _i32(ci), _f64(va)))
        P = sp.csr_matrix((va, ci, rp), shape=(self.n_dof_owned, self.n_dof_local))
        P.sort_indices()
        return P

    def set_coupled_potential(self, on=True):
        self.check(self.lib.knp_pc_set_coupled_potential(self.ctx, 1 if on else 0))

    def precond_phi_csr(self, embedded=True):
        """The potential block of P (coupled across the membrane after ``set_coupled_potential``): node-indexed CSR, or -- ``embedded``
        -- placed at the potential unknowns 4*node+3 of an n_dof x n_dof matrix, which is what the hierarchy builders take."""
        import scipy.sparse as sp
        sz = (C.c_int64 * _lib.KNP_SZ_COUNT)()
        self.check(self.lib.knp_get_sizes(self.ctx, sz))
        nnz = sz[_lib.SZ_NNZ_P_PHI]
        no = self.n_nodes_owned
        rp = np.empty(no + 1, dtype=np.int32)
        ci = np.empty(nnz, dtype=np.int32)
        va = np.empty(nnz, dtype=np.float64)
        self.check(self.lib.knp_get_precond_phi_csr(self.ctx, _i32(rp), _i32(ci), _f64(va)))
        if not embedded:
            M = sp.csr_matrix((va, ci, rp), shape=(no, self.n_nodes))
            M.sort_indices()
            return M
        # CSR arrays written directly (row 4 node + 3 = the node's row, every other row empty): no detour through a COO matrix and its sort.
        # Explicit zeros are dropped -- right-angled simplices: half of the same-side pairs of a structured mesh have no stiffness coupling
        keep = va != 0.0
        node_of_entry = np.repeat(np.arange(no, dtype=np.int32), np.diff(rp))
        cnt = np.bincount(node_of_entry[keep], minlength=no)
        indptr = np.zeros(self.n_dof_owned + 1, dtype=np.int64)
        indptr[4 * np.arange(no, dtype=np.int64) + 4] = cnt
        np.cumsum(indptr, out=indptr)
        idx_t = np.int32 if max(int(indptr[-1]), self.n_dof_local) < 2 ** 31 - 1 else np.int64
        M = sp.csr_matrix((va[keep], (4 * ci[keep].astype(idx_t) + 3), indptr.astype(idx_t)), shape=(self.n_dof_owned, self.n_dof_local))
        M.sort_indices()           # (a check only when the library's rows are already sorted)
        return M

    # ---- step timers (events in the library: one ctypes call per mark, one synchronisation per read) ------------------------
    def timer_mark(self, join_assembly=False):
        self.check(self.lib.knp_timer_mark(self.ctx, 1 if join_assembly else 0))

    def timer_pending(self):
        return int(self.lib.knp_timer_pending(self.ctx))

    def timer_read(self):
        """seconds between consecutive marks since the last read (synchronises on the last mark)"""
        n = self.timer_pending()
        if n < 2:
            if n:
                out = (C.c_double * 1)()
                k = C.c_int32()
                self.check(self.lib.knp_timer_read(self.ctx, 1, out, C.byref(k)))
            return np.zeros(0)
        out = (C.c_double * n)()
        k = C.c_int32()
        self.check(self.lib.knp_timer_read(self.ctx, n, out, C.byref(k)))
        return np.array(out[:k.value], dtype=np.float64)

    # ---- instrumentation -----------------------------------------------------------------
    def profile_enable(self, mask=0x1f):
        self.check(self.lib.knp_profile_enable(self.ctx, int(mask)))

    def profile_reset(self):
        self.check(self.lib.knp_profile_reset(self.ctx))

    def stats(self):
        """||B b|| of the last solve and the exchange / read-back counters since the last ``profile_reset``."""
        out = (C.c_double * 8)()
        self.check(self.lib.knp_get_stats(self.ctx, out))
        return {"bnorm": out[0], "allreduces": int(out[1]), "halos": int(out[2]), "readbacks": int(out[3]), "fused": int(out[4]),
                "norm_fallbacks": int(out[5]), "blocked": int(out[6]), "fused_levels": int(out[7])}

    def traffic_model(self):
        """bytes one application of each kernel class must move (knp_get_traffic_model)"""
        out = (C.c_double * 5)()
        self.check(self.lib.knp_get_traffic_model(self.ctx, out))
        return {"spmv": out[0], "pc": out[1], "assembly_matrix": out[2], "assembly_rhs": out[3], "vector": out[4]}

    def profile_get(self):
        names = ["spmv", "orthogonalisation", "pc", "assembly", "other"]
        out = {}
        for k, nm in enumerate(names):
            ms = C.c_double()
            n = C.c_int64()
            self.check(self.lib.knp_profile_get(self.ctx, k, C.byref(ms), C.byref(n)))
            out[nm] = (ms.value, n.value)
        return out
