"""CPU baseline driver around oracle/knpemi_cpu.c -- TEST INFRASTRUCTURE / BASELINE ONLY, never imported by the product.

The same timestep as ``OracleKNPEMI.run(solver="gmres")`` (reference loop src/CGx/KNPEMI/KNPEMIx_solver.py:365-468), with the
kernels that carry the time on the host in C/OpenMP: cell-by-cell block assembly of A scatter-added into CSR, the CSR SpMV
behind every ``A @ x`` of GMRES and of the V-cycle levels, the dense coarsest-level solve.  GMRES(30), the V-cycle and the
block-triangular preconditioner are the oracle's own restatements (knpemi_oracle.gmres_left / pc_amg_vcycle / pc_btcc) applied
to matrices whose ``@`` is the C kernel; vector algebra goes through NumPy with the BLAS thread count set to the same number of
threads.  Timed at 1 thread and at all cores of the box.
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np
import scipy.sparse as sp

import knpemi_oracle as K

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libknpemi_cpu.so")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: build it with __graft_entry__.build() (gcc -O3 -fopenmp)")
        L = C.CDLL(path)
        L.knp_cpu_max_threads.restype = C.c_int
        L.knp_cpu_set_threads.argtypes = [C.c_int]
        L.knp_cpu_spmv.argtypes = [C.c_int, _i32p, _i32p, _f64p, _f64p, _f64p]
        L.knp_cpu_dense_matvec.argtypes = [C.c_int, _f64p, _f64p, _f64p]
        L.knp_cpu_assemble_volume.argtypes = [C.c_int, C.c_int, _i32p, _f64p, _f64p, _f64p, C.c_double, C.c_double, _f64p, _f64p,
                                              C.c_int64, _f64p]
        L.knp_cpu_scatter_add.argtypes = [C.c_int64, _i32p, _f64p, _f64p]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


class CMat:
    """CSR matrix whose product with a vector runs in the C/OpenMP kernel"""

    def __init__(self, A):
        A = sp.csr_matrix(A)
        self.shape = A.shape
        self.rp = np.ascontiguousarray(A.indptr, dtype=np.int32)
        self.ci = np.ascontiguousarray(A.indices, dtype=np.int32)
        self.v = np.ascontiguousarray(A.data, dtype=np.float64)

    def __matmul__(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(self.shape[0])
        lib().knp_cpu_spmv(self.shape[0], _p(self.rp, _i32p), _p(self.ci, _i32p), _p(self.v, _f64p), _p(x, _f64p), _p(y, _f64p))
        return y


class CDense:
    def __init__(self, M):
        self.M = np.ascontiguousarray(M, dtype=np.float64)

    def __matmul__(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(self.M.shape[0])
        lib().knp_cpu_dense_matvec(self.M.shape[0], _p(self.M, _f64p), _p(x, _f64p), _p(y, _f64p))
        return y


class _Lv:
    pass


def wrap_hierarchy(h):
    """levels with C-kernel operators (values as given: the caller passes the fp32-rounded ones when the GPU stores fp32)"""
    out = _Lv()
    out.levels = []
    for lv in h.levels:
        w = _Lv()
        w.A, w.dinv, w.lambda_max = CMat(lv.A), lv.dinv, lv.lambda_max
        w.P = CMat(lv.P) if lv.P is not None else None
        w.R = CMat(lv.R) if lv.R is not None else None
        w.S = CMat(lv.S) if getattr(lv, "S", None) is not None else None
        w.Pt = CMat(lv.Pt) if getattr(lv, "Pt", None) is not None else CMat(lv.A @ sp.diags(lv.dinv))
        out.levels.append(w)
    out.coarse_inv = CDense(h.coarse_inv) if h.coarse_inv is not None else None
    return out


class Twin:
    def __init__(self, o: "K.OracleKNPEMI"):
        self.o = o
        A = o.assemble_A()
        self.n = A.shape[0]
        self.rp = np.ascontiguousarray(A.indptr, dtype=np.int32)
        self.ci = np.ascontiguousarray(A.indices, dtype=np.int32)
        self.nnz = A.nnz
        keys = (np.repeat(np.arange(self.n, dtype=np.int64), np.diff(A.indptr)) * self.n + A.indices)      # sorted (CSR, sorted columns)
        self._keys = keys

        def slot(r, c):
            k = np.searchsorted(keys, r.astype(np.int64) * self.n + c.astype(np.int64))
            assert np.array_equal(keys[k], r.astype(np.int64) * self.n + c)
            return k.astype(np.int32)
        R4, C4 = 4 * o.rowsA, 4 * o.colsA                      # (nc, nl, nl) node indices of each local pair
        sl = np.empty(R4.shape + (10,), dtype=np.int32)
        for j in range(3):
            sl[..., j] = slot(R4 + j, C4 + j)
            sl[..., 3 + j] = slot(R4 + j, C4 + 3)
            sl[..., 6 + j] = slot(R4 + 3, C4 + j)
        sl[..., 9] = slot(R4 + 3, C4 + 3)
        self.slots = np.ascontiguousarray(sl)
        self.Mloc = np.ascontiguousarray(o.Mloc, dtype=np.float64)
        self.Kloc = np.ascontiguousarray(o.Kloc, dtype=np.float64)
        self._slot = slot
        self.vals = np.zeros(self.nnz)

    def assemble_A(self):
        o, p = self.o, self.o.p
        nc, nl = o.cells.shape
        cbar = np.empty((3, nc))
        for j in range(3):
            kj = np.where(o.cell_side[:, None] == 0, o.k[0][j][o.cells], o.k[1][j][o.cells])
            cbar[j] = kj.mean(axis=1)
        D = np.ascontiguousarray(p.D, dtype=np.float64)
        z = np.ascontiguousarray(p.z, dtype=np.float64)
        lib().knp_cpu_assemble_volume(nc, nl, _p(self.slots, _i32p), _p(self.Mloc, _f64p), _p(self.Kloc, _f64p), _p(cbar, _f64p),
                                      p.dt, p.psi, _p(D, _f64p), _p(z, _f64p), self.nnz, _p(self.vals, _f64p))
        # membrane terms: the oracle's facet integrals (a few thousand facets on these meshes), scatter-added
        r, c, v = self._membrane_coo()
        if r.size:
            sl = self._slot(r, c)
            vv = np.ascontiguousarray(v, dtype=np.float64)
            lib().knp_cpu_scatter_add(len(sl), _p(sl, _i32p), _p(vv, _f64p), _p(self.vals, _f64p))
        M = CMat.__new__(CMat)
        M.shape, M.rp, M.ci, M.v = (self.n, self.n), self.rp, self.ci, self.vals
        return M

    def _membrane_coo(self):
        """rows / cols / values of the membrane blocks exactly as OracleKNPEMI.assemble_A builds them"""
        o, p = self.o, self.o.p
        F, C_M = p.F, p.C_M
        d = o.dim
        ni = np.repeat(o.fnode_i[:, :, None], d, axis=2)
        ne = np.repeat(o.fnode_e[:, :, None], d, axis=2)
        nib = np.repeat(o.fnode_i[:, None, :], d, axis=1)
        neb = np.repeat(o.fnode_e[:, None, :], d, axis=1)
        al_i, al_e = o._alpha_q(0), o._alpha_q(1)
        rows, cols, vals = [], [], []
        for j in range(3):
            Ci = o._facet_mass(al_i[j] * C_M / (F * p.z[j]))
            Ce = o._facet_mass(al_e[j] * C_M / (F * p.z[j]))
            rows += [4 * ni + j, 4 * ni + j, 4 * ne + j, 4 * ne + j]
            cols += [4 * nib + 3, 4 * neb + 3, 4 * neb + 3, 4 * nib + 3]
            vals += [Ci, -Ci, Ce, -Ce]
        Mg = (C_M / F) * o._facet_mass()
        rows += [4 * ni + 3, 4 * ni + 3, 4 * ne + 3, 4 * ne + 3]
        cols += [4 * nib + 3, 4 * neb + 3, 4 * neb + 3, 4 * nib + 3]
        vals += [Mg, -Mg, Mg, -Mg]
        if not rows:
            return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0)
        return (np.concatenate([a.ravel() for a in rows]), np.concatenate([a.ravel() for a in cols]),
                np.concatenate([a.ravel() for a in vals]))

    def step(self, x, M, ns, rtol):
        o = self.o
        o.t += o.p.dt
        for mdl in o.models:
            if mdl.kind == "hh":
                o.update_t_mod()
                o.update_gating(mdl)
        A = self.assemble_A()
        b = o.assemble_b()
        x, it, _ = K.gmres_left(A, b, x, M, ns=ns, rtol=rtol)
        o.unpack(x)
        return x, it


def usable_cores():
    """cores this process may actually use: affinity mask and cgroup CPU quota (v2 cpu.max, v1 cfs quota), whichever is smaller"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: (t.split()[0], t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            txt = open(path).read().strip()
            if parse:
                q, per = parse(txt)
                if q != "max":
                    n = min(n, max(1, int(float(q) / float(per) + 0.5)))
            else:
                q = int(txt)
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError, IndexError):
            pass
    return n


def time_kernels(o, pc_factory, rtol=1e-9, budget_s=8.0, max_steps=3):
    """Continue the oracle's run with the C/OpenMP kernels, at 1 thread and at all cores.  ``pc_factory(wrap)`` returns the
    preconditioner built on ``wrap(hierarchy)`` operators.  Returns the dict that goes into cpu_baseline["omp_twin"]."""
    from threadpoolctl import threadpool_limits
    L = lib()
    ncores = int(L.knp_cpu_max_threads())
    tw = Twin(o)
    M = pc_factory(wrap_hierarchy)
    ns = o.nullspace()
    out = {"kind": "port-omp", "unit": "MDoF/s", "host_cores": ncores,
           "what": "same timestep as the NumPy oracle with assembly, CSR SpMV (GMRES and every V-cycle level) and the dense coarse solve in "
                   "C/OpenMP (oracle/knpemi_cpu.c); not DOLFINx/PETSc"}
    # "all cores": the box may expose more hardware threads than this job may use (cgroup share, affinity): candidates are the
    # powers of two up to the usable count, the parallel leg runs with the one a short SpMV probe finds fastest, and says so
    A0 = tw.assemble_A()
    xv = np.ones(A0.shape[1])
    best, best_t = 1, None
    usable = usable_cores()
    out["usable_cores"] = usable
    cands = [1] + [c for c in (2, 4, 8, 16, 32, 64, 128, 256, 512) if c <= usable]
    if usable not in cands:
        cands.append(usable)
    for c in cands:
        L.knp_cpu_set_threads(c)
        A0 @ xv
        t1 = time.perf_counter()
        for _ in range(3):
            A0 @ xv
        dt_ = time.perf_counter() - t1
        if best_t is None or dt_ < best_t:
            best, best_t = c, dt_
    out["parallel_threads_chosen_by_probe"] = best
    x = o.pack()
    for nt in ((1, best) if best > 1 else (1,)):
        L.knp_cpu_set_threads(nt)
        with threadpool_limits(limits=nt):
            t_all, its = [], []
            t0 = time.perf_counter()
            while len(t_all) < max_steps and (not t_all or time.perf_counter() - t0 + t_all[-1] < budget_s):
                t1 = time.perf_counter()
                x, it = tw.step(x, M, ns, rtol)
                t_all.append(time.perf_counter() - t1)
                its.append(it)
        sec = sum(t_all) / len(t_all)
        out["threads_1" if nt == 1 else "threads_all"] = {"cores": nt, "value": o.n_dof / sec / 1e6, "s_per_step": sec, "steps": len(t_all),
                                                            "its_per_step": sum(its) / len(its)}
    L.knp_cpu_set_threads(1)
    return out
