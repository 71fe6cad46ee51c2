"""The smoothed-aggregation setup of ``amg.py`` on the GPU.

Same algorithm, same random priorities (so the aggregates are the ones the NumPy version finds), but every heavy step
runs on the device through torch: the Galerkin products and the prolongator smoothing are sparse CSR x CSR products
(hipSPARSE SpGEMM behind ``torch.sparse.mm``), the power iterations are device SpMVs, strength / aggregation are
segmented reductions.  On a 3.7 M-DoF cube the host version needs 17 s for the two hierarchies of the block-triangular
preconditioner, which is more than 2 000 timesteps' worth of solves; this one needs about a second.

The result is handed back as SciPy matrices in an ``amg.Hierarchy`` (what ``amg.upload`` and the parity tests consume).
PyTorch is plumbing here (sparse products, sorting), exactly as for device memory and ``torch.distributed``.
"""
from __future__ import annotations

import warnings

import numpy as np
import scipy.sparse as sp
import torch

from . import amg

warnings.filterwarnings("ignore", message="Sparse CSR tensor support is in beta state")
_THETA_DECAY = amg._THETA_DECAY


class _Csr:
    """CSR matrix on the device: crow (n+1), col (nnz, sorted within a row), val (nnz)."""

    __slots__ = ("crow", "col", "val", "shape")

    def __init__(self, crow, col, val, shape):
        self.crow, self.col, self.val, self.shape = crow, col, val, (int(shape[0]), int(shape[1]))

    @property
    def nnz(self):
        return int(self.col.numel())

    def rows(self):
        n = self.shape[0]
        return torch.repeat_interleave(torch.arange(n, device=self.col.device), self.crow[1:] - self.crow[:-1])

    def torch(self):
        return torch.sparse_csr_tensor(self.crow, self.col, self.val, size=self.shape)

    def scipy(self):
        # index conversion on the device (half the transfer, no single-threaded NumPy pass over 10^8 entries); every _Csr comes out of
        # _from_coo / _from_scipy with sorted rows, so SciPy's own O(nnz) check is skipped as well
        idx = torch.int32 if max(self.shape[1], self.nnz) < 2 ** 31 - 1 else torch.int64
        M = sp.csr_matrix((self.val.cpu().numpy(), self.col.to(idx).cpu().numpy(), self.crow.to(idx).cpu().numpy()), shape=self.shape)
        M.has_sorted_indices = True
        return M

    def diagonal(self):
        r = self.rows()
        d = torch.zeros(self.shape[0], dtype=self.val.dtype, device=self.val.device)
        m = r == self.col
        d.index_add_(0, r[m], self.val[m])
        return d

    def matvec(self, x):
        return (self.torch() @ x.unsqueeze(1)).squeeze(1)


def _from_scipy(M, device):
    M = sp.csr_matrix(M, dtype=np.float64)
    M.sort_indices()
    # indices travel as they are (int32 for everything below 2^31 entries) and are widened on the device
    return _Csr(torch.as_tensor(M.indptr, device=device).to(torch.int64), torch.as_tensor(M.indices, device=device).to(torch.int64),
                torch.as_tensor(M.data, device=device), M.shape)


def _restrict_fields(A: _Csr, fields, block: int = 4) -> _Csr:
    """``amg.restrict_to_fields`` on the device: rows and columns of the other fields dropped (same size), explicit zeros dropped"""
    dev = A.val.device
    n = A.shape[0]
    fmask = torch.zeros(block, dtype=torch.bool, device=dev)
    fmask[list(fields)] = True
    r = A.rows()
    keep = fmask[r % block] & fmask[A.col % block] & (A.val != 0)
    r = r[keep]
    crow = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    crow[1:] = torch.cumsum(torch.bincount(r, minlength=n), 0)
    return _Csr(crow, A.col[keep], A.val[keep], A.shape)


def _from_coo(r, c, v, shape, drop_zeros=False):
    """COO (duplicates summed) -> CSR with sorted columns"""
    n, m = int(shape[0]), int(shape[1])
    key = r * m + c
    key, inv = torch.unique(key, return_inverse=True)            # sorted
    val = torch.zeros(key.numel(), dtype=v.dtype, device=v.device)
    val.index_add_(0, inv, v)
    if drop_zeros:
        keep = val != 0
        key, val = key[keep], val[keep]
    rr = torch.div(key, m, rounding_mode="floor")
    cc = key - rr * m
    crow = torch.zeros(n + 1, dtype=torch.int64, device=v.device)
    crow[1:] = torch.cumsum(torch.bincount(rr, minlength=n), 0)
    return _Csr(crow, cc, val, (n, m))


def _spgemm(A: _Csr, B: _Csr) -> _Csr:
    C = torch.sparse.mm(A.torch(), B.torch())
    out = _Csr(C.crow_indices().to(torch.int64), C.col_indices().to(torch.int64), C.values(), (A.shape[0], B.shape[1]))
    # hipSPARSE returns unsorted columns within a row: sort through the COO route
    return _from_coo(out.rows(), out.col, out.val, out.shape)


def _transpose(A: _Csr) -> _Csr:
    return _from_coo(A.col, A.rows(), A.val, (A.shape[1], A.shape[0]))


def _row_max(crow, vals_at_cols, fill):
    if vals_at_cols.numel() == 0:
        return torch.full((crow.numel() - 1,), fill, dtype=torch.float64, device=crow.device)
    return torch.segment_reduce(vals_at_cols, "max", offsets=crow, initial=float(fill))


def _lambda_max(A: _Csr, dinv, iters=20, seed=1):
    rng = np.random.default_rng(seed)
    x = torch.as_tensor(rng.standard_normal(A.shape[0]), device=dinv.device)
    x = x / torch.linalg.norm(x)
    At = A.torch()
    lam = 1.0
    for _ in range(iters):
        y = dinv * (At @ x.unsqueeze(1)).squeeze(1)
        lam = float(torch.linalg.norm(y))
        if lam == 0.0:
            return 1.0
        x = y / lam
    return amg._LAMBDA_PAD * lam


def _strength(A: _Csr, theta):
    """symmetric SA strength pattern (no values): sorted keys r*n+c"""
    n = A.shape[0]
    r, c, v = A.rows(), A.col, A.val
    d = torch.abs(A.diagonal())
    keep = (r != c) & (torch.abs(v) >= theta * torch.sqrt(d[r] * d[c])) & (v != 0)
    r, c = r[keep], c[keep]
    key = torch.unique(torch.cat([r * n + c, c * n + r]))
    return key


def _pattern_csr(key, n):
    rr = torch.div(key, n, rounding_mode="floor")
    cc = key - rr * n
    crow = torch.zeros(n + 1, dtype=torch.int64, device=key.device)
    crow[1:] = torch.cumsum(torch.bincount(rr, minlength=n), 0)
    return crow, cc


def _aggregate(crow, col, n, seed, distance):
    """amg.aggregate on the device (same priorities, same rounds)"""
    dev = crow.device
    rng = np.random.default_rng(seed)
    prio = torch.as_tensor(rng.permutation(n).astype(np.float64) + 1.0, device=dev)
    state = torch.zeros(n, dtype=torch.int8, device=dev)
    for _ in range(200):
        und = state == 0
        if not bool(und.any()):
            break
        w = torch.where(und, prio, torch.zeros_like(prio))
        m1 = torch.maximum(w, _row_max(crow, w[col], 0.0))
        m2 = torch.maximum(m1, _row_max(crow, m1[col], 0.0)) if distance >= 2 else m1
        new_root = und & (w >= m2)
        state[new_root] = 1
        r = new_root.to(torch.float64)
        r1 = torch.maximum(r, _row_max(crow, r[col], 0.0))
        r2 = torch.maximum(r1, _row_max(crow, r1[col], 0.0)) if distance >= 2 else r1
        state[(state == 0) & (r2 > 0)] = 2
    roots = torch.nonzero(state == 1).squeeze(1)
    agg = torch.full((n,), -1, dtype=torch.int64, device=dev)
    agg[roots] = torch.arange(roots.numel(), device=dev)
    rootid = torch.where(state == 1, agg, torch.full_like(agg, -1)).to(torch.float64)
    nb = _row_max(crow, rootid[col], -1.0)
    take = (agg < 0) & (nb >= 0)
    agg[take] = nb[take].to(torch.int64)
    node_of_prio = torch.empty(n + 2, dtype=torch.int64, device=dev)
    node_of_prio[prio.to(torch.int64)] = torch.arange(n, device=dev)
    for _ in range(3):
        left = agg < 0
        if not bool(left.any()):
            break
        key = torch.where(agg >= 0, prio, torch.full_like(prio, -1.0))
        best = _row_max(crow, key[col], -1.0)
        ok = left & (best > 0)
        agg[ok] = agg[node_of_prio[best[ok].to(torch.int64)]]
    left = torch.nonzero(agg < 0).squeeze(1)
    nagg = int(roots.numel())
    if left.numel():
        agg[left] = nagg + torch.arange(left.numel(), device=dev)
        nagg += int(left.numel())
    return agg, nagg


class _Laps:
    """KNP_AMG_TIMING=1: where the device-side setup spends its time (synchronising laps; developer aid)"""

    def __init__(self, device):
        import os
        self.on = bool(os.environ.get("KNP_AMG_TIMING")) and str(device).startswith("cuda")
        self.t, self.last = {}, 0.0
        if self.on:
            import time
            torch.cuda.synchronize()
            self.last = time.perf_counter()

    def __call__(self, name):
        if self.on:
            import time
            torch.cuda.synchronize()
            now = time.perf_counter()
            self.t[name] = self.t.get(name, 0.0) + now - self.last
            self.last = now

    def report(self, rows):
        if self.on:
            import sys
            print("[amg_gpu] levels", rows, {k: round(v, 3) for k, v in self.t.items()}, file=sys.stderr, flush=True)


def build_hierarchy(P, theta: float = 0.08, max_levels: int = 12, coarse_size: int = 2500, agg_distance=2, device="cuda", node_fields=None,
                    split_decoupled: bool = True, smoother_degree: int = 1, eliminate_independent: bool = True, fields=None):
    """Device version of ``amg.build_hierarchy`` (same arguments, same kind of result; ``fields``: P is the matrix of all fields and the
    hierarchy is built on ``amg.restrict_to_fields(P, fields)``, restricted on the device; ``node_fields``: aggregation on the node
    graph of the first field, shared by all fields; ``split_decoupled``: unknowns without off-diagonal entries are solved by the
    smoother and not carried to coarser levels, only with a degree-1 smoother -- see amg.build_hierarchy)."""
    split_decoupled = bool(split_decoupled) and int(smoother_degree) == 1
    eliminate_independent = bool(eliminate_independent) and int(smoother_degree) == 1
    lap = _Laps(device)
    A = _from_scipy(P, device)
    if fields is not None:
        A = _restrict_fields(A, fields)
        A_host = A.scipy()
    else:
        A_host = sp.csr_matrix(P, dtype=np.float64)
        A_host.sort_indices()
    lap("level-0 upload")
    levels = []
    after_elimination = False
    sync = node_fields is not None and len(node_fields[1]) > 1
    stride, fields = (int(node_fields[0]), tuple(int(f) for f in node_fields[1])) if sync else (1, (0,))
    while True:
        n = A.shape[0]
        diag = A.diagonal()
        active = diag != 0
        dinv = torch.where(active, 1.0 / torch.where(active, diag, torch.ones_like(diag)), torch.zeros_like(diag))
        lam = _lambda_max(A, dinv)
        lap("lambda_max")
        iso = torch.zeros(n, dtype=torch.bool, device=diag.device)
        if split_decoupled:
            r_a, c_a, v_a = A.rows(), A.col, A.val
            offd = (r_a != c_a) & (v_a != 0)
            iso = active & (torch.bincount(r_a[offd], minlength=n) == 0)
            if sync and bool(iso.any()):       # a node is split off only if all of its fields are decoupled
                iso_n = iso[fields[0]::stride].clone()
                for f in fields[1:]:
                    iso_n &= iso[f::stride]
                iso = torch.zeros(n, dtype=torch.bool, device=diag.device)
                for f in fields:
                    iso[f::stride] = iso_n
        any_iso = bool(iso.any())
        n_core = int((active & ~iso).sum())
        n_all = int(active.sum())
        if n_all <= coarse_size or n_core == 0 or len(levels) >= max_levels - 1 or after_elimination:
            levels.append(amg.Level(A_host, dinv.cpu().numpy(), lam))
            break
        inject = n_core <= coarse_size and n_all > amg.DENSE_LIMIT
        if not inject and not sync and eliminate_independent and n_core <= amg.ELIMINATION_LEVEL_MAX:
            # small level: the independent-set elimination of amg.build_hierarchy, done by the same host code
            el = amg.try_elimination_level(A_host, diag.cpu().numpy(), dinv.cpu().numpy(), active.cpu().numpy(), iso.cpu().numpy(), lam,
                                           coarse_size, len(levels))
            if el is not None:
                levels.append(el[0])
                A_host = el[1]
                A = _from_scipy(A_host, diag.device)
                after_elimination = True
                continue
        if any_iso:
            dinv = torch.where(iso, dinv / amg.cheby_first_coefficient(lam), dinv)
            active = active & ~iso
        n_act = n_core
        th = theta * _THETA_DECAY ** len(levels)
        if sync:
            # node graph of the first field -> aggregates of nodes -> the same aggregates and strength pattern for every field
            nf, nn, f0 = len(fields), n // stride, fields[0]
            r_, c_, v_ = A.rows(), A.col, A.val
            m0 = (r_ % stride == f0) & (c_ % stride == f0)
            A0 = _from_coo(torch.div(r_[m0], stride, rounding_mode="floor"), torch.div(c_[m0], stride, rounding_mode="floor"), v_[m0], (nn, nn))
            skey_n = _strength(A0, th)
            if not levels:      # same compatibility rule as the host builder (amg.build_hierarchy)
                deg0 = torch.bincount(torch.div(skey_n, nn, rounding_mode="floor"), minlength=nn)
                for f in fields[1:]:
                    mf = (r_ % stride == f) & (c_ % stride == f)
                    Af = _from_coo(torch.div(r_[mf], stride, rounding_mode="floor"), torch.div(c_[mf], stride, rounding_mode="floor"), v_[mf], (nn, nn))
                    degf = torch.bincount(torch.div(_strength(Af, th), nn, rounding_mode="floor"), minlength=nn)
                    if bool(((degf == 0) & (deg0 > 0)).any()) or bool(((diag[f::stride] <= 0) & (diag[f0::stride] > 0)).any()):
                        return build_hierarchy(P, theta, max_levels, coarse_size, agg_distance, device, None, split_decoupled, smoother_degree, eliminate_independent,
                                               fields)
            act_n = active[f0::stride]
            ian = torch.nonzero(act_n).squeeze(1)
            n_act_n = int(ian.numel())
            if inject:
                agg_n, nagg_n = torch.arange(n_act_n, device=diag.device), n_act_n
            elif n_act_n == nn:
                crow_s, col_s = _pattern_csr(skey_n, nn)
                agg_n, nagg_n = _aggregate(crow_s, col_s, nn, len(levels), amg._dist(agg_distance, len(levels)))
            else:
                newid = torch.cumsum(act_n.to(torch.int64), 0) - 1
                rr = torch.div(skey_n, nn, rounding_mode="floor")
                cc = skey_n - rr * nn
                both = act_n[rr] & act_n[cc]
                crow_s, col_s = _pattern_csr(newid[rr[both]] * n_act_n + newid[cc[both]], n_act_n)
                agg_n, nagg_n = _aggregate(crow_s, col_s, n_act_n, len(levels), amg._dist(agg_distance, len(levels)))
            rn = torch.div(skey_n, nn, rounding_mode="floor")
            cn = skey_n - rn * nn
            skey = torch.sort(torch.cat([(stride * rn + f) * n + (stride * cn + f) for f in fields]))[0]
            ia = torch.cat([stride * ian + f for f in fields])
            agg = torch.cat([nf * agg_n + k for k in range(nf)])
            nagg = nf * nagg_n
        else:
            skey = _strength(A, th)
            # aggregation on the active sub-graph
            ia = torch.nonzero(active).squeeze(1)
            if inject:
                agg, nagg = torch.arange(n_act, device=diag.device), n_act
            elif n_act == n:
                crow_s, col_s = _pattern_csr(skey, n)
                agg, nagg = _aggregate(crow_s, col_s, n, len(levels), amg._dist(agg_distance, len(levels)))
            else:
                newid = torch.cumsum(active.to(torch.int64), 0) - 1
                rr = torch.div(skey, n, rounding_mode="floor")
                cc = skey - rr * n
                both = active[rr] & active[cc]
                sub = newid[rr[both]] * n_act + newid[cc[both]]
                crow_s, col_s = _pattern_csr(sub, n_act)
                agg, nagg = _aggregate(crow_s, col_s, n_act, len(levels), amg._dist(agg_distance, len(levels)))
        lap("strength + aggregation")
        if nagg >= 0.9 * n_act and not inject:
            levels.append(amg.Level(A_host, dinv.cpu().numpy(), lam))
            break
        dev = A.val.device
        ones = torch.ones(n_act, dtype=torch.float64, device=dev)
        T = _from_coo(ia, agg, ones, (n, nagg))
        # filtered matrix: weak off-diagonals lumped onto the diagonal
        r, c, v = A.rows(), A.col, A.val
        keyA = r * n + c
        pos = torch.searchsorted(skey, keyA)
        pos = torch.clamp(pos, max=max(skey.numel() - 1, 0))
        strong = (skey[pos] == keyA) if skey.numel() else torch.zeros_like(keyA, dtype=torch.bool)
        keepF = strong | (r == c)
        rowsum = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, r, v)
        rowsumF = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, r[keepF], v[keepF])
        lump = rowsum - rowsumF
        AF = _from_coo(torch.cat([r[keepF], torch.arange(n, device=dev)]), torch.cat([c[keepF], torch.arange(n, device=dev)]),
                       torch.cat([v[keepF], lump]), (n, n))
        dF = AF.diagonal()
        okF = dF != 0
        dFinv = torch.where(okF, 1.0 / torch.where(okF, dF, torch.ones_like(dF)), torch.zeros_like(dF))
        if inject:
            Pm = T                                           # the coupled unknowns move to a level of their own, unchanged
        else:
            lamF = _lambda_max(AF, dFinv, iters=15)
            omega = 4.0 / (3.0 * lamF)
            AFT = _spgemm(AF, T)
            rAFT = AFT.rows()
            Pm = _from_coo(torch.cat([T.rows(), rAFT]), torch.cat([T.col, AFT.col]),
                           torch.cat([T.val, -omega * dFinv[rAFT] * AFT.val]), (n, nagg), drop_zeros=True)
        lap("prolongator")
        R = _transpose(Pm)
        AP = _spgemm(A, Pm)
        Ac = _spgemm(R, AP)
        lap("Galerkin product")
        # S = (I - c2 Dinv A) Pm: prolongation + post-smoothing step of the fused cycle as one operator
        rAP = AP.rows()
        c2 = amg.cheby_first_coefficient(lam)
        S = _from_coo(torch.cat([Pm.rows(), rAP]), torch.cat([Pm.col, AP.col]), torch.cat([Pm.val, -c2 * dinv[rAP] * AP.val]), (n, nagg))
        lap("S")
        Rt_h = U_h = None
        if levels:      # intermediate level: both legs of the fused cycle as plain products (amg.coarse_fused_operators)
            rA, cA, vA = A.rows(), A.col, A.val
            AD = _Csr(A.crow, A.col, vA * dinv[cA], A.shape)
            RAD = _spgemm(R, AD)
            Rt = _from_coo(torch.cat([R.rows(), RAD.rows()]), torch.cat([R.col, RAD.col]), torch.cat([R.val, -c2 * RAD.val]), (nagg, n))
            dg = torch.arange(n, device=dev)
            U = _from_coo(torch.cat([rA, dg, S.rows()]), torch.cat([cA, dg, S.col + n]),
                          torch.cat([-(c2 * c2) * dinv[rA] * vA * dinv[cA], 2.0 * c2 * dinv, S.val]), (n, n + nagg))
            Rt_h, U_h = Rt.scipy(), U.scipy()
        lap("Rt, U")
        levels.append(amg.Level(A_host, dinv.cpu().numpy(), lam, Pm.scipy(), R.scipy(), S.scipy(), Rt_h, U_h))
        A = Ac
        A_host = Ac.scipy()
        lap("copies to the host")
        if sync:
            stride, fields = len(fields), tuple(range(len(fields)))
    # the dense pseudo-inverse stays on the host (LAPACK): the device eigen-solver is not accurate enough for the nearly
    # singular potential block (residual |A A+ A - A| of 0.2-0.4 instead of 1e-13 on MI355X / ROCm 7.2)
    last = levels[-1].A
    coarse_inv = amg.dense_pseudo_inverse(last) if last.shape[0] <= amg.DENSE_LIMIT else None
    h = amg.Hierarchy(levels, coarse_inv)
    h.node_fields = len(node_fields[1]) if sync else 0
    lap("dense inverse")
    lap.report([lv.A.shape[0] for lv in levels])
    return h
