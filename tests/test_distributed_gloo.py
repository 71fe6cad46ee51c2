"""N>1 path on CPU: two gloo ranks exercise the partitioner, the ghost layer, the halo plan and exchange
and the all-reduce hook -- the exact Python plumbing libknpemi_hip calls back into on the GPU -- with the
oracle's NumPy assembly standing in for the local compute (tests only)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, size, port, kind, N, generator, q):
    try:
        for p in (os.path.join(ROOT, "knp-emi-cgx_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
            sys.path.insert(0, p)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=size)
        import knpemi_oracle as K
        from cgx_hip import mesh as meshmod
        from cgx_hip.parallel import (Comm, HaloPlan, all_reduce_sum_, partition_mesh, stacked_cubes_local_mesh,
                                      stacked_squares_local_mesh)
        comm = Comm()
        assert comm.size == size and comm.rank == rank
        # ---- global problem (every rank builds it to check against)
        if generator in ("partition", "kway"):
            gen = meshmod.create_unit_square if kind == "square" else meshmod.create_unit_cube
            c, t = gen(N)
            tags = meshmod.mark_subdomains_box(c, t)
            g, gt, _ = meshmod.gamma_integration_entities(t, tags, (1,), (2,))
            # "kway": the multilevel graph partition of the weighted nodal graph (cgx_hip/partition.py) instead of coordinate bisection
            lm = partition_mesh(c * 1e-6, t, tags, g, gt, size, rank, intra_tags=(1,) if generator == "kway" else None,
                                method="kway" if generator == "kway" else None)
            og = K.OracleKNPEMI(c, t, tags, models=K.CI_MODELS(), mesh_conversion_factor=1e-6)
        else:
            gen = stacked_squares_local_mesh if kind == "square" else stacked_cubes_local_mesh
            lm = gen(N, size, rank, scale=1e-6)
            # global mesh = concatenation of the owned parts: rebuild it from a single-rank generator of the
            # stacked domain by gathering every rank's owned vertices / cells
            parts = comm.all_gather_object((lm.l2g[:lm.n_vertices_owned], lm.coords[:lm.n_vertices_owned],
                                            lm.l2g[lm.cells[:lm.n_cells_owned]], lm.cell_tags[:lm.n_cells_owned]))
            nvg = lm.n_vertices_global
            c = np.zeros((nvg, lm.coords.shape[1]))
            for gid, xy, _, _ in parts:
                c[gid] = xy
            t = np.concatenate([p[2] for p in parts]).astype(np.int32)
            tags = np.concatenate([p[3] for p in parts])
            og = K.OracleKNPEMI(c, t, tags, models=K.CI_MODELS(), mesh_conversion_factor=1.0)
        # ---- local problem on the rank's piece (owned + ghost layer), assembled by the oracle
        ol = K.OracleKNPEMI(lm.coords, lm.cells, lm.cell_tags, gamma=lm.gamma, gamma_tag=lm.gamma_tags,
                            models=K.CI_MODELS(), mesh_conversion_factor=1.0)
        # same smooth non-uniform state on both
        def state(o, gid_coords):
            s = 1.0 + 0.05 * np.sin(3e6 * gid_coords[:, 0] + 1.0) * np.cos(2e6 * gid_coords[:, 1] + 0.5)
            for side in range(2):
                for j in range(3):
                    o.k[side][j] = o.k[side][j] * s
        state(og, og.coords)
        state(ol, ol.coords)
        ol.stimulus_area = og.stimulus_area
        Ag = og.assemble_A()
        Al = ol.assemble_A()
        nvo = lm.n_vertices_owned
        n_owned_nodes = int(((ol.lay.node_i[:nvo] >= 0).sum() + (ol.lay.node_e[:nvo] >= 0).sum()))
        n_own = 4 * n_owned_nodes
        # map local dofs -> global dofs
        def gdof(o_loc, o_glob, l2g):
            out = np.empty(o_loc.n_dof, dtype=np.int64)
            for nodes_l, nodes_g in ((o_loc.lay.node_i, o_glob.lay.node_i), (o_loc.lay.node_e, o_glob.lay.node_e)):
                v = np.nonzero(nodes_l >= 0)[0]
                assert (nodes_g[l2g[v]] >= 0).all()
                for f in range(4):
                    out[4 * nodes_l[v] + f] = 4 * nodes_g[l2g[v]] + f
            return out
        l2gd = gdof(ol, og, lm.l2g)
        # owned rows of the local matrix equal the corresponding global rows
        Al_own = Al[:n_own].tocoo()
        Gsub = Ag[l2gd[:n_own]].tocsr()
        rows, cols, vals = Al_own.row, l2gd[Al_own.col], Al_own.data
        import scipy.sparse as sp
        Al_glob = sp.csr_matrix((vals, (rows, cols)), shape=(n_own, og.n_dof))
        diff = abs(Al_glob - Gsub).max()
        assert diff <= 1e-12 * abs(Ag).max(), f"owned rows differ from the global operator: {diff}"
        bl = ol.assemble_b()[:n_own]
        bg = og.assemble_b()[l2gd[:n_own]]
        assert np.allclose(bl, bg, rtol=1e-11, atol=1e-13 * np.abs(bg).max())
        # ---- halo exchange + distributed SpMV == global SpMV
        plan = HaloPlan(comm, lm, ol.lay.node_i.astype(np.int32), ol.lay.node_e.astype(np.int32), torch.device("cpu"))
        rng = np.random.default_rng(5)
        xg = rng.standard_normal(og.n_dof)
        xl = torch.zeros(ol.n_dof, dtype=torch.float64)
        xl[:n_own] = torch.as_tensor(xg[l2gd[:n_own]])
        plan.exchange(xl)
        assert np.array_equal(xl.numpy(), xg[l2gd]), "ghost values wrong after halo exchange"
        yl = Al[:n_own] @ xl.numpy()
        assert np.allclose(yl, (Ag @ xg)[l2gd[:n_own]], rtol=1e-11, atol=1e-13 * np.abs(Ag @ xg).max())
        # ---- all-reduce hook: distributed dot == global dot; every dof owned exactly once
        cnt = torch.tensor([float(n_own)], dtype=torch.float64)
        all_reduce_sum_(cnt, comm)
        assert int(cnt.item()) == og.n_dof
        d = torch.tensor([float(xl[:n_own].numpy() @ xl[:n_own].numpy())], dtype=torch.float64)
        all_reduce_sum_(d, comm)
        assert abs(d.item() - xg @ xg) <= 1e-10 * (xg @ xg)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("kind,N,generator", [("square", 12, "partition"), ("square", 8, "stacked"), ("cube", 4, "stacked"), ("square", 16, "kway")])
def test_two_rank_halo_and_reductions(kind, N, generator):
    size = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, size, port, kind, N, generator, q)) for r in range(size)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(size)]
    for p in procs:
        p.join(timeout=60)
    for r, msg in res:
        assert msg == "ok", f"rank {r}:\n{msg}"


def _setup_worker(rank, size, port, q):
    try:
        for p in (os.path.join(ROOT, "knp-emi-cgx_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
            sys.path.insert(0, p)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=size)
        import scipy.sparse as sp
        from cgx_hip import amg, dist_amg
        from cgx_hip.parallel import Comm, exchange_arrays
        comm = Comm()
        # ---- packed point-to-point exchange: ints and floats, empty and missing messages
        out = {}
        for r in range(size):
            if r != rank and (rank + r) % 2 == 1:
                out[r] = [np.arange(rank * 10, rank * 10 + r + 3, dtype=np.int64), np.linspace(0, 1, r + 2) + rank, np.zeros(0)]
        got = exchange_arrays(comm, out)
        for r in range(size):
            if r != rank and (rank + r) % 2 == 1:
                a, b, c = got[r]
                assert a.dtype == np.int64 and np.array_equal(a, np.arange(r * 10, r * 10 + rank + 3))
                assert np.allclose(b, np.linspace(0, 1, rank + 2) + r) and c.size == 0
            else:
                assert r not in got
        # ---- distributed hierarchy of a 2D Laplacian split by rows: every exchange of the setup goes through exchange_arrays
        n1 = 24
        T = sp.diags([-1, 2.0001, -1], [-1, 0, 1], shape=(n1, n1))
        A = (sp.kron(sp.identity(n1), T) + sp.kron(T, sp.identity(n1))).tocsr()
        n = A.shape[0]
        lo, hi = rank * n // size, (rank + 1) * n // size
        rows = A[lo:hi].tocsr()
        cols = np.unique(rows.indices)
        ghost = cols[(cols < lo) | (cols >= hi)]
        cmap = np.full(n, -1)
        cmap[lo:hi] = np.arange(hi - lo)
        cmap[ghost] = (hi - lo) + np.arange(len(ghost))
        Aloc = sp.csr_matrix((rows.data, cmap[rows.indices], rows.indptr), shape=(hi - lo, hi - lo + len(ghost)))
        owner = np.minimum(ghost * size // n, size - 1)
        owner = np.array([r for g in ghost for r in range(size) if r * n // size <= g < (r + 1) * n // size])
        halo = dist_amg.LevelHalo(comm, hi - lo, ghost, owner, lo, "cpu")
        levels, tail = dist_amg.build_distributed_hierarchy(comm, Aloc, halo, lo, ghost, owner, coarse_size=20, replicate_below=60, device="cpu")
        desc = dist_amg.describe(levels, tail, comm)
        # the replicated tail is the same matrix on every rank, and its first operator is the Galerkin product of the serial setup's size
        sig = (desc["rows"], [round(float(abs(lv.A).sum()), 9) for lv in tail.levels]) if tail is not None else (desc["rows"], [])
        sigs = comm.all_gather_object(sig)
        assert all(s == sigs[0] for s in sigs)
        assert desc["rows"][0] == n and len(desc["rows"]) >= 3
        # per-level aggregation distance (SolverKNPEMI.ion_agg_distance: "2,1" in 3D): distance-1 aggregates below the finest level give
        # larger coarse levels and at least as many of them; the schedule continues into the replicated tail, identical on every rank
        lv21, tail21 = dist_amg.build_distributed_hierarchy(comm, Aloc, halo, lo, ghost, owner, coarse_size=20, replicate_below=60, device="cpu",
                                                            agg_distance=[2, 1])
        d21 = dist_amg.describe(lv21, tail21, comm)
        assert d21["rows"][:2] == desc["rows"][:2] and len(d21["rows"]) >= len(desc["rows"]) and d21["rows"][2] > desc["rows"][2], (d21, desc)
        sig21 = (d21["rows"], [round(float(abs(lv.A).sum()), 9) for lv in tail21.levels]) if tail21 is not None else (d21["rows"], [])
        sigs21 = comm.all_gather_object(sig21)
        assert all(s == sigs21[0] for s in sigs21)
        q.put((rank, "ok"))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:      # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("size", [2, 3])
def test_packed_setup_exchange_and_distributed_hierarchy(size):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_setup_worker, args=(r, size, port, q)) for r in range(size)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(size)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] == "ok", f"rank {r[0]}:\n{r[1]}"
