"""Two ranks sharing the one GPU of the test box (gloo for the exchanges, staged through the host): runs
the real library with its halo / all-reduce hooks and checks the partitioned solve against the oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, size, port, q):
    try:
        for p in (os.path.join(ROOT, "knp-emi-cgx_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
            sys.path.insert(0, p)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=size)
        from parity_utils import ci_config, run_native
        s = run_native(ci_config(N=16, steps=2, rtol=1e-13))
        ni, ne = s.potential_norms()
        lm = s.problem.local_mesh
        nvo = lm.n_vertices_owned
        phim = s.problem.phi_m_prev.numpy()
        q.put((rank, "ok", ni, ne, lm.l2g[:nvo].copy(), phim[:nvo].copy(), list(s.iterations), s.backend.n_dof_global))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:      # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


def test_two_ranks_one_gpu_match_oracle():
    size = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, size, port, q)) for r in range(size)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(size)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] == "ok", f"rank {r[0]}:\n{r[1]}"
    from parity_utils import run_oracle
    o = run_oracle(N=16, steps=2)
    oi, oe = o.potential_norms()
    phim = np.zeros(o.n_v)
    for r in res:
        assert abs(r[2] - oi) <= 1e-6 * oi
        assert abs(r[3] - oe) <= 1e-4 * oe
        assert r[7] == o.n_dof
        phim[r[4]] = r[5]
    gam = (o.lay.node_i >= 0) & (o.lay.node_e >= 0)
    assert np.allclose(phim[gam], o.phi_m[gam], rtol=1e-6)
