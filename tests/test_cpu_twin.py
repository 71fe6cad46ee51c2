"""The C/OpenMP CPU-baseline kernels (oracle/knpemi_cpu.c, test infrastructure) against the NumPy oracle they restate."""
import numpy as np
import pytest

import knpemi_oracle as K
from cgx_hip import amg
from parity_utils import make_oracle


@pytest.mark.parametrize("kind,N", [("square", 12), ("cube", 4)])
def test_c_twin_assembly_spmv_and_step_match_the_oracle(kind, N):
    import knpemi_cpu_twin as T
    o = make_oracle(N, kind)
    o2 = make_oracle(N, kind)
    for oo in (o, o2):          # non-uniform state so that every block is exercised
        X = oo.coords / oo.coords.max()
        s = 1.0 + 0.05 * np.sin(3.0 * X[:, 0] + 1.0) * np.cos(2.0 * X[:, 1] + 0.5)
        for side in range(2):
            for j in range(3):
                oo.k[side][j] = oo.k[side][j] * (s if (side + j) % 2 == 0 else 2.0 - s)
    tw = T.Twin(o)
    A = tw.assemble_A()
    Ao = o.assemble_A()
    assert np.array_equal(A.rp, Ao.indptr) and np.array_equal(A.ci, Ao.indices)
    assert np.max(np.abs(A.v - Ao.data)) <= 1e-13 * np.max(np.abs(Ao.data))
    x = np.random.default_rng(0).standard_normal(o.n_dof)
    assert np.max(np.abs(A @ x - Ao @ x)) <= 1e-12 * np.max(np.abs(Ao @ x))
    # one timestep: same GMRES, same V-cycle, operators applied by the C kernels
    P = o.assemble_P()
    h = amg.build_hierarchy(P, coarse_size=100)
    M_c = (lambda hh: K.pc_amg_vcycle(hh.levels, hh.coarse_inv, 1, 1, 1))(T.wrap_hierarchy(h))
    M_o = K.pc_amg_vcycle(h.levels, h.coarse_inv, 1, 1, 1)
    ns = o.nullspace()
    xc, itc = tw.step(o.pack(), M_c, ns, 1e-10)
    xo, its = o2.run(1, solver="gmres", pc=lambda PP: M_o, rtol=1e-10)
    assert itc == its[0]
    for f in range(4):
        assert np.max(np.abs(xc[f::4] - xo[f::4])) <= 1e-9 * np.max(np.abs(xo[f::4])), f


def test_time_kernels_reports_one_and_all_cores():
    import knpemi_cpu_twin as T
    o = make_oracle(16)
    P = o.assemble_P()
    h = amg.build_hierarchy(P, coarse_size=100)
    out = T.time_kernels(o, lambda wrap: (lambda hh: K.pc_amg_vcycle(hh.levels, hh.coarse_inv, 1, 1, 1))(wrap(h)), budget_s=1.0, max_steps=2)
    assert out["kind"] == "port-omp" and out["threads_1"]["cores"] == 1 and out["threads_1"]["value"] > 0
    assert out["parallel_threads_chosen_by_probe"] >= 1
    if out["parallel_threads_chosen_by_probe"] > 1:
        assert out["threads_all"]["cores"] == out["parallel_threads_chosen_by_probe"] and out["threads_all"]["value"] > 0
