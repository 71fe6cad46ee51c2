"""A run fed from the reference's mesh file format (XDMF + HDF5, tests/golden/xdmf) through the C ABI, against the oracle on
the arrays that were written into those files (src/CGx/utils/mixed_dim_problem.py:634-681 is the reader this replaces)."""
import os

import numpy as np
import pytest

from parity_utils import ci_config, run_native

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "xdmf")


@pytest.mark.parametrize("name,kind", [("square8", "square"), ("cube3_mesh", "cube")])
def test_run_from_xdmf_files_matches_oracle_on_the_same_arrays(name, kind):
    import knpemi_oracle as K
    E = np.load(os.path.join(G, "expected.npz"))
    pre = "sq" if kind == "square" else "cu"
    cfg = ci_config(N=8, steps=2, rtol=1e-13, kind=kind)
    cfg["input_dir"] = G + os.sep
    if kind == "square":
        cfg["cell_tag_file"], cfg["facet_tag_file"] = "square8.xdmf", "square8_facets.xdmf"
    else:
        cfg["cell_tag_file"], cfg["facet_tag_file"] = "cube3_mesh.xdmf", "cube3_facets.xdmf"
    cfg["solver"]["ksp_settings"]["ksp_max_it"] = 5000
    cfg["solver"]["ksp_settings"]["amg_coarse_size"] = 100
    s = run_native(cfg)
    assert "XDMF" in s.problem.mesh_description and all(r > 0 for r in s.reasons)
    lm = s.problem.local_mesh
    assert lm.cells.shape[0] == len(E[pre + "_cells"]) and set(np.unique(lm.gamma_tags)) == {4}
    o = K.OracleKNPEMI(E[pre + "_coords"], E[pre + "_cells"].astype(np.int32), E[pre + "_ct"], models=K.CI_MODELS(), mesh_conversion_factor=1e-6)
    o.run(2, solver="lu_gauge")
    ni, ne = s.potential_norms()
    oi, oe = o.potential_norms()
    assert abs(ni - oi) <= 1e-6 * oi and abs(ne - oe) <= 1e-5 * oe
    gam = (o.lay.node_i >= 0) & (o.lay.node_e >= 0)
    assert np.allclose(s.problem.phi_m_prev.numpy()[gam], o.phi_m[gam], rtol=1e-6)
    for j in range(3):
        vi, ve = o.lay.node_i >= 0, o.lay.node_e >= 0
        assert np.allclose(s.problem.wh[0][j].numpy()[vi], o.k[0][j][vi], rtol=1e-7)
        assert np.allclose(s.problem.wh[1][j].numpy()[ve], o.k[1][j][ve], rtol=1e-7)
