"""The oracle's MMS restatement against the reference's recorded verification errors
(src/CGx/utils/errors.py:8-28; five levels, found to be N = 8, 16, 32, 64, 128 in 2D and 3D).  CPU only.
This pins the 2D *and* the 3D discretisation (mesh splits, subdomain/facet markers, '+' = intra orientation,
forms incl. all membrane terms, facet quadrature, Dirichlet handling) to numbers produced by the reference."""
import numpy as np
import pytest

import mms_oracle as M


def test_mms_2d_matches_recorded_errors_and_converges_quadratically():
    errs = np.array([M.run_mms(2, N) for N in (8, 16, 32, 64)])
    rec = M.RECORDED_2D[:4]
    # potentials: 5 significant digits on every level
    assert np.allclose(errs[:, 6:], rec[:, 6:], rtol=2e-5)
    # concentrations: the reference's error functional differs slightly in quadrature (<= 0.5 %)
    assert np.allclose(errs[:, :6], rec[:, :6], rtol=5e-3)
    rates = np.log2(errs[:-1] / errs[1:])
    assert np.all(rates[-1] > 1.97) and np.all(rates[-1] < 2.03)


@pytest.mark.parametrize("N,level", [(8, 0), (16, 1)])
def test_mms_3d_matches_recorded_errors(N, level):
    e = M.run_mms(3, N)
    rec = M.RECORDED_3D[level]
    assert np.allclose(e[6:], rec[6:], rtol=2e-5)
    assert np.allclose(e[:6], rec[:6], rtol=5e-3)
