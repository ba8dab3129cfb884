"""Pins the CPU oracle to every known answer the reference's own sources/tests
hold for the regrid path."""
import numpy as np
import pytest

from oracle import oracle as orc


def test_linterp_1d_known_answer():
    # IceRegridder.hpp:150-155: xpoints {3,5,6,8}, xx=6.5 -> indices {2,3}, weights {.75,.25}
    idx, w = orc.linterp_1d([3, 5, 6, 8], 6.5)
    assert idx.tolist() == [2, 3]
    assert w.tolist() == [0.75, 0.25]
    idx, w = orc.linterp_1d_b([3, 5, 6, 8], 6.5)
    assert idx.tolist() == [2, 3]
    assert w.tolist() == [0.75, 0.25]


def test_linterp_1d_b_bounds():
    hc = np.arange(40) * 100.0 - 50.0      # write_icebin_in_base.py:44
    # below the first point: clamps to (0,1) and extrapolates (IceRegridder_L0.cpp:83)
    idx, w = orc.linterp_1d_b(hc, -80.0)
    assert idx.tolist() == [0, 1]
    assert w[1] == pytest.approx(-0.3) and w[0] == pytest.approx(1.3)
    # exactly on a point: lower_bound gives that point as i1 -> weight 1 on it
    idx, w = orc.linterp_1d_b(hc, 150.0)
    assert idx.tolist() == [1, 2] and w.tolist() == [0.0, 1.0]
    # above the last point: error (IceRegridder_L0.cpp:84-85)
    with pytest.raises(orc.OracleError):
        orc.linterp_1d_b(hc, 3851.0)
    # the two-sided variant clamps instead (IceRegridder.cpp:194-195)
    idx, w = orc.linterp_1d(hc, 3950.0)
    assert idx.tolist() == [38, 39] and w[1] == pytest.approx(2.0)


def test_nearest_1d():
    # IceRegridder_L0.cpp:43-67: ties go to the lower point
    xp = [0.0, 10.0, 20.0]
    assert orc.nearest_1d(xp, -5) == 0
    assert orc.nearest_1d(xp, 4.9) == 0
    assert orc.nearest_1d(xp, 5.0) == 0
    assert orc.nearest_1d(xp, 5.1) == 1
    assert orc.nearest_1d(xp, 25) == 2


def test_elev_mask_fixture(elev_mask_g20):
    # mask counts {4: 5102, 2: 4456, 0: 1158}; max elevation 3672 m < 3850 m
    m = elev_mask_g20["mask"]
    assert m.shape == (76, 141)
    vals, cnt = np.unique(m, return_counts=True)
    assert dict(zip(vals.tolist(), cnt.tolist())) == {0: 1158, 2: 4456, 4: 5102}
    land, ice = orc.elevmask_pism(m, elev_mask_g20["thk"], elev_mask_g20["topg"])   # ElevMask.cpp:57-70
    assert np.isfinite(ice).sum() == 4456
    assert np.isfinite(land).sum() == 5614
    assert np.nanmax(ice) < 3850.0 and np.nanmax(ice) == pytest.approx(3672.2594790403978)
    sel = (m.reshape(-1) == 2)
    np.testing.assert_array_equal(ice[sel], (elev_mask_g20["topg"] + elev_mask_g20["thk"]).reshape(-1)[sel])


def test_sparse_set_first_seen():
    s = orc.SparseSet(100)
    assert [s.add_dense(k) for k in (7, 3, 7, 99, 3, 0)] == [0, 1, 0, 2, 1, 3]
    assert s.to_sparse().tolist() == [7, 3, 99, 0]
    assert s.dense_extent == 4 and s.sparse_extent == 100


def test_coo_matvec_fill_and_nan():
    # pylib/icebin_cython.cpp:158-192: untouched rows keep fill; ignore_nan skips NaN inputs
    import scipy.sparse
    M = scipy.sparse.coo_matrix(([1.0, 2.0, 3.0], ([0, 0, 2], [0, 1, 2])), shape=(3, 3))
    y = orc.coo_matvec(M, [1.0, 1.0, 1.0], fill=-9.0)
    assert y.tolist() == [3.0, -9.0, 3.0]
    y = orc.coo_matvec(M, [1.0, np.nan, 1.0], fill=-9.0, ignore_nan=True)
    assert y.tolist() == [1.0, -9.0, 3.0]
    y = orc.coo_matvec(M, [1.0, np.nan, 1.0], fill=-9.0, ignore_nan=False)
    assert np.isnan(y[0]) and y[1] == -9.0
