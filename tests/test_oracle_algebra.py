"""Independent checks of the oracle's matrix algebra: a hand-computed case, a
scipy re-derivation of every compute_* branch from the Ur matrices, and the edge
semantics SURVEY.md 7 lists (pre-populated dims, zero-area cells, emission order)."""
import numpy as np
import pytest
import scipy.sparse as sp

from icebin_amd import synthetic as syn
from oracle import oracle as orc


def hand_grids():
    # 2 A cells (sparse ids 5, 9 of nA=12), 4 I cells, 5 exchange cells; I cell 2 straddles both A cells.
    ex_indices = np.array([[5, 0], [5, 1], [5, 2], [9, 2], [9, 3]], np.int32)
    ex_area = np.array([4.0, 4.0, 1.0, 3.0, 4.0])
    return dict(nI=4, nA=12, ex_indices=ex_indices, ex_area=ex_area,
                A_to_sparse=np.array([5, 9]), A_native_area=np.array([10.0, 9.0]),
                A_proj_area=np.array([8.0, 9.0]), hcdefs=np.array([0.0, 100.0, 200.0]),
                hc_stride_A=1, hc_stride_HC=12, interp_style=0)


def test_hand_computed_AvI():
    rg = orc.Regridder(hand_grids())
    em = np.array([50.0, np.nan, 150.0, 0.0])       # I cell 1 masked
    w = rg.matrix_d("AvI", em, scale=False, correctA=False)
    assert w.dims[0].tolist() == [5, 9] and w.dims[1].tolist() == [0, 2, 3]
    assert w.to_scipy().toarray().tolist() == [[4.0, 1.0, 0.0], [0.0, 3.0, 4.0]]
    assert w.wM.tolist() == [5.0, 7.0] and w.Mw.tolist() == [4.0, 4.0, 4.0]
    s = rg.matrix_d("AvI", em, scale=True, correctA=False)
    assert s.to_scipy().toarray().tolist() == [[0.8, 0.2, 0.0], [0.0, 3.0 / 7.0, 4.0 / 7.0]]
    c = rg.matrix_d("AvI", em, scale=True, correctA=True)
    # wM = (native/proj) * rowsum ; M = (1/(native/proj)) * (1/rowsum) * ApvI
    assert c.wM.tolist() == [1.25 * 5.0, 7.0]
    np.testing.assert_allclose(c.to_scipy().toarray(), [[0.8 / 1.25, 0.2 / 1.25, 0], [0, 3 / 7, 4 / 7]], rtol=1e-15)
    # apply: rows of M dotted with the field, Eigen column order
    y = s.apply(np.array([10.0, 20.0, 30.0]))
    assert y.tolist() == [0.8 * 10.0 + 0.2 * 20.0, (3.0 / 7.0) * 20.0 + (4.0 / 7.0) * 30.0]


def test_hand_computed_EvI_interp():
    rg = orc.Regridder(hand_grids())
    em = np.array([50.0, np.nan, 150.0, -20.0])     # elevation -20 is clamped to 0 (IceRegridder_L0.cpp:123)
    w = rg.matrix_d("EvI", em, scale=False, correctA=False)
    # cell0: elev 50 -> classes 0,1 weights .5,.5 ; cell2: 150 -> classes 1,2 ; cell3: 0 -> class 0 w=1 (w1==0 skipped)
    E = w.dims[0].tolist()
    assert E == [5 + 0 * 12, 5 + 1 * 12, 5 + 2 * 12, 9 + 1 * 12, 9 + 2 * 12, 9 + 0 * 12]
    d = w.to_scipy().toarray()
    assert d[0].tolist() == [2.0, 0.0, 0.0] and d[1].tolist() == [2.0, 0.5, 0.0] and d[2].tolist() == [0.0, 0.5, 0.0]
    assert d[3].tolist() == [0.0, 1.5, 0.0] and d[4].tolist() == [0.0, 1.5, 0.0] and d[5].tolist() == [0.0, 0.0, 4.0]
    with pytest.raises(orc.OracleError, match="out of bounds"):
        rg.matrix_d("EvI", np.array([50.0, 1.0, 250.0, 0.0]))     # above last hcdef (:84-85)
    g = hand_grids(); g["interp_style"] = 1
    n = orc.Regridder(g).matrix_d("EvI", em, scale=False)           # ELEV_CLASS_INTERP (:146-150)
    assert n.nnz == 4 and set(n.val.tolist()) == {4.0, 1.0, 3.0}


def ur_matrices(g, em):
    """Ur matrices in sparse index space, straight from the generator definitions."""
    iA, iI = g["ex_indices"][:, 0].astype(np.int64), g["ex_indices"][:, 1].astype(np.int64)
    area = g["ex_area"]
    nX = len(area)
    ok = ~np.isnan(em[iI])
    x = np.arange(nX)
    pos = ok & (area > 0)
    GvAp = sp.csr_matrix((area[pos], (x[pos], iA[pos])), shape=(nX, g["nA"]))
    GvI = sp.csr_matrix((area[ok], (x[ok], iI[ok])), shape=(nX, g["nI"]))
    hc = g["hcdefs"]; nhc = len(hc)
    elev = np.maximum(em[iI[ok]], 0.0)
    i1 = np.clip(np.searchsorted(hc, elev, side="left"), 1, None); i0 = i1 - 1
    ratio = (elev - hc[i0]) / (hc[i1] - hc[i0])
    rows = np.concatenate([x[ok], x[ok]])
    cols = np.concatenate([iA[ok] * g["hc_stride_A"] + i0 * g["hc_stride_HC"], iA[ok] * g["hc_stride_A"] + i1 * g["hc_stride_HC"]])
    vals = np.concatenate([area[ok] * (1 - ratio), area[ok] * ratio])
    nz = vals != 0
    GvEp = sp.csr_matrix((vals[nz], (rows[nz], cols[nz])), shape=(nX, g["nA"] * nhc))
    r = g["A_native_area"] / g["A_proj_area"]
    sA = np.zeros(g["nA"]); sA[g["A_to_sparse"]] = r
    sE = np.zeros(g["nA"] * nhc)
    for k in range(nhc):
        sE[g["A_to_sparse"] * g["hc_stride_A"] + k * g["hc_stride_HC"]] = r
    return GvAp, GvI, GvEp, sA, sE


def rowsum(M):
    return np.asarray(M.sum(axis=1)).ravel()


def colsum(M):
    return np.asarray(M.sum(axis=0)).ravel()


def inv(v):
    with np.errstate(divide="ignore"):
        return np.where(v != 0, 1.0 / v, 0.0)


@pytest.mark.parametrize("order", ["sorted", "shuffled"])
@pytest.mark.parametrize("scale", [True, False])
@pytest.mark.parametrize("correctA", [True, False])
def test_all_matrices_vs_scipy(order, scale, correctA):
    g = syn.make_grids("g50", order=order)
    em = syn.dome_elevmask(g)
    rg = orc.Regridder(g)
    GvAp, GvI, GvEp, sA, sE = ur_matrices(g, em)
    D = sp.diags
    ur = {"A": (GvAp, sA), "E": (GvEp, sE)}
    for name in ("AvI", "EvI", "IvA", "IvE", "EvA", "AvE", "AvX", "XvE"):
        w = rg.matrix_d(name, em, scale=scale, correctA=correctA)
        B, A = name[0], name[2]
        if A in "IX" :    # compute_AEvI
            G, s = ur[B]
            X = G.T if A == "X" else G.T @ D(inv(rowsum(GvI))) @ GvI
            wM = rowsum(X) * (s if correctA else 1.0)
            Mw = colsum(X)
            M = D(inv(wM)) @ X if scale else X
        elif B in "IX":   # compute_IvAE
            G, s = ur[A]
            X = G if B == "X" else GvI.T @ D(inv(rowsum(G))) @ G
            wM = rowsum(X)
            Mw = colsum(X) * (s if correctA else 1.0)
            M = X @ D(s) if correctA else X
            if scale:
                M = D(inv(wM)) @ M
        else:             # compute_EvA
            (GE, sE_), (GA, sA_) = ur[B], ur[A]
            X = GE.T @ D(inv(rowsum(GA))) @ GA
            wM = rowsum(X) * (sE_ if correctA else 1.0)
            Mw = colsum(X) * (sA_ if correctA else 1.0)
            M = X @ D(sA_) if correctA else X
            if scale:
                M = D(inv(wM)) @ M
        M = M.tocsr()
        got = sp.coo_matrix((w.val, (w.dims[0][w.row], w.dims[1][w.col])), shape=M.shape).tocsr()
        diff = abs(got - M)
        assert diff.max() <= 1e-13 * abs(M).max(), name
        assert got.nnz == M.nnz, name
        np.testing.assert_allclose(w.wM, wM[w.dims[0]], rtol=1e-13)
        np.testing.assert_allclose(w.Mw, Mw[w.dims[1]], rtol=1e-13)


def test_prepopulated_dims_append():
    # callers share one dimE across EvI / IvE / XvE and pass identity dimI (IceCoupler.cpp:366-371,462,467)
    g = syn.make_grids("g50")
    em = syn.dome_elevmask(g)
    rg = orc.Regridder(g)
    dimI = orc.SparseSet(g["nI"], init=np.arange(g["nI"]))
    dimE = orc.SparseSet(g["nA"] * 40)
    w1 = rg.matrix_d("EvI", em, dims=(dimE, dimI), scale=False, correctA=False)
    assert w1.ncol == g["nI"] and np.array_equal(w1.dims[1], np.arange(g["nI"]))
    nE1 = dimE.dense_extent
    w2 = rg.matrix_d("IvE", em, dims=(dimI, dimE), scale=True, correctA=False)
    assert dimE.dense_extent == nE1 and w2.nrow == g["nI"]
    fresh = rg.matrix_d("EvI", em, scale=False, correctA=False)
    np.testing.assert_array_equal(fresh.dims[0], w1.dims[0])
    # identity dimI: masked ice cells are empty columns with Mw == 0
    assert (w1.Mw == 0).sum() == np.isnan(em).sum()
    # same values, different column numbering
    a = sp.coo_matrix((fresh.val, (fresh.row, fresh.dims[1][fresh.col])), shape=(fresh.nrow, g["nI"])).tocsr()
    assert abs(a - w1.to_scipy().tocsr()).max() == 0.0
    # wM == 0 rows get fill in apply (mask_result, IceCoupler.cpp:186-201)
    y = w2.apply(np.ones(w2.ncol), fill=-7.0)
    assert np.all(y[np.isnan(em)] == -7.0) and np.all(np.abs(y[~np.isnan(em)] - 1) < 1e-12)


def test_zero_area_cells():
    # GvAp drops area<=0 (IceRegridder_L0.cpp:209); GvI keeps them (:186-187)
    g = syn.make_grids("g50", zero_area_every=7)
    em = syn.dome_elevmask(g)
    rg = orc.Regridder(g)
    for name in ("AvI", "IvA", "EvI", "IvE", "EvA", "AvE"):
        w = rg.matrix_d(name, em, scale=True, correctA=True)
        assert np.all(np.isfinite(w.val)), name
        assert not np.any(w.val == 0.0)


def test_apply_is_eigen_column_order():
    rng = np.random.default_rng(3)
    n, m, nnz = 7, 50, 200
    r, c = rng.integers(0, n, nnz), rng.integers(0, m, nnz)
    v = rng.standard_normal(nnz)
    w = orc.Weighted.from_coo(n, m, r, c, v, np.ones(n), np.ones(m))
    x = rng.standard_normal((3, m))
    y = w.apply(x)
    # duplicates were summed in emission order, then y[i] accumulates over ascending column
    for k in range(3):
        ref = np.zeros(n)
        dense = {}
        for a, b, val in zip(r, c, v):
            dense[(a, b)] = dense.get((a, b), 0.0) + val if (a, b) in dense else val
        for (a, b) in sorted(dense, key=lambda t: (t[1], t[0])):
            ref[a] = ref[a] + dense[(a, b)] * x[k, b]
        np.testing.assert_array_equal(y[k], ref)
    np.testing.assert_array_equal(w.apply_weight(1, x), [np.sum(x[k]) if False else _seq(x[k]) for k in range(3)])


def _seq(a):
    s = 0.0
    for t in a:
        s = s + 1.0 * t
    return s


def test_force_conservation_branch():
    # [INFERRED, parity unpinned] non-conservative matrix: factor (Mw.A)/(wM.B0) per variable
    w = orc.Weighted.from_coo(2, 2, [0, 1], [0, 1], [0.5, 0.5], [1.0, 1.0], [1.0, 1.0], conservative=False)
    x = np.array([[2.0, 4.0]])
    assert w.apply(x, force_conservation=False).tolist() == [[1.0, 2.0]]
    assert w.apply(x, force_conservation=True).tolist() == [[2.0, 4.0]]


def _xue_pair(config="g50", shift=35.0):
    """XvE (unscaled, correctA) for an elevation mask and for a slightly grown / raised one."""
    g = syn.make_grids(config)
    em0 = syn.dome_elevmask(g)
    em1 = syn.dome_elevmask(g, frac=0.80)
    em1 = np.where(np.isfinite(em1), em1 + shift, np.nan)
    rg = orc.Regridder(g)
    nX, nE = len(g["ex_area"]), g["nA"] * len(g["hcdefs"])
    out = []
    for em in (em1, em0):
        dimX = orc.SparseSet(nX, init=np.arange(nX))
        out.append(rg.matrix_d("XvE", em, dims=(dimX, None), scale=False, correctA=True))
    return g, out[0], out[1], nE


def test_e1ve0_against_scipy_and_identity_case():
    # compute_E1vE0c (slib/icebin/e1ve0.cpp:55-106): E1vE0c = diag(1/Mw1) * XuE1^T * (diag(1/wM0) XuE0 - diag(1/wM1) XuE1)
    import scipy.sparse as sp
    g, XuE1, XuE0, nE = _xue_pair()
    i, j, v = orc.e1ve0c([XuE1], [XuE0], nE)
    assert len(v) > 0 and np.all(np.diff(i * nE + j) > 0)                 # sorted by (iE1, iE0), consolidated
    nX = XuE1.sparse_extents[0]

    def sparse_mat(w):
        return sp.coo_matrix((w.val, (w.dims[0][w.row], w.dims[1][w.col])), shape=(nX, nE)).tocsr()
    M1, M0 = sparse_mat(XuE1), sparse_mat(XuE0)
    inv = lambda w, n: sp.diags(1.0 / np.bincount(w.dims[0], w.wM, n).clip(1e-300))   # noqa: E731
    D = inv(XuE0, nX) @ M0 - inv(XuE1, nX) @ M1
    Mw1 = np.bincount(XuE1.dims[1], XuE1.Mw, nE)
    ref = (sp.diags(np.where(Mw1 > 0, 1.0 / np.where(Mw1 > 0, Mw1, 1), 0.0)) @ (M1.T @ D)).tocoo()
    got = sp.coo_matrix((v, (i, j)), shape=(nE, nE))
    diff = abs(got.tocsr() - ref.tocsr())
    assert (diff.max() if diff.nnz else 0.0) <= 1e-12 * np.max(np.abs(ref.data))
    assert got.nnz >= ref.tocsr().nnz > 1000
    # where both masks carry ice, (XvE0 - XvE1) 1 = 0: rows of E1vE0c fed only by such exchange cells sum to 0
    both = np.isin(np.arange(nX), XuE0.dims[0][np.unique(XuE0.row)]) & np.isin(np.arange(nX), XuE1.dims[0][np.unique(XuE1.row)])
    touched_by_new = np.asarray((M1[~both].T @ np.ones((~both).sum()))).ravel() != 0
    rs = np.asarray(got.sum(axis=1)).ravel()
    assert np.max(np.abs(rs[~touched_by_new])) < 1e-12 and np.any(rs[touched_by_new] != 0)
    # E1 == E0: the correction vanishes identically (stored entries, all exactly zero)
    i0, j0, v0 = orc.e1ve0c([XuE0], [XuE0], nE)
    assert len(v0) > 0 and not np.any(v0)
