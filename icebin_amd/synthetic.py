"""Deterministic synthetic grids of the shapes BASELINE.json names.

The reference's exchange grid comes from a CGAL polygon overlap
(slib/icebin/gridgen/GridGen_Exchange.cpp:175-284) that cannot be regenerated
here, so the named configurations are produced as: a rectilinear ice grid I
(nx x ny cells of dx metres in the projected plane, SeaRISE sizes from
gridgen/searise_grid.cpp:112-122) overlapped with a rectilinear "projected GCM"
grid A of cell size Wx x Wy metres.  Exchange-cell areas are then exact products
of 1-D interval overlaps.  Cells are ordered by (iA, iI) ascending, the order of
ExchangeGrid's constructor (AbbrGrid.cpp:10-21, outer loop over A cells at
GridGen_Exchange.cpp:256).  This module generates INPUTS only; it contains no
regridding arithmetic.
"""
import numpy as np

SEED = 20240501

# name -> (Lx_km, Ly_km) SeaRISE domains (searise_grid.cpp:112-122)
_DOMAINS = {"greenland": (1500, 2800), "antarctica": (6000, 6000)}
# GCM grids (slib/icebin/modele/grids.cpp:14-17): name -> (im, jm, Wx_km, Wy_km)
# Wx/Wy: size of one GCM cell in the projected plane near 72N (2.5 deg lon ~ 86 km, 2 deg lat ~ 222 km).
_GCM = {"2x2.5": (144, 90, 86.0, 222.0), "1x1": (360, 180, 34.4, 111.0), "0.5x0.5": (720, 360, 17.2, 55.5)}

CONFIGS = {
    # BASELINE.json configs[0..4]
    "g20": dict(sheet="greenland", dx_km=20, gcm="2x2.5"),
    "g5": dict(sheet="greenland", dx_km=5, gcm="2x2.5"),
    "g1": dict(sheet="greenland", dx_km=1, gcm="1x1"),
    "g1h": dict(sheet="greenland", dx_km=1, gcm="0.5x0.5"),
    "a1h": dict(sheet="antarctica", dx_km=1, gcm="0.5x0.5"),
    # small shapes for fast tests
    "tiny": dict(sheet="greenland", dx_km=100, gcm="2x2.5"),
    "g50": dict(sheet="greenland", dx_km=50, gcm="2x2.5"),
}


def _segments(n, d, edges_a):
    """Overlay ice-cell edges k*d (k=0..n) with GCM edges; return per segment
    (ice cell, gcm cell, length)."""
    e_i = np.arange(n + 1, dtype=np.float64) * d
    lo, hi = e_i[0], e_i[-1]
    inner = edges_a[(edges_a > lo) & (edges_a < hi)]
    cuts = np.union1d(e_i, inner)
    left, right = cuts[:-1], cuts[1:]
    mid = 0.5 * (left + right)
    ci = np.minimum((mid // d).astype(np.int64), n - 1)
    ca = np.searchsorted(edges_a, mid, side="right") - 1
    return ci, ca, right - left


def make_grids(config="g5", nhc=40, x_fastest=False, ratio_amp=0.03, order="sorted", zero_area_every=0):
    """Build the arrays a GCMRegridder_Standard + IceRegridder_L0 hold.

    Returns a dict: nI, nA, nx, ny, ex_indices int32[nX,2] (iA, iI), ex_area f64[nX],
    A_to_sparse int64[nA_dense], A_native_area, A_proj_area, hcdefs, hc strides.
    order: "sorted" (by iA then iI), "shuffled" (seeded permutation; exercises
    the general path).  zero_area_every=k>0 zeroes every k-th overlap (edge case
    IceRegridder_L0.cpp:209).
    """
    cfg = CONFIGS[config] if isinstance(config, str) else dict(config)
    Lx, Ly = _DOMAINS[cfg["sheet"]]
    dx_km = cfg["dx_km"]
    nx, ny = (Lx + dx_km) // dx_km, (Ly + dx_km) // dx_km      # searise_grid.cpp:112-122
    im, jm, Wx_km, Wy_km = _GCM[cfg["gcm"]]
    d = dx_km * 1000.0
    Wx, Wy = Wx_km * 1000.0, Wy_km * 1000.0
    # GCM cell edges, offset so they do not line up with ice-cell edges
    x_off, y_off = -0.37 * Wx, -0.61 * Wy
    nAx = int(np.ceil((nx * d - x_off) / Wx))
    nAy = int(np.ceil((ny * d - y_off) / Wy))
    ex = x_off + np.arange(nAx + 1) * Wx
    ey = y_off + np.arange(nAy + 1) * Wy
    i0, j0 = (im - nAx) // 3, (jm - nAy) // 2          # where the block sits in the global im x jm grid
    assert i0 >= 0 and j0 >= 0, "GCM grid too small for this ice domain"

    six, sax, slx = _segments(nx, d, ex)
    siy, say, sly = _segments(ny, d, ey)
    # exchange cells = product of x- and y-segments
    IX, IY = np.meshgrid(six, siy, indexing="ij")
    AX, AY = np.meshgrid(sax, say, indexing="ij")
    LX, LY = np.meshgrid(slx, sly, indexing="ij")
    iI = (IY * nx + IX) if x_fastest else (IX * ny + IY)    # PISM-style: y fastest (elev_mask.cdl dims (x,y))
    iA = (j0 + AY) * im + (i0 + AX)                         # ModelE: i fastest
    area = LX * LY
    iI, iA, area = iI.ravel(), iA.ravel(), area.ravel()
    if order == "sorted":
        perm = np.lexsort((iI, iA))
    else:
        perm = np.random.default_rng(SEED + 7).permutation(len(iI))
    iI, iA, area = iI[perm], iA[perm], area[perm]
    if zero_area_every:
        area = area.copy()
        area[::zero_area_every] = 0.0

    # realised A cells: the whole nAx x nAy block, ascending sparse index (Grid cells.sorted())
    KY, KX = np.meshgrid(np.arange(nAy), np.arange(nAx), indexing="ij")
    A_to_sparse = ((j0 + KY) * im + (i0 + KX)).ravel().astype(np.int64)
    proj_area = np.full(A_to_sparse.shape, Wx * Wy)
    ratio = 1.0 + ratio_amp * np.cos(np.pi * (KY.ravel() + 0.5) / nAy)      # in [1-amp, 1+amp]
    native_area = proj_area * ratio
    hcdefs = np.arange(nhc, dtype=np.float64) * 100.0 - 50.0   # write_icebin_in_base.py:44
    # centroids of the ice cells by sparse index (AbbrGrid::centroid_xy; read only by the smoother)
    CX, CY = np.meshgrid((np.arange(nx) + 0.5) * d, (np.arange(ny) + 0.5) * d, indexing="ij")
    if x_fastest:
        CX, CY = CX.T, CY.T
    centroid = np.stack([CX.ravel(), CY.ravel()], axis=1)

    return dict(
        config=config, nx=int(nx), ny=int(ny), dx=d, x_fastest=x_fastest,
        nI=int(nx * ny), nA=int(im * jm), im=im, jm=jm,
        ex_indices=np.stack([iA, iI], axis=1).astype(np.int32), ex_area=area,
        A_to_sparse=A_to_sparse, A_native_area=native_area, A_proj_area=proj_area,
        hcdefs=hcdefs, hc_stride_A=1, hc_stride_HC=int(im * jm),   # icebin_cython.cpp:69
        interp_style=0, I_centroid_xy=centroid,
        # the projected GCM block (for the exchange-grid generator's tests): cell (kx, ky) is
        # [A_xedges[kx], A_xedges[kx+1]] x [A_yedges[ky], A_yedges[ky+1]], sparse index (j0+ky)*im + (i0+kx)
        A_xedges=ex, A_yedges=ey, A_i0=int(i0), A_j0=int(j0),
    )


def dome_elevmask(grids, all_unmasked=False, frac=0.76):
    """elev = 3200*max(0, 1-r^2) m; NaN where elev <= 0 (about 45 % unmasked)."""
    nx, ny = grids["nx"], grids["ny"]
    cx = (np.arange(nx) + 0.5) / nx * 2.0 - 1.0
    cy = (np.arange(ny) + 0.5) / ny * 2.0 - 1.0
    X, Y = np.meshgrid(cx, cy, indexing="ij")
    r2 = (X / frac) ** 2 + (Y / frac) ** 2
    elev = 3200.0 * np.maximum(0.0, 1.0 - r2)
    if all_unmasked:
        em = elev
    else:
        em = np.where(elev > 0.0, elev, np.nan)
    em = em.T if grids["x_fastest"] else em
    return np.ascontiguousarray(em).reshape(-1)


def fields(nf, n, seed=SEED):
    """x_f[i] = sin(2*pi*(f+1)*i/n) + 0.01*U(0,1), shape [nf, n] field-major."""
    rng = np.random.default_rng(seed)
    i = np.arange(n, dtype=np.float64)
    out = np.empty((nf, n), np.float64)
    for f in range(nf):
        out[f] = np.sin(2.0 * np.pi * (f + 1) * i / n) + 0.01 * rng.random(n)
    return out
