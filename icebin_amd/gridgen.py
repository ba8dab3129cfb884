"""Exchange-grid generation (slib/icebin/gridgen/GridGen_Exchange.cpp:175-284) over the C-ABI: a rectilinear
ice grid in the projected plane under convex GCM-cell polygons.  Inputs only -- the overlap arithmetic runs
in gridgen.hip."""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import check, lib, ptr


def make_exchange_grid(xedges, yedges, polys, iA, x_fastest=False):
    """xedges [nx+1], yedges [ny+1]: ice-cell edges (ascending); polys: list of [nv, 2] vertex arrays
    (counter-clockwise, projected XY) of the realised GCM cells; iA: their sparse indices (ascending).
    Returns dict(indices=int32[nX, 2] (iA, iI), overlaps=f64[nX]) sorted by (iA, iI)."""
    xe, ye = np.ascontiguousarray(xedges, np.float64), np.ascontiguousarray(yedges, np.float64)
    iA = np.ascontiguousarray(iA, np.int64)
    assert len(polys) == len(iA)
    polyptr = np.zeros(len(polys) + 1, np.int32)
    polyptr[1:] = np.cumsum([len(p) for p in polys])
    v = np.concatenate([np.asarray(p, np.float64).reshape(-1, 2) for p in polys]) if len(polys) else np.zeros((0, 2))
    vx, vy = np.ascontiguousarray(v[:, 0]), np.ascontiguousarray(v[:, 1])
    d = _capi.ExgridDesc(nx=len(xe) - 1, ny=len(ye) - 1, xedges=ptr(xe).value, yedges=ptr(ye).value, x_fastest=int(bool(x_fastest)),
                         npoly=len(polys), polyptr=ptr(polyptr).value, vx=ptr(vx).value if len(vx) else None,
                         vy=ptr(vy).value if len(vy) else None, iA=ptr(iA).value if len(iA) else None)
    h = C.c_void_p()
    check(lib().ibh_exgrid_generate(C.byref(d), C.byref(h)))
    try:
        n = C.c_int64()
        check(lib().ibh_exgrid_size(h, C.byref(n)))
        idx, ov = np.empty((n.value, 2), np.int32), np.empty(n.value, np.float64)
        check(lib().ibh_exgrid_get(h, ptr(idx), ptr(ov)))
    finally:
        lib().ibh_exgrid_destroy(h)
    return dict(indices=idx, overlaps=ov)
