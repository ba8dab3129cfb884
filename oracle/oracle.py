"""ctypes front-end of the CPU ORACLE (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (icebin_amd/) never imports
this module.  See icebin_oracle.h for scope and parity status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "icebin_oracle.c")
    hdr = os.path.join(_HERE, "icebin_oracle.h")
    stale = (not os.path.exists(so)) or any(
        os.path.getmtime(f) > os.path.getmtime(so) for f in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return so


class _Regridder(C.Structure):
    _fields_ = [
        ("nX", C.c_long), ("ex_indices", C.c_void_p), ("ex_area", C.c_void_p),
        ("nI", C.c_long),
        ("nA", C.c_long), ("nA_dense", C.c_int), ("A_to_sparse", C.c_void_p),
        ("A_native_area", C.c_void_p), ("A_proj_area", C.c_void_p),
        ("nhc", C.c_int), ("hcdefs", C.c_void_p),
        ("hc_stride_A", C.c_long), ("hc_stride_HC", C.c_long),
        ("interp_style", C.c_int),
        ("I_centroid_xy", C.c_void_p),
    ]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_sset_new.restype = C.c_void_p
        L.orc_sset_new.argtypes = [C.c_long]
        L.orc_sset_free.argtypes = [C.c_void_p]
        L.orc_sset_add_dense.argtypes = [C.c_void_p, C.c_long]
        L.orc_sset_add_dense.restype = C.c_int
        L.orc_sset_dense_extent.argtypes = [C.c_void_p]
        L.orc_sset_sparse_extent.argtypes = [C.c_void_p]
        L.orc_sset_sparse_extent.restype = C.c_long
        L.orc_sset_to_sparse_ptr.argtypes = [C.c_void_p]
        L.orc_sset_to_sparse_ptr.restype = C.POINTER(C.c_long)
        L.orc_matrix_d.argtypes = [C.POINTER(_Regridder), C.c_void_p, C.c_char_p, C.c_void_p,
                                   C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                   C.POINTER(C.c_void_p), C.c_char_p, C.c_int]
        L.orc_weighted_free.argtypes = [C.c_void_p]
        for f in ("nrow", "ncol", "conservative", "scaled"):
            getattr(L, "orc_weighted_" + f).argtypes = [C.c_void_p]
        L.orc_weighted_nnz.argtypes = [C.c_void_p]
        L.orc_weighted_nnz.restype = C.c_long
        L.orc_weighted_dim.argtypes = [C.c_void_p, C.c_int]
        L.orc_weighted_dim.restype = C.c_void_p
        L.orc_weighted_wM.argtypes = [C.c_void_p]
        L.orc_weighted_wM.restype = C.POINTER(C.c_double)
        L.orc_weighted_Mw.argtypes = [C.c_void_p]
        L.orc_weighted_Mw.restype = C.POINTER(C.c_double)
        L.orc_weighted_coo.argtypes = [C.c_void_p] * 4
        L.orc_weighted_from_coo.argtypes = [C.c_int, C.c_int, C.c_long, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_weighted_from_coo.restype = C.c_void_p
        L.orc_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p]
        L.orc_apply_transformed.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                            C.c_double, C.c_void_p]
        L.orc_apply_weight.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_time_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_time_apply.restype = C.c_double
        L.orc_linterp_1d_b.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        L.orc_linterp_1d.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        L.orc_nearest_1d.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.orc_elevmask_pism.argtypes = [C.c_long] + [C.c_void_p] * 5
        L.orc_coo_matvec.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_long, C.c_long,
                                     C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_e1ve0c.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.POINTER(C.POINTER(C.c_long)),
                                 C.POINTER(C.POINTER(C.c_long)), C.POINTER(C.POINTER(C.c_double))]
        L.orc_e1ve0c.restype = C.c_long
        L.orc_free.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleError(RuntimeError):
    pass


class SparseSet:
    """spsparse::SparseSet<long,int> restatement (first-seen dense numbering)."""

    def __init__(self, sparse_extent=-1, init=None):
        self._h = lib().orc_sset_new(int(sparse_extent))
        if init is not None:
            for s in np.asarray(init, dtype=np.int64):
                lib().orc_sset_add_dense(self._h, int(s))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _LIB is not None:          # (at interpreter shutdown the module globals may already be gone: leak, do not raise)
            _LIB.orc_sset_free(h)

    @property
    def dense_extent(self):
        return lib().orc_sset_dense_extent(self._h)

    @property
    def sparse_extent(self):
        return lib().orc_sset_sparse_extent(self._h)

    def add_dense(self, s):
        return lib().orc_sset_add_dense(self._h, int(s))

    def to_sparse(self):
        n = self.dense_extent
        p = lib().orc_sset_to_sparse_ptr(self._h)
        return np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n].copy()


def _sset_view(h):
    n = lib().orc_sset_dense_extent(h)
    p = lib().orc_sset_to_sparse_ptr(h)
    arr = np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n].copy()
    return arr, lib().orc_sset_sparse_extent(h)


class Weighted:
    """Oracle-side linear::Weighted_Eigen result."""

    def __init__(self, h, keep=()):
        self._h = h
        self._keep = keep
        L = lib()
        self.nrow, self.ncol, self.nnz = L.orc_weighted_nrow(h), L.orc_weighted_ncol(h), L.orc_weighted_nnz(h)
        self.conservative = bool(L.orc_weighted_conservative(h))
        self.scaled = bool(L.orc_weighted_scaled(h))
        self.wM = np.ctypeslib.as_array(L.orc_weighted_wM(h), shape=(max(self.nrow, 1),))[:self.nrow].copy()
        self.Mw = np.ctypeslib.as_array(L.orc_weighted_Mw(h), shape=(max(self.ncol, 1),))[:self.ncol].copy()
        self.dims = []
        self.sparse_extents = []
        for k in (0, 1):
            a, ext = _sset_view(L.orc_weighted_dim(h, k))
            self.dims.append(a)
            self.sparse_extents.append(ext)
        self.row = np.empty(self.nnz, np.int32)
        self.col = np.empty(self.nnz, np.int32)
        self.val = np.empty(self.nnz, np.float64)
        L.orc_weighted_coo(h, _p(self.row), _p(self.col), _p(self.val))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _LIB is not None:
            _LIB.orc_weighted_free(h)

    @classmethod
    def from_coo(cls, nrow, ncol, row, col, val, wM, Mw, conservative=True, scaled=True):
        row = np.ascontiguousarray(row, np.int32)
        col = np.ascontiguousarray(col, np.int32)
        val = np.ascontiguousarray(val, np.float64)
        wM = np.ascontiguousarray(wM, np.float64)
        Mw = np.ascontiguousarray(Mw, np.float64)
        h = lib().orc_weighted_from_coo(nrow, ncol, len(val), _p(row), _p(col), _p(val), _p(wM), _p(Mw),
                                        int(conservative), int(scaled))
        return cls(h)

    def apply(self, A_b, fill=np.nan, force_conservation=False):
        A = np.ascontiguousarray(np.atleast_2d(A_b), np.float64)
        assert A.shape[1] == self.ncol, (A.shape, self.ncol)
        B = np.empty((A.shape[0], self.nrow), np.float64)
        lib().orc_apply(self._h, _p(A), A.shape[0], float(fill), int(force_conservation), _p(B))
        return B if np.ndim(A_b) == 2 else B[0]

    def apply_transformed(self, V_b, T, b, fill=np.nan):
        V = np.ascontiguousarray(V_b, np.float64)
        T = np.ascontiguousarray(T, np.float64)
        b = np.ascontiguousarray(b, np.float64)
        assert V.shape == (T.shape[0], self.ncol) and b.shape == (T.shape[1],)
        B = np.empty((T.shape[1], self.nrow), np.float64)
        lib().orc_apply_transformed(self._h, _p(V), V.shape[0], _p(T), _p(b), T.shape[1], float(fill), _p(B))
        return B

    def apply_weight(self, dim, A_b):
        A = np.ascontiguousarray(np.atleast_2d(A_b), np.float64)
        out = np.empty(A.shape[0], np.float64)
        lib().orc_apply_weight(self._h, dim, _p(A), A.shape[0], _p(out))
        return out

    def time_apply(self, A_b, reps=1, threads=1):
        A = np.ascontiguousarray(np.atleast_2d(A_b), np.float64)
        B = np.empty((A.shape[0], self.nrow), np.float64)
        return lib().orc_time_apply(self._h, _p(A), A.shape[0], _p(B), reps, threads)

    def to_scipy(self):
        import scipy.sparse
        return scipy.sparse.coo_matrix((self.val, (self.row, self.col)), shape=(self.nrow, self.ncol))


class Regridder:
    """Holds the arrays of a GCMRegridder_Standard + one IceRegridder_L0."""

    def __init__(self, grids):
        g = grids
        self._arrs = dict(
            ex_indices=np.ascontiguousarray(g["ex_indices"], np.int32).reshape(-1),
            ex_area=np.ascontiguousarray(g["ex_area"], np.float64),
            A_to_sparse=np.ascontiguousarray(g["A_to_sparse"], np.int64),
            A_native_area=np.ascontiguousarray(g["A_native_area"], np.float64),
            A_proj_area=np.ascontiguousarray(g["A_proj_area"], np.float64),
            hcdefs=np.ascontiguousarray(g["hcdefs"], np.float64),
        )
        if g.get("I_centroid_xy") is not None:
            self._arrs["I_centroid_xy"] = np.ascontiguousarray(g["I_centroid_xy"], np.float64).reshape(-1)
        a = self._arrs
        self.nI, self.nA, self.nhc = int(g["nI"]), int(g["nA"]), len(a["hcdefs"])
        self.c = _Regridder(
            nX=len(a["ex_area"]), ex_indices=_p(a["ex_indices"]).value, ex_area=_p(a["ex_area"]).value,
            nI=self.nI, nA=self.nA, nA_dense=len(a["A_to_sparse"]),
            A_to_sparse=_p(a["A_to_sparse"]).value, A_native_area=_p(a["A_native_area"]).value,
            A_proj_area=_p(a["A_proj_area"]).value, nhc=self.nhc, hcdefs=_p(a["hcdefs"]).value,
            hc_stride_A=int(g.get("hc_stride_A", 1)), hc_stride_HC=int(g.get("hc_stride_HC", g["nA"])),
            interp_style=int(g.get("interp_style", 0)),
            I_centroid_xy=_p(a["I_centroid_xy"]).value if "I_centroid_xy" in a else None)

    def matrix_d(self, spec, elevmaskI, dims=(None, None), scale=True, correctA=False, sigma=(0., 0., 0.)):
        em = np.ascontiguousarray(elevmaskI, np.float64).reshape(-1)
        assert em.shape[0] == self.nI
        sig = np.asarray(sigma, np.float64)
        out = C.c_void_p()
        err = C.create_string_buffer(512)
        d0 = dims[0]._h if dims[0] is not None else None
        d1 = dims[1]._h if dims[1] is not None else None
        rc = lib().orc_matrix_d(C.byref(self.c), _p(em), spec.encode(), d0, d1, int(scale), int(correctA),
                                _p(sig), C.byref(out), err, 512)
        if rc != 0:
            raise OracleError(err.value.decode())
        return Weighted(out.value, keep=(self, dims))


def linterp_1d_b(xpoints, xx):
    xp = np.ascontiguousarray(xpoints, np.float64)
    idx = np.zeros(2, np.int64)
    w = np.zeros(2, np.float64)
    rc = lib().orc_linterp_1d_b(_p(xp), len(xp), float(xx), _p(idx), _p(w))
    if rc != 0:
        raise OracleError("Elevation %g out of bounds (%g, %g)" % (xx, xp[0], xp[-1]))
    return idx, w


def linterp_1d(xpoints, xx):
    xp = np.ascontiguousarray(xpoints, np.float64)
    idx = np.zeros(2, np.int32)
    w = np.zeros(2, np.float64)
    lib().orc_linterp_1d(_p(xp), len(xp), float(xx), _p(idx), _p(w))
    return idx, w


def nearest_1d(xpoints, xx):
    xp = np.ascontiguousarray(xpoints, np.float64)
    return lib().orc_nearest_1d(_p(xp), len(xp), float(xx))


def elevmask_pism(mask, thk, topg):
    mask = np.ascontiguousarray(mask, np.int8).reshape(-1)
    thk = np.ascontiguousarray(thk, np.float64).reshape(-1)
    topg = np.ascontiguousarray(topg, np.float64).reshape(-1)
    land = np.empty_like(thk)
    ice = np.empty_like(thk)
    lib().orc_elevmask_pism(len(mask), _p(mask), _p(thk), _p(topg), _p(land), _p(ice))
    return land, ice


def coo_matvec(M, xx, fill=np.nan, ignore_nan=False):
    """icebin.coo_multiply(M, x, fill) as used by tests/test_conserv/test_conserv.py:139."""
    M = M.tocoo()
    yy = np.full(M.shape[0], fill, np.float64)
    xx = np.ascontiguousarray(xx, np.float64)
    row = np.ascontiguousarray(M.row, np.int32)
    col = np.ascontiguousarray(M.col, np.int32)
    dat = np.ascontiguousarray(M.data, np.float64)
    lib().orc_coo_matvec(_p(yy), _p(xx), int(ignore_nan), M.shape[0], M.shape[1], len(dat), _p(row), _p(col), _p(dat))
    return yy


def e1ve0c(XuE1s, XuE0s, nE):
    """compute_E1vE0c (slib/icebin/e1ve0.cpp:55-106): lists of oracle Weighted (one per ice sheet) -> (iE1, iE0, val)
    sorted by (iE1, iE0), sparse E indices."""
    n = len(XuE1s)
    a1 = (C.c_void_p * n)(*[w._h for w in XuE1s])
    a0 = (C.c_void_p * n)(*[w._h for w in XuE0s])
    pi, pj, pv = C.POINTER(C.c_long)(), C.POINTER(C.c_long)(), C.POINTER(C.c_double)()
    cnt = lib().orc_e1ve0c(n, a1, a0, int(nE), C.byref(pi), C.byref(pj), C.byref(pv))
    i = np.ctypeslib.as_array(pi, shape=(max(cnt, 1),))[:cnt].copy()
    j = np.ctypeslib.as_array(pj, shape=(max(cnt, 1),))[:cnt].copy()
    v = np.ctypeslib.as_array(pv, shape=(max(cnt, 1),))[:cnt].copy()
    for q in (pi, pj, pv):
        lib().orc_free(q)
    return i, j, v
