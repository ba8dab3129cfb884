# cython: language_level=3
# distutils: language = c++
"""_icebin: the Cython module of the reference (pylib/_icebin.pyx:50-175) over the MI355X build.

Same classes, method names, keyword names and defaults -- `import icebin; icebin.GCMRegridder(...)`,
`.regrid_matrices(sheet, elevmaskI, scale=True, correctA=True, sigma=(0,0,0), conserve=True)`,
`RegridMatrices.matrix(spec)` -> `linear_Weighted` with apply_M / apply_weight / apply_wM / apply_Mw / to_coo /
get_weights / shape (ibmisc.linear_Weighted, sphinx/source/matrix_formats.rst:139-195) -- bound to the
header-only C++ host mirror (icebin_amd/host/icebin_hip.hpp), which calls the C-ABI of libicebin_hip.so.
File arguments are IceBin files in the NetCDF-classic container of icebin_amd/ncio.py.
"""
from cython.operator cimport dereference as deref
from libcpp cimport bool
from libcpp.string cimport string
from libcpp.vector cimport vector
from libcpp.utility cimport move
cimport cicebin
import numpy as np
import scipy.sparse

_INTERP = {"Z_INTERP": 0, "ELEV_CLASS_INTERP": 1}      # IceRegridder.hpp:36-39


cdef vector[double] _vec_d(a):
    cdef vector[double] v
    cdef double[::1] m = np.ascontiguousarray(a, np.float64).reshape(-1)
    cdef Py_ssize_t i
    v.reserve(m.shape[0])
    for i in range(m.shape[0]):
        v.push_back(m[i])
    return v


cdef class linear_Weighted:
    """ibmisc.linear_Weighted (matrix_formats.rst:139-195): M, wM, Mw and dims of one regrid matrix."""
    cdef cicebin.Weighted *cself
    cdef object _dims

    def __cinit__(self):
        self.cself = NULL
        self._dims = [None, None]

    def __dealloc__(self):
        if self.cself != NULL:
            del self.cself

    @property
    def conservative(self):
        return self.cself.conservative

    @property
    def scaled(self):
        return self.cself.scaled

    @property
    def shape(self):
        """Sparse shape of the matrix."""
        cdef cicebin.array2l sh = self.cself.shape()
        return (sh[0], sh[1])

    @property
    def shape_d(self):
        cdef cicebin.array2i sh = self.cself.shape_d()
        return (sh[0], sh[1])

    def dim(self, int k):
        """dims[k]: dense -> sparse table (int64)."""
        if self._dims[k] is None:
            self._dims[k] = np.asarray(self.cself.dim_to_sparse(k), dtype=np.int64)
        return self._dims[k]

    @property
    def wM(self):
        return np.asarray(self.cself.wM(), dtype=np.float64)

    @property
    def Mw(self):
        return np.asarray(self.cself.Mw(), dtype=np.float64)

    def get_weights(self, int idim):
        """Weight vector in the sparse space (0 = B/output, 1 = A/input); zero where unset."""
        out = np.zeros(self.shape[idim])
        out[self.dim(idim)] = self.wM if idim == 0 else self.Mw
        return out

    def to_coo(self):
        """scipy.sparse.coo_matrix in the ORIGINAL sparse index spaces (matrix_formats.rst:188-192)."""
        cdef vector[int] row, col
        cdef vector[double] val
        self.cself.M_coo(row, col, val)
        r, c = np.asarray(row, dtype=np.int64), np.asarray(col, dtype=np.int64)
        return scipy.sparse.coo_matrix((np.asarray(val, dtype=np.float64), (self.dim(0)[r], self.dim(1)[c])), shape=self.shape)

    def apply_M(self, A_s, double fill=np.nan, bool force_conservation=True):
        """Applies the regrid matrix to A_s (vectors in the ORIGINAL sparse space of the input grid): a 1-D
        vector or a 2-D array of row vectors; un-set output cells get `fill` (matrix_formats.rst:156-165)."""
        A = np.asarray(A_s, np.float64)
        one = A.ndim == 1
        A2 = A.reshape(1, -1) if one else A.reshape(A.shape[0], -1)
        if A2.shape[1] != self.shape[1]:
            raise ValueError("apply_M: input has %d cells, the input grid has %d" % (A2.shape[1], self.shape[1]))
        cdef double[:, ::1] Ad = np.ascontiguousarray(A2[:, self.dim(1)])
        cdef vector[double] B
        nrow_d = self.shape_d[0]
        if Ad.shape[1] == 0 or nrow_d == 0:
            B_d = np.zeros((A2.shape[0], nrow_d))
        else:
            B = self.cself.apply(cicebin.ArrayViewCD(&Ad[0, 0], Ad.shape[0], Ad.shape[1]), fill, force_conservation)
            B_d = np.asarray(B, dtype=np.float64).reshape(A2.shape[0], nrow_d)
        B_s = np.full((A2.shape[0], self.shape[0]), fill, np.float64)
        B_s[:, self.dim(0)] = B_d
        return B_s[0] if one else B_s

    def apply_weight(self, int dim, A_s):
        """Dot product of a weight vector with A_s (dim 0: wM / output grid, 1: Mw / input grid)."""
        A = np.asarray(A_s, np.float64)
        one = A.ndim == 1
        A2 = A.reshape(1, -1) if one else A.reshape(A.shape[0], -1)
        if A2.shape[1] != self.shape[dim]:
            raise ValueError("apply_weight: input has %d cells, expected %d" % (A2.shape[1], self.shape[dim]))
        w = self.wM if dim == 0 else self.Mw
        d = self.dim(dim)
        out = np.array([np.dot(w[w != 0], row[d][w != 0]) for row in A2])
        return out[0] if one else out

    def apply_wM(self, A_s):
        return self.apply_weight(0, A_s)

    def apply_Mw(self, A_s):
        return self.apply_weight(1, A_s)


cdef class RegridMatrices:
    """cdef class RegridMatrices (_icebin.pyx:50-75)."""
    cdef cicebin.RegridMatrices *cself
    cdef object _keep

    def __cinit__(self):
        self.cself = NULL

    def __dealloc__(self):
        if self.cself != NULL:
            del self.cself

    def matrix(self, str spec_name):
        """Compute a regrid matrix.
        spec_name: 'EvI', 'AvI', 'IvA', 'IvE', 'EvA', 'AvE' (also 'AvX', 'XvA', 'EvX', 'XvE')
        returns: linear_Weighted"""
        cdef cicebin.Weighted *lw = cicebin.RegridMatrices_matrix(self.cself, spec_name.encode())
        cdef linear_Weighted ret = linear_Weighted()
        ret.cself = lw
        return ret


cdef class GCMRegridder:
    """cdef class GCMRegridder (_icebin.pyx:77-175)."""
    cdef cicebin.GCMRegridder_Standard *cself
    cdef object _gridA, _sheets

    def __cinit__(self):
        self.cself = NULL

    def __dealloc__(self):
        if self.cself != NULL:
            del self.cself

    def __init__(self, *args):
        from icebin_amd import ncio
        self._sheets = {}
        if len(args) == 4:                    # GCMRegridder(gridA_fname, gridA_vname, hcdefs, correctA), _icebin.pyx:89-100
            gridA_fname, gridA_vname, hcdefs, correctA = args
            a = ncio.get_abbr_grid(ncio.Dataset.read(gridA_fname), gridA_vname)
            self._init(a["sparse_extent"], a["to_sparse"], a["native_area"], hcdefs, correctA, None)
        elif len(args) == 3:                  # in-memory form: (gridA dict, hcdefs, correctA)
            gridA, hcdefs, correctA = args
            self._init(gridA["nA"], gridA["to_sparse"], gridA["native_area"], hcdefs, correctA, gridA.get("hc_strides"))
        elif len(args) == 1:                  # GCMRegridder(regridder_fname): load an IceBin input file, _icebin.pyx:81-88
            self._read(args[0], "m")
        elif len(args) == 0:
            pass
        else:
            raise ValueError('Invalid arguments: {}'.format(args))

    cdef _init(self, nA, to_sparse, native_area, hcdefs, correctA, hc_strides):
        cdef cicebin.AbbrGrid g
        cdef long[::1] ts = np.ascontiguousarray(to_sparse, np.int64).astype(np.int_)
        cdef Py_ssize_t i
        g.sparse_extent = nA
        for i in range(ts.shape[0]):
            g.dim_to_sparse.push_back(ts[i])
        g.native_area = _vec_d(native_area)
        cdef cicebin.array2l st
        # Indexing({"A","HC"}, {0,0}, {nA,nhc}, {1,0}): HC has the largest stride (icebin_cython.cpp:69)
        st[0] = 1 if hc_strides is None else hc_strides[0]
        st[1] = nA if hc_strides is None else hc_strides[1]
        if self.cself != NULL:
            del self.cself
        self.cself = new cicebin.GCMRegridder_Standard()
        self.cself.init(move(g), move(_vec_d(hcdefs)), st, True if correctA else False)
        self._gridA = dict(nA=int(nA), to_sparse=np.asarray(to_sparse, np.int64), native_area=np.asarray(native_area, np.float64))

    def _read(self, fname, vname):
        from icebin_amd import ncio
        ds = ncio.Dataset.read(fname)
        info = ds.variables[vname + ".info"]
        a = ncio.get_abbr_grid(ds, vname + ".agridA")
        _, extent, indices = ncio.get_indexing(ds, vname + ".indexingHC")
        st = ncio.indexing_strides(extent, indices)
        self._init(a["sparse_extent"], a["to_sparse"], a["native_area"], np.asarray(ds.variables[vname + ".hcdefs"].data),
                   int(info.correctA) != 0, (int(st[0]), int(st[1])))
        for name in ncio._strlist(info.sheets):
            v = vname + "." + name
            gI = ncio.get_abbr_grid(ds, v + ".agridI")
            nI = int(gI["sparse_extent"])
            cen = None
            if np.any(gI["centroid_xy"] != 0):
                cen = np.zeros((nI, 2))
                cen[gI["to_sparse"]] = gI["centroid_xy"]
            idx, area = ncio.get_exchange_grid(ds, v + ".aexgrid")
            self._add_sheet_arrays(name, nI, idx, area, np.asarray(ds.variables[v + ".gridA_proj_area"].data),
                                   str(ds.variables[v + ".info"].interp_style), cen)

    @property
    def nA(self):
        return self.cself.nA()

    @property
    def nE(self):
        return self.cself.nE()

    @property
    def nhc(self):
        return self.cself.nhc()

    def wA(self, sheet_name, snative, fill=0.):
        """Returns weights (as a vector) of overall grid."""
        if snative == 'native':
            native = True
        elif snative == 'proj':
            native = False
        else:
            raise ValueError("Invalid argument: snative must be 'native' or 'proj'")
        return np.asarray(self.cself.wA(sheet_name.encode(), native, fill), dtype=np.float64)

    cdef _add_sheet_arrays(self, name, nI, indices, overlaps, proj_area, interp_style, centroid_xy):
        if interp_style not in _INTERP:
            raise ValueError("unknown interp_style %r" % (interp_style,))
        cdef cicebin.ExchangeGrid ex
        cdef int[::1] ix = np.ascontiguousarray(indices, np.int32).reshape(-1)
        cdef Py_ssize_t i
        ex.indices.reserve(ix.shape[0])
        for i in range(ix.shape[0]):
            ex.indices.push_back(ix[i])
        ex.overlaps = _vec_d(overlaps)
        cdef vector[double] proj, cen
        if proj_area is not None:
            proj = _vec_d(proj_area)
        if centroid_xy is not None:
            cen = _vec_d(centroid_xy)
        self.cself.add_sheet(name.encode(), nI, ex, proj, _INTERP[interp_style], cen)
        self._sheets[name] = int(nI)

    def add_sheet(self, name, *args):
        """add_sheet(name, gridI_fname, gridI_vname, exgrid_fname, exgrid_vname, interp_style)   (_icebin.pyx:152-162;
        the grid files hold an AbbrGrid record of I and an ExchangeGrid record), or the in-memory form
        add_sheet(name, gridI, exgrid, interp_style='Z_INTERP', gridA_proj_area=None) with gridI = dict(nI=[, centroid_xy=])
        and exgrid = dict(indices=int32[nX,2] (iA,iI), overlaps=f64[nX])."""
        from icebin_amd import ncio
        if len(args) == 5 and isinstance(args[0], str):
            gridI_fname, gridI_vname, exgrid_fname, exgrid_vname, interp_style = args
            gI = ncio.get_abbr_grid(ncio.Dataset.read(gridI_fname), gridI_vname)
            nI = int(gI["sparse_extent"])
            cen = np.zeros((nI, 2))
            cen[gI["to_sparse"]] = gI["centroid_xy"]
            idx, area = ncio.get_exchange_grid(ncio.Dataset.read(exgrid_fname), exgrid_vname)
            self._add_sheet_arrays(name, nI, idx, area, None, interp_style, cen)
        else:
            gridI, exgrid = args[0], args[1]
            interp_style = args[2] if len(args) > 2 else "Z_INTERP"
            proj = args[3] if len(args) > 3 else None
            self._add_sheet_arrays(name, int(gridI["nI"]), exgrid["indices"], exgrid["overlaps"], proj, interp_style,
                                   gridI.get("centroid_xy"))

    def regrid_matrices(self, str sheet_name, elevmaskI, bool scale=True, bool correctA=True, sigma=(0, 0, 0), conserve=True):
        """_icebin.pyx:164-175 (`conserve` is accepted and, as in icebin_cython.cpp:215-236, not used)."""
        if sheet_name not in self._sheets:
            raise KeyError(sheet_name)
        cdef double[::1] em = np.ascontiguousarray(np.asarray(elevmaskI, np.float64).reshape(-1))
        cdef cicebin.RegridMatrices *crm = cicebin.new_regrid_matrices(
            self.cself, sheet_name.encode(), &em[0] if em.shape[0] else NULL, em.shape[0], scale, correctA,
            sigma[0], sigma[1], sigma[2], conserve)
        cdef RegridMatrices rm = RegridMatrices()
        rm.cself = crm
        rm._keep = self             # RegridMatrices_Dynamic keeps a raw pointer to the sheet (RegridMatrices_Dynamic.hpp:23)
        return rm


def coo_multiply(M, xx, fill=np.nan, ignore_nan=False):
    """icebin.coo_multiply (coo_matvec, pylib/icebin_cython.cpp:158-192)."""
    from icebin_amd import linear
    return linear.coo_multiply(M, xx, fill=fill, ignore_nan=ignore_nan)
