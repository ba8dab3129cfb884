"""GCMRegridder / RegridMatrices: the Cython-exposed surface of IceBin's regrid path
(pylib/_icebin.pyx:50-175) over the C-ABI.  Same method names, keyword names and defaults."""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import check, lib, ptr
from .linear import SparseSet, linear_Weighted

_INTERP = {"Z_INTERP": 0, "ELEV_CLASS_INTERP": 1}     # IceRegridder.hpp:36-39


class RegridMatrices:
    """cdef class RegridMatrices (_icebin.pyx:50-75) over RegridMatrices_Dynamic."""

    def __init__(self, handle, keep):
        self._h = handle
        self._keep = keep

    def __del__(self):
        try:
            _capi.destroy("ibh_regrid_matrices_destroy", getattr(self, "_h", None))
        except Exception:      # interpreter shutdown
            pass
        self._h = None

    def matrix(self, spec_name):
        """Compute a regrid matrix: 'EvI', 'AvI', 'IvA', 'IvE', 'EvA', 'AvE' (also 'AvX', 'XvA',
        'EvX', 'XvE').  Returns a linear_Weighted with its own dims (_icebin.pyx:56-75)."""
        h = C.c_void_p()
        check(lib().ibh_regrid_matrices_matrix(self._h, spec_name.encode(), C.byref(h)))
        return linear_Weighted(h, keep=(self,))

    def matrix_d(self, spec_name, dims=(None, None), scale=True, correctA=False, sigma=(0., 0., 0.)):
        """RegridMatrices_Dynamic::matrix_d(spec, dims, RegridParams(scale, correctA, sigma))
        (RegridMatrices_Dynamic.hpp:51-54).  dims: two SparseSet (appended to) or None."""
        sig = np.asarray(sigma, np.float64)
        h = C.c_void_p()
        d0 = dims[0]._h if dims[0] is not None else None
        d1 = dims[1]._h if dims[1] is not None else None
        check(lib().ibh_regrid_matrices_matrix_d(self._h, spec_name.encode(), d0, d1, int(scale), int(correctA),
                                                ptr(sig), C.byref(h)))
        return linear_Weighted(h, keep=(self, dims))


    def matrix_d_sharded(self, comm, spec_name, dims=(None, None), scale=True, correctA=False):
        """ibh_regrid_matrices_matrix_d_sharded: matrix_d with the assembly shared by the ranks of `comm`
        (icebin_amd.distributed.Communicator).  Collective -- every rank makes the same call -- and every rank receives the whole
        matrix, bitwise what matrix_d builds."""
        h = C.c_void_p()
        d0 = dims[0]._h if dims[0] is not None else None
        d1 = dims[1]._h if dims[1] is not None else None
        check(lib().ibh_regrid_matrices_matrix_d_sharded(self._h, comm._h, spec_name.encode(), d0, d1, int(scale), int(correctA), C.byref(h)))
        return linear_Weighted(h, keep=(self, dims))

    def matrix_batch(self, jobs, sigma=None):
        """The matrices of one coupling step in one call (IceCoupler.cpp:361-468): jobs is a list of
        (spec_name, (dim0, dim1), scale, correctA) -- dims as in matrix_d -- and the results are those of the
        matrix_d calls in that order; independent builds run concurrently inside the library."""
        n = len(jobs)
        names = (C.c_char_p * n)(*[j[0].encode() for j in jobs])
        d0 = (C.c_void_p * n)(*[(j[1][0]._h.value if j[1][0] is not None else None) for j in jobs])
        d1 = (C.c_void_p * n)(*[(j[1][1]._h.value if j[1][1] is not None else None) for j in jobs])
        sc = (C.c_int32 * n)(*[int(j[2]) for j in jobs])
        ca = (C.c_int32 * n)(*[int(j[3]) for j in jobs])
        out = (C.c_void_p * n)()
        sig = None if sigma is None else np.asarray(sigma, np.float64)
        check(lib().ibh_regrid_matrices_matrix_batch(self._h, n, names, d0, d1, sc, ca, ptr(sig), out))
        return [linear_Weighted(C.c_void_p(out[k]), keep=(self, jobs[k][1])) for k in range(n)]


class _Sheet:
    def __init__(self, handle, nI, arrays, interp_style="Z_INTERP", centroid=None):
        self.h, self.nI, self.arrays = handle, nI, arrays
        self.interp_style, self.centroid = interp_style, centroid


class GCMRegridder:
    """cdef class GCMRegridder (_icebin.pyx:77-175).

    The reference builds it from NetCDF grid files (4-arg form) or loads it from an IceBin
    input file (1-arg form); NetCDF is outside this build's scope, so
    the in-memory form takes the arrays those files hold:
        GCMRegridder(gridA, hcdefs, correctA)   with gridA = dict(nA=, to_sparse=, native_area=)
    """

    def __init__(self, *args):
        if len(args) == 3:
            gridA, hcdefs, correctA = args
            self._nA = int(gridA["nA"])
            self._A_to_sparse = np.ascontiguousarray(gridA["to_sparse"], np.int64)
            self._A_native = np.ascontiguousarray(gridA["native_area"], np.float64)
            self._hcdefs = np.ascontiguousarray(hcdefs, np.float64)
            self.correctA = bool(correctA)
            # Indexing({"A","HC"}, {0,0}, {nA,nhc}, {1,0}): HC has the largest stride (icebin_cython.cpp:69)
            self._hc_strides = (1, self._nA)
        elif len(args) == 1:
            # GCMRegridder(regridder_fname): "Load from existing GCMRegridder file" (_icebin.pyx:81-88)
            self._sheets = {}
            self._read(args[0], "m")
            return
        elif len(args) == 4:
            # GCMRegridder(gridA_fname, gridA_vname, hcdefs, correctA) (_icebin.pyx:89-100): the grid file holds
            # an AbbrGrid-shaped record of A (dim, native_area) under vname
            from . import ncio
            gridA_fname, gridA_vname, hcdefs, correctA = args
            a = ncio.get_abbr_grid(ncio.Dataset.read(gridA_fname), gridA_vname)
            self.__init__(dict(nA=a["sparse_extent"], to_sparse=a["to_sparse"], native_area=a["native_area"]), hcdefs, correctA)
            return
        else:
            raise ValueError("Invalid arguments: {}".format(args))
        self._sheets = {}

    # ---- GCMRegridder_Standard::ncio (GCMRegridder.cpp:104-150) -----------------------------------
    def ncio_write(self, fname, vname="m"):
        """Write the IceBin input file (the arrays the regrid path reads; grid specs / polygons are not
        part of it): `m.info`, `m.agridA.*`, `m.indexingHC`, `m.hcdefs`, and per sheet `m.<sheet>.info`,
        `.gridA_proj_area`, `.agridI.*`, `.aexgrid.indices/overlaps`."""
        from . import ncio
        ds = ncio.Dataset()
        names = list(self._sheets)
        ds.add_var(vname + ".info", np.int32, (), 0, dict(correctA=np.int32(self.correctA), sheets=names))
        ncio.put_abbr_grid(ds, vname + ".agridA", self._A_to_sparse, self._nA, self._A_native, name="gridA")
        nhc = len(self._hcdefs)
        # Indexing({"A","HC"}, base {0,0}, extent {nA,nhc}, indices by descending stride)
        ncio.put_indexing(ds, vname + ".indexingHC", [0, 0], [self._nA, nhc], [1, 0] if self._hc_strides[1] >= self._hc_strides[0] else [0, 1])
        ds.add_var(vname + ".hcdefs", np.float64, (ds.add_dim(vname + ".nhc", nhc),), self._hcdefs)
        ds.add_dim("agridA.ndata", len(self._A_to_sparse))
        for name in names:
            s = self._sheets[name]
            idx, area, proj = s.arrays
            v = vname + "." + name
            ds.add_var(v + ".info", np.int32, (), 0, dict(name=name, interp_style=s.interp_style))
            ds.add_var(v + ".gridA_proj_area", np.float64, ("agridA.ndata",), proj)
            cen = None if s.centroid is None else np.asarray(s.centroid, np.float64).reshape(-1, 2)
            ncio.put_abbr_grid(ds, v + ".agridI", np.arange(s.nI, dtype=np.int64), s.nI, np.zeros(s.nI), centroid_xy=cen, name="gridI")
            ncio.put_exchange_grid(ds, v + ".aexgrid", idx, area)
        ds.write(fname)

    def _read(self, fname, vname):
        from . import ncio
        ds = ncio.Dataset.read(fname)
        info = ds.variables[vname + ".info"]
        a = ncio.get_abbr_grid(ds, vname + ".agridA")
        self._nA = int(a["sparse_extent"])
        self._A_to_sparse = np.ascontiguousarray(a["to_sparse"], np.int64)
        self._A_native = np.ascontiguousarray(a["native_area"], np.float64)
        self._hcdefs = np.ascontiguousarray(ds.variables[vname + ".hcdefs"].data, np.float64)
        self.correctA = bool(info.correctA)
        _, extent, indices = ncio.get_indexing(ds, vname + ".indexingHC")
        st = ncio.indexing_strides(extent, indices)                  # a file may carry either layout
        self._hc_strides = (int(st[0]), int(st[1]))
        for name in ncio._strlist(info.sheets):
            v = vname + "." + name
            sinfo = ds.variables[v + ".info"]
            gI = ncio.get_abbr_grid(ds, v + ".agridI")
            nI = int(gI["sparse_extent"])
            cen = None
            if np.any(gI["centroid_xy"] != 0):
                cen = np.zeros((nI, 2))
                cen[gI["to_sparse"]] = gI["centroid_xy"]          # dense -> sparse ice index
            idx, area = ncio.get_exchange_grid(ds, v + ".aexgrid")
            self.add_sheet(name, dict(nI=nI, centroid_xy=cen), dict(indices=idx, overlaps=area), str(sinfo.interp_style),
                           np.asarray(ds.variables[v + ".gridA_proj_area"].data, np.float64))

    @property
    def nA(self):
        return self._nA

    @property
    def nhc(self):
        return len(self._hcdefs)

    @property
    def nE(self):
        return self._nA * len(self._hcdefs)      # GCMRegridder.hpp:273

    def add_sheet(self, name, gridI, exgrid, interp_style="Z_INTERP", gridA_proj_area=None):
        """name: sheet name; gridI = dict(nI=[, centroid_xy=f64[nI,2] by sparse index, needed only for
        sigma != 0]); exgrid = dict(indices=int32[nX,2] (iA,iI), overlaps=f64[nX]);
        gridA_proj_area: projected area of the realised A cells (IceRegridder::init computes it by
        projecting A's polygons, IceRegridder.cpp:109-118); None = no projection (== native, :106-108)."""
        if interp_style not in _INTERP:
            raise ValueError("unknown interp_style %r" % (interp_style,))
        idx = np.ascontiguousarray(exgrid["indices"], np.int32).reshape(-1)
        area = np.ascontiguousarray(exgrid["overlaps"], np.float64)
        proj = self._A_native if gridA_proj_area is None else np.ascontiguousarray(gridA_proj_area, np.float64)
        cen = gridI.get("centroid_xy")
        cen = None if cen is None else np.ascontiguousarray(cen, np.float64).reshape(-1)
        assert cen is None or len(cen) == 2 * int(gridI["nI"])
        d = _capi.RegridderDesc(
            nX=len(area), ex_indices=ptr(idx).value, ex_area=ptr(area).value, nI=int(gridI["nI"]),
            nA=self._nA, nA_dense=len(self._A_to_sparse), A_to_sparse=ptr(self._A_to_sparse).value,
            A_native_area=ptr(self._A_native).value, A_proj_area=ptr(proj).value,
            nhc=len(self._hcdefs), hcdefs=ptr(self._hcdefs).value,
            hc_stride_A=self._hc_strides[0], hc_stride_HC=self._hc_strides[1], interp_style=_INTERP[interp_style],
            I_centroid_xy=None if cen is None else ptr(cen).value)
        h = C.c_void_p()
        check(lib().ibh_regridder_create(C.byref(d), C.byref(h)))
        self._sheets[name] = _Sheet(h, int(gridI["nI"]), (idx, area, proj), interp_style, cen)

    def __del__(self):
        try:
            for s in getattr(self, "_sheets", {}).values():
                _capi.destroy("ibh_regridder_destroy", s.h)
                s.h = None
        except Exception:      # interpreter shutdown
            pass

    def wA(self, sheet_name, snative, fill=0.):
        """Returns weights (as a vector) of overall grid (_icebin.pyx:116-125)."""
        if snative == "native":
            native = True
        elif snative == "proj":
            native = False
        else:
            raise ValueError("Invalid argument: snative must be 'native' or 'proj'")
        out = np.empty(self._nA, np.float64)
        check(lib().ibh_regridder_wA(self._sheets[sheet_name].h, int(native), float(fill), ptr(out)))
        return out

    def regrid_matrices(self, sheet_name, elevmaskI, scale=True, correctA=True, sigma=(0, 0, 0), conserve=True):
        """_icebin.pyx:164-175.  `conserve` is accepted and, as in the reference's C++ shim
        (icebin_cython.cpp:215-236), not used."""
        sheet = self._sheets[sheet_name]       # KeyError like ice_regridders().index.at()
        sig = np.asarray(sigma, np.float64)
        h = C.c_void_p()
        if hasattr(elevmaskI, "is_cuda") and elevmaskI.is_cuda:
            # elevation mask already in HBM (torch.float64 CUDA tensor): no trip over PCIe
            import torch
            em = elevmaskI.reshape(-1).contiguous()
            assert em.dtype == torch.float64
            st = torch.cuda.current_stream(em.device).cuda_stream
            check(lib().ibh_regrid_matrices_create_device(sheet.h, C.c_void_p(em.data_ptr()), em.numel(), int(scale),
                                                         int(correctA), ptr(sig), C.c_void_p(st), C.byref(h)))
            return RegridMatrices(h, keep=(self,))
        em = np.ascontiguousarray(np.asarray(elevmaskI, np.float64).reshape(-1))
        check(lib().ibh_regrid_matrices_create(sheet.h, ptr(em), len(em), int(scale), int(correctA), ptr(sig), C.byref(h)))
        return RegridMatrices(h, keep=(self,))


def from_synthetic(grids):
    """GCMRegridder with one sheet 'greenland' from icebin_amd.synthetic.make_grids() output."""
    g = grids
    mm = GCMRegridder(dict(nA=g["nA"], to_sparse=g["A_to_sparse"], native_area=g["A_native_area"]), g["hcdefs"], True)
    mm._hc_strides = (int(g["hc_stride_A"]), int(g["hc_stride_HC"]))
    mm.add_sheet("greenland", dict(nI=g["nI"], centroid_xy=g.get("I_centroid_xy")), dict(indices=g["ex_indices"], overlaps=g["ex_area"]),
                 "Z_INTERP" if g.get("interp_style", 0) == 0 else "ELEV_CLASS_INTERP", g["A_proj_area"])
    return mm
