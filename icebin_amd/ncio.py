"""On-disk formats of the regrid path, self-contained.

The reference writes NetCDF through ibmisc::NcIO / netcdf-cxx4 (AbbrGrid.cpp:23-29,167-194,
GCMRegridder.cpp:104-150, IceRegridder.cpp:75-90; "Eigen format" of Weighted_Eigen,
sphinx/source/matrix_formats.rst:9-63, writers modele/global_ec.cpp:571-629, IceCoupler.cpp:473-488).
No NetCDF library is part of this image, so this module carries its own reader/writer of the NetCDF
*classic* container family -- CDF-1, CDF-2 (64-bit offsets) and CDF-5 (64-bit data, which has the
int64 type the Eigen format needs) -- for fixed-size variables, and lays the reference's variable /
dimension / attribute names out in it.  Two things the classic data model cannot express are mapped:
  * attributes holding a LIST of strings (`m.info:sheets`, `BvA.info:dim_names`) become one char
    attribute with the names joined by ','  (names never contain commas);
  * nothing else: dotted names, int64 variables and scalar variables are legal in CDF-5.
Files written by the reference's NetCDF-4 (HDF5) build are READ through hdf5.py, this package's own
reader of the HDF5 subset such files use (Dataset.read dispatches on the signature); writing stays classic.

No regridding arithmetic lives here: arrays in, arrays out.
"""
import struct
from collections import OrderedDict

import numpy as np

# NetCDF classic type codes -> big-endian numpy dtypes
_NC = {1: ">i1", 2: "S1", 3: ">i2", 4: ">i4", 5: ">f4", 6: ">f8", 7: ">u1", 8: ">u2", 9: ">u4", 10: ">i8", 11: ">u8"}
_NC_OF = {"int8": 1, "S1": 2, "int16": 3, "int32": 4, "float32": 5, "float64": 6, "uint8": 7, "uint16": 8,
          "uint32": 9, "int64": 10, "uint64": 11}
_DIM, _VAR, _ATT = 10, 11, 12


def _pad4(n):
    return (4 - n % 4) % 4


class Var:
    """One variable: dims (tuple of dimension names), attrs (OrderedDict), data (numpy array)."""

    def __init__(self, dims, data, attrs=None):
        self.dims, self.data = tuple(dims), data
        self.attrs = OrderedDict(attrs or {})

    def __getattr__(self, k):          # ncvar.base, ncvar.extent ... like netCDF4 / scipy netcdf variables
        try:
            return self.__dict__["attrs"][k]
        except KeyError:
            raise AttributeError(k)


class Dataset:
    """In-memory image of a classic NetCDF file: dims name -> length, attrs, variables name -> Var."""

    def __init__(self):
        self.dims, self.attrs, self.variables = OrderedDict(), OrderedDict(), OrderedDict()

    def add_dim(self, name, n):
        if name in self.dims:
            if self.dims[name] != n:
                raise ValueError("dimension %s redefined: %d != %d" % (name, self.dims[name], n))
        else:
            self.dims[name] = int(n)
        return name

    def add_var(self, name, dtype, dims, data, attrs=None):
        shape = tuple(self.dims[d] for d in dims)
        arr = np.asarray(data if data is not None else 0, dtype=dtype)
        arr = np.ascontiguousarray(arr).reshape(shape) if arr.size == int(np.prod(shape, dtype=np.int64)) else arr
        if arr.shape != shape:
            raise ValueError("variable %s: data shape %s != dims %s" % (name, arr.shape, shape))
        if name in self.variables:
            raise ValueError("variable %s defined twice" % name)
        self.variables[name] = Var(dims, arr, attrs)
        return self.variables[name]

    # ---- writer (CDF-5) ------------------------------------------------------------------------
    def write(self, path):
        def name_b(s):
            b = s.encode("utf-8")
            return struct.pack(">q", len(b)) + b + b"\0" * _pad4(len(b))

        def att_list(attrs):
            if not attrs:
                return struct.pack(">iq", 0, 0)
            out = struct.pack(">iq", _ATT, len(attrs))
            for k, v in attrs.items():
                if isinstance(v, (list, tuple)) and v and isinstance(v[0], str):
                    v = ",".join(v)
                if isinstance(v, str):
                    b = v.encode("utf-8")
                    out += name_b(k) + struct.pack(">iq", 2, len(b)) + b + b"\0" * _pad4(len(b))
                else:
                    a = np.atleast_1d(np.asarray(v))
                    if a.dtype == np.bool_:
                        a = a.astype(np.int32)
                    code = _NC_OF[a.dtype.name]
                    raw = a.astype(_NC[code]).tobytes()
                    out += name_b(k) + struct.pack(">iq", code, a.size) + raw + b"\0" * _pad4(len(raw))
            return out

        dim_ids = {d: i for i, d in enumerate(self.dims)}
        head = b"CDF\x05" + struct.pack(">q", 0)
        if self.dims:
            head += struct.pack(">iq", _DIM, len(self.dims))
            for d, n in self.dims.items():
                head += name_b(d) + struct.pack(">q", n)
        else:
            head += struct.pack(">iq", 0, 0)
        head += att_list(self.attrs)
        # variable headers need the data offsets: build them with a placeholder first to learn the header size
        def var_list(begins):
            if not self.variables:
                return struct.pack(">iq", 0, 0)
            out = struct.pack(">iq", _VAR, len(self.variables))
            for (name, v), beg in zip(self.variables.items(), begins):
                code = _NC_OF["S1" if v.data.dtype.kind == "S" else v.data.dtype.name]
                nbytes = v.data.size * np.dtype(_NC[code]).itemsize
                out += name_b(name) + struct.pack(">q", len(v.dims))
                out += b"".join(struct.pack(">q", dim_ids[d]) for d in v.dims)
                out += att_list(v.attrs) + struct.pack(">iqq", code, nbytes + _pad4(nbytes), beg)
            return out
        hlen = len(head) + len(var_list([0] * len(self.variables)))
        begins, off = [], hlen
        for v in self.variables.values():
            code = _NC_OF["S1" if v.data.dtype.kind == "S" else v.data.dtype.name]
            nbytes = v.data.size * np.dtype(_NC[code]).itemsize
            begins.append(off)
            off += nbytes + _pad4(nbytes)
        with open(path, "wb") as f:
            f.write(head + var_list(begins))
            for v in self.variables.values():
                code = _NC_OF["S1" if v.data.dtype.kind == "S" else v.data.dtype.name]
                raw = np.ascontiguousarray(v.data).astype(_NC[code]).tobytes()
                f.write(raw + b"\0" * _pad4(len(raw)))

    # ---- reader (CDF-1 / CDF-2 / CDF-5) --------------------------------------------------------
    @classmethod
    def read(cls, path):
        with open(path, "rb") as f:
            buf = f.read()
        if buf[:4] == b"\x89HDF":
            return _read_netcdf4(path)
        if buf[:3] != b"CDF" or buf[3] not in (1, 2, 5):
            raise ValueError("%s is not a NetCDF classic file (CDF-1/2/5)" % path)
        ver = buf[3]
        pos = [4]
        cnt_fmt = ">q" if ver == 5 else ">i"          # NON_NEG counts
        off_fmt = ">i" if ver == 1 else ">q"          # begin offsets

        def take(fmt):
            v = struct.unpack_from(fmt, buf, pos[0])
            pos[0] += struct.calcsize(fmt)
            return v[0] if len(v) == 1 else v

        def name():
            n = take(cnt_fmt)
            s = buf[pos[0]:pos[0] + n].decode("utf-8")
            pos[0] += n + _pad4(n)
            return s

        def att_list():
            tag, n = take(">i"), take(cnt_fmt)
            out = OrderedDict()
            if tag == 0:
                return out
            if tag != _ATT:
                raise ValueError("corrupt header: expected attribute list")
            for _ in range(n):
                k = name()
                code, ne = take(">i"), take(cnt_fmt)
                dt = np.dtype(_NC[code])
                raw = buf[pos[0]:pos[0] + ne * dt.itemsize]
                pos[0] += ne * dt.itemsize + _pad4(ne * dt.itemsize)
                if code == 2:
                    out[k] = raw.decode("utf-8")
                else:
                    a = np.frombuffer(raw, dt).astype(dt.newbyteorder("="))
                    out[k] = a[0] if a.size == 1 else a
            return out

        ds = cls()
        numrecs = take(cnt_fmt)
        tag, n = take(">i"), take(cnt_fmt)
        dim_names = []
        if tag == _DIM:
            for _ in range(n):
                d = name()
                ln = take(cnt_fmt)
                ds.dims[d] = ln if ln else numrecs
                dim_names.append(d)
        ds.attrs = att_list()
        tag, n = take(">i"), take(cnt_fmt)
        if tag == _VAR:
            for _ in range(n):
                vn = name()
                nd = take(cnt_fmt)
                ids = [take(cnt_fmt) for _ in range(nd)]
                attrs = att_list()
                code, _vsize, begin = take(">i"), take(cnt_fmt), take(off_fmt)
                dims = tuple(dim_names[i] for i in ids)
                shape = tuple(ds.dims[d] for d in dims)
                dt = np.dtype(_NC[code])
                cnt = int(np.prod(shape)) if shape else 1
                a = np.frombuffer(buf, dt, cnt, begin).reshape(shape)
                a = a if code == 2 else a.astype(dt.newbyteorder("="))
                ds.variables[vn] = Var(dims, a, attrs)
        return ds


def _read_netcdf4(path):
    """A NetCDF-4 (HDF5) file, as the reference's build writes them (NcIO(fname, 'w', "nc4"), global_ec.cpp:539,567):
    through this package's own HDF5 reader (hdf5.py); ICEBIN_USE_NETCDF4=1 asks for the `netCDF4` package instead
    (a cross-check where it is installed).  List-of-strings attributes come back as Python lists."""
    import os
    if os.environ.get("ICEBIN_USE_NETCDF4", "0") != "1":
        from . import hdf5
        return hdf5.read_netcdf4(path)[0]
    import netCDF4
    ds = Dataset()
    with netCDF4.Dataset(path) as nc:
        for d, v in nc.dimensions.items():
            ds.dims[d] = len(v)
        for k in nc.ncattrs():
            ds.attrs[k] = nc.getncattr(k)
        for vn, v in nc.variables.items():
            attrs = OrderedDict((k, v.getncattr(k)) for k in v.ncattrs())
            for k, a in list(attrs.items()):
                if isinstance(a, (list, tuple)) and a and isinstance(a[0], str):
                    attrs[k] = ",".join(a)
            ds.variables[vn] = Var(v.dimensions, np.asarray(v[...]), attrs)
    return ds


def _strlist(v):
    if isinstance(v, str):
        return [s for s in v.split(",") if s]
    return [str(s) for s in np.atleast_1d(v)]


# ---- ibmisc::Indexing / spsparse::SparseSet ncio ---------------------------------------------------
def put_indexing(ds, vname, base, extent, indices):
    """ibmisc::Indexing::ncio: an int scalar carrying base / extent / indices (pylib/icebin/ibgrid.py:89-94)."""
    ds.add_var(vname, np.int32, (), 0, OrderedDict(base=np.asarray(base, np.int64), extent=np.asarray(extent, np.int64),
                                                   indices=np.asarray(indices, np.int32)))


def get_indexing(ds, vname):
    v = ds.variables[vname]
    return np.atleast_1d(v.base), np.atleast_1d(v.extent), np.atleast_1d(v.indices)


def indexing_strides(extent, indices):
    """Strides per dimension; `indices` lists dimension ids by descending stride (GCMRegridder.cpp:43,
    ibgrid.py:65-70)."""
    rank = len(extent)
    strides = np.zeros(rank, np.int64)
    strides[indices[rank - 1]] = 1
    for d in range(rank - 2, -1, -1):
        strides[indices[d]] = strides[indices[d + 1]] * extent[indices[d + 1]]
    return strides


def put_sparse_set(ds, vname, to_sparse, sparse_extent):
    """SparseSet::ncio: `int64 dimB(dimB.dense_extent)`, `dimB:sparse_extent` (matrix_formats.rst:23-27)."""
    to_sparse = np.asarray(to_sparse, np.int64)
    d = ds.add_dim(vname + ".dense_extent", len(to_sparse))
    if vname not in ds.variables:
        ds.add_var(vname, np.int64, (d,), to_sparse, OrderedDict(sparse_extent=np.int64(sparse_extent)))
    return d


def get_sparse_set(ds, vname):
    v = ds.variables[vname]
    return np.asarray(v.data, np.int64), int(v.sparse_extent)


# ---- Weighted_Eigen "Eigen format" (matrix_formats.rst:9-63) -----------------------------------------
def put_weighted(ds, vname, dim_names, dims, sparse_extents, row, col, val, wM, Mw, conservative, scaled=True):
    """Weighted_Eigen::ncio(ncio, vname, {dimB, dimA}): dims are written (once) under their own names,
    M as dense-indexed COO, wM / Mw as dense vectors."""
    dB = put_sparse_set(ds, dim_names[0], dims[0], sparse_extents[0])
    dA = put_sparse_set(ds, dim_names[1], dims[1], sparse_extents[1])
    ds.add_var(vname + ".info", np.int32, (), 0,
               OrderedDict(type="EIGEN", conservative=np.int32(bool(conservative)), scaled=np.int32(bool(scaled)),
                           dim_names=[vname + "." + dim_names[0], vname + "." + dim_names[1]]))
    ds.add_var(vname + ".M.info", np.int64, (), 0,
               OrderedDict(shape=np.asarray([len(dims[0]), len(dims[1])], np.int64), conservative="t" if conservative else "f"))
    nnz = ds.add_dim(vname + ".M.nnz", len(val))
    rank = ds.add_dim(vname + ".M.rank", 2)
    ds.add_var(vname + ".M.indices", np.int32, (nnz, rank), np.stack([row, col], axis=1) if len(val) else np.zeros((0, 2), np.int32))
    ds.add_var(vname + ".M.values", np.float64, (nnz,), val)
    ds.add_var(vname + ".Mw", np.float64, (dA,), Mw)
    ds.add_var(vname + ".wM", np.float64, (dB,), wM)


def get_weighted(ds, vname):
    """-> dict(dim_names, dims, sparse_extents, row, col, val, wM, Mw, conservative, scaled)."""
    info = ds.variables[vname + ".info"]
    if info.attrs.get("type", "EIGEN") != "EIGEN":
        raise ValueError("%s: matrix type %r is not the Eigen format (the zlib 'compressed' format is out of scope)" %
                         (vname, info.attrs.get("type")))
    names = [n.split(".", 1)[1] if n.startswith(vname + ".") else n for n in _strlist(info.dim_names)]
    dims, ext = [], []
    for n in names:
        d, e = get_sparse_set(ds, n)
        dims.append(d)
        ext.append(e)
    ind = np.asarray(ds.variables[vname + ".M.indices"].data, np.int32).reshape(-1, 2)
    return dict(dim_names=names, dims=dims, sparse_extents=ext, row=ind[:, 0].copy(), col=ind[:, 1].copy(),
                val=np.asarray(ds.variables[vname + ".M.values"].data, np.float64),
                wM=np.asarray(ds.variables[vname + ".wM"].data, np.float64),
                Mw=np.asarray(ds.variables[vname + ".Mw"].data, np.float64),
                conservative=bool(info.attrs.get("conservative", 1)), scaled=bool(info.attrs.get("scaled", 1)))


# ---- the IceBin input file: GCMRegridder_Standard::ncio (GCMRegridder.cpp:104-150) -----------------
def put_abbr_grid(ds, vname, to_sparse, sparse_extent, native_area, centroid_xy=None, ijk=None, name="", sproj="",
                  indexing=None):
    """AbbrGrid::ncio (AbbrGrid.cpp:167-194): info, indexing, dim, ijk, native_area, centroid_xy.  The grid
    spec (polygons) is not part of the regrid path and is not written."""
    n = len(to_sparse)
    ds.add_var(vname + ".info", np.int32, (), 0, OrderedDict(coordinates="XY", parameterization="L0", name=name, sproj=sproj))
    if indexing is not None:
        put_indexing(ds, vname + ".indexing", *indexing)
    d = put_sparse_set(ds, vname + ".dim", to_sparse, sparse_extent)
    three, two = ds.add_dim("three", 3), ds.add_dim("two", 2)
    ds.add_var(vname + ".ijk", np.int32, (d, three), np.zeros((n, 3), np.int32) if ijk is None else ijk)
    ds.add_var(vname + ".native_area", np.float64, (d,), native_area)
    ds.add_var(vname + ".centroid_xy", np.float64, (d, two), np.zeros((n, 2)) if centroid_xy is None else centroid_xy)


def get_abbr_grid(ds, vname):
    to_sparse, ext = get_sparse_set(ds, vname + ".dim")
    return dict(to_sparse=to_sparse, sparse_extent=ext,
                native_area=np.asarray(ds.variables[vname + ".native_area"].data, np.float64),
                centroid_xy=np.asarray(ds.variables[vname + ".centroid_xy"].data, np.float64),
                name=ds.variables[vname + ".info"].attrs.get("name", ""))


def put_exchange_grid(ds, vname, indices, overlaps):
    """ExchangeGrid::ncio (AbbrGrid.cpp:23-29): `<v>.indices(<v>.nindices)` interleaved (iA, iI), `<v>.overlaps`."""
    indices = np.ascontiguousarray(indices, np.int32).reshape(-1)
    overlaps = np.ascontiguousarray(overlaps, np.float64)
    ds.add_var(vname + ".indices", np.int32, (ds.add_dim(vname + ".nindices", len(indices)),), indices)
    ds.add_var(vname + ".overlaps", np.float64, (ds.add_dim(vname + ".noverlaps", len(overlaps)),), overlaps)


def get_exchange_grid(ds, vname):
    return (np.asarray(ds.variables[vname + ".indices"].data, np.int32).reshape(-1, 2),
            np.asarray(ds.variables[vname + ".overlaps"].data, np.float64))
