"""The self-contained NetCDF-classic container (icebin_amd/ncio.py) and the reference's on-disk layouts
(SURVEY.md 8f rank 2): CPU part.  The reader is checked against an independent implementation
(scipy.io.netcdf_file writes CDF-1 / CDF-2), the CDF-5 writer against the reader, and the Eigen-format /
IceBin-input-file layouts against the names matrix_formats.rst:9-63 and GCMRegridder.cpp:104-150 spell out."""
import os

import numpy as np
import pytest

from icebin_amd import ncio


def test_reader_against_scipy_netcdf(tmp_path):
    from scipy.io import netcdf_file
    for version in (1, 2):
        p = str(tmp_path / ("scipy_v%d.nc" % version))
        f = netcdf_file(p, "w", version=version)
        f.history = "made by scipy"
        f.createDimension("n", 5)
        f.createDimension("BvA.M.rank", 2)
        v = f.createVariable("BvA.M.indices", "i", ("n", "BvA.M.rank"))
        v[:] = np.arange(10).reshape(5, 2)
        v.shape_att = np.array([5, 6], np.int32)
        w = f.createVariable("vals", "d", ("n",))
        w[:] = np.linspace(0, 1, 5)
        w.units = "m2"
        s = f.createVariable("scalar", "i", ())
        s.data[...] = 7                 # (scipy's assignValue does not handle 0-d arrays)
        f.close()
        ds = ncio.Dataset.read(p)
        assert ds.dims == {"n": 5, "BvA.M.rank": 2} and ds.attrs["history"] == "made by scipy"
        assert np.array_equal(ds.variables["BvA.M.indices"].data, np.arange(10).reshape(5, 2))
        assert np.array_equal(ds.variables["BvA.M.indices"].shape_att, [5, 6])
        assert np.array_equal(ds.variables["vals"].data, np.linspace(0, 1, 5)) and ds.variables["vals"].units == "m2"
        assert ds.variables["scalar"].data == 7 and ds.variables["scalar"].dims == ()


def test_cdf5_roundtrip_all_types(tmp_path):
    rng = np.random.default_rng(1)
    ds = ncio.Dataset()
    ds.attrs["title"] = "roundtrip"
    ds.add_dim("a", 3); ds.add_dim("b.c", 4); ds.add_dim("empty", 0)
    for i, dt in enumerate(("int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64", "float32", "float64")):
        x = (rng.random((3, 4)) * 100).astype(dt)
        ds.add_var("v." + dt, dt, ("a", "b.c"), x, {"k": np.asarray([1, 2, 3], dt), "note": "n%d" % i, "one": np.int64(2 ** 40)})
    ds.add_var("big", np.int64, ("a",), [2 ** 62, -2 ** 62, 5])
    ds.add_var("nothing", np.float64, ("empty",), np.zeros(0))
    ds.add_var("s", np.int32, (), 0, {"names": ["x.y", "z"], "nan": np.float64("nan")})
    p = str(tmp_path / "t.nc")
    ds.write(p)
    assert open(p, "rb").read(4) == b"CDF\x05"
    rd = ncio.Dataset.read(p)
    assert rd.dims == ds.dims and rd.attrs["title"] == "roundtrip"
    for k, v in ds.variables.items():
        r = rd.variables[k]
        assert r.dims == v.dims and r.data.dtype == v.data.dtype and np.array_equal(r.data, v.data), k
    assert rd.variables["s"].names == "x.y,z" and ncio._strlist(rd.variables["s"].names) == ["x.y", "z"]
    assert np.isnan(rd.variables["s"].nan) and rd.variables["v.int8"].one == 2 ** 40
    assert np.array_equal(rd.variables["v.uint16"].k, [1, 2, 3])


def test_eigen_format_layout_and_roundtrip(tmp_path):
    # the example of matrix_formats.rst:17-40: BvA with dimB (5 of 40) and dimA (6 of 50), 6 nonzeros
    dimB, dimA = np.array([3, 7, 11, 20, 39]), np.array([17, 1, 2, 30, 44, 49])
    row, col = np.array([0, 0, 1, 2, 3, 4], np.int32), np.array([0, 1, 1, 2, 4, 5], np.int32)
    val = np.array([28.9, 1.5, 2.5, -0.0, 1e-300, 3.141592653589793])
    wM, Mw = np.arange(5) + 0.5, np.arange(6) * 1.25
    ds = ncio.Dataset()
    ncio.put_weighted(ds, "BvA", ("dimB", "dimA"), (dimB, dimA), (40, 50), row, col, val, wM, Mw, conservative=False)
    assert ds.dims == {"dimB.dense_extent": 5, "dimA.dense_extent": 6, "BvA.M.nnz": 6, "BvA.M.rank": 2}
    assert set(ds.variables) == {"dimB", "dimA", "BvA.info", "BvA.M.info", "BvA.M.indices", "BvA.M.values", "BvA.Mw", "BvA.wM"}
    assert ds.variables["dimB"].data.dtype == np.int64 and ds.variables["dimB"].sparse_extent == 40
    assert ds.variables["BvA.info"].type == "EIGEN" and ds.variables["BvA.info"].conservative == 0
    assert ds.variables["BvA.M.info"].conservative == "f" and list(ds.variables["BvA.M.info"].shape) == [5, 6]
    assert ds.variables["BvA.Mw"].dims == ("dimA.dense_extent",) and ds.variables["BvA.wM"].dims == ("dimB.dense_extent",)
    # a second matrix sharing dimA re-uses the dimension variable (IceCoupler.cpp:479-487 writes dims once)
    ncio.put_weighted(ds, "CvA", ("dimC", "dimA"), (np.arange(2), dimA), (2, 50), [0], [5], [1.0], [1, 1], Mw, conservative=True)
    p = str(tmp_path / "m.nc")
    ds.write(p)
    d = ncio.get_weighted(ncio.Dataset.read(p), "BvA")
    assert d["dim_names"] == ["dimB", "dimA"] and d["sparse_extents"] == [40, 50] and not d["conservative"]
    assert np.array_equal(d["dims"][0], dimB) and np.array_equal(d["dims"][1], dimA)
    assert np.array_equal(d["row"], row) and np.array_equal(d["col"], col)
    assert np.array_equal(d["val"].view(np.uint64), val.view(np.uint64))          # bit-exact, incl. -0.0
    assert np.array_equal(d["wM"], wM) and np.array_equal(d["Mw"], Mw)
    assert ncio.get_weighted(ncio.Dataset.read(p), "CvA")["conservative"]


def test_indexing_strides_both_layouts():
    # Indexing({"A","HC"}, {0,0}, {nA,nhc}, {1,0}): HC has the largest stride (icebin_cython.cpp:69)
    assert list(ncio.indexing_strides([12960, 40], [1, 0])) == [1, 12960]
    assert list(ncio.indexing_strides([12960, 40], [0, 1])) == [40, 1]


def test_other_containers_are_refused_clearly(tmp_path):
    # (HDF5 files go to the package's own reader: tests/test_hdf5.py)
    p = str(tmp_path / "x.nc")
    open(p, "wb").write(b"garbage")
    with pytest.raises(ValueError, match="not a NetCDF classic"):
        ncio.Dataset.read(p)


# ---- the C++ container (icebin_amd/host/ncio.hpp) against this one ---------------------------------------------------------
def _ncio_exe():
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe, src = os.path.join(root, "tests", "cpp", "test_ncio"), os.path.join(root, "tests", "cpp", "test_ncio.cpp")
    hdr = os.path.join(root, "icebin_amd", "host", "ncio.hpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-DICEBIN_NCIO_HDF5", "-o", exe, src, "-lz"])
    return exe


def test_cpp_container_is_read_by_python_and_vice_versa(tmp_path):
    import subprocess
    exe = _ncio_exe()
    f1 = str(tmp_path / "from_cpp.nc")
    assert subprocess.run([exe, "write", f1]).returncode == 0
    ds = ncio.Dataset.read(f1)                                  # C++ wrote, Python reads
    assert ds.attrs["title"] == "ncio.hpp round trip"
    assert list(ds.dims.items()) == [("dimB.dense_extent", 4), ("BvA.M.nnz", 3), ("BvA.M.rank", 2), ("odd", 5)]
    assert np.array_equal(ds.variables["dimB"].data, [5, 1 << 40, -3, 7]) and int(ds.variables["dimB"].sparse_extent) == 1234567890123
    assert np.array_equal(ds.variables["BvA.M.indices"].data, [[0, 1], [2, 3], [4, 5]])
    assert np.array_equal(ds.variables["BvA.M.values"].data, [1.5, -2.25e-300, 3.0e300])
    assert ds.variables["m.info"].sheets == "greenland,antarctica" and ds.variables["odd.bytes"].data.tobytes() == b"abcde"
    f2 = str(tmp_path / "from_python.nc")                       # Python writes the same content, C++ dumps both: identical dumps
    ds.write(f2)
    d1 = subprocess.run([exe, "dump", f1], capture_output=True, text=True)
    d2 = subprocess.run([exe, "dump", f2], capture_output=True, text=True)
    assert d1.returncode == 0 and d2.returncode == 0 and d1.stdout == d2.stdout and "wsum=9.0000000000000011e+300" in d1.stdout
    # scipy's independent writer (CDF-1) through the C++ reader
    from scipy.io import netcdf_file
    f3 = str(tmp_path / "from_scipy.nc")
    with netcdf_file(f3, "w") as nc:
        nc.createDimension("n", 3)
        v = nc.createVariable("x", "d", ("n",))
        v[:] = [1.0, 2.0, 4.0]
        v.units = "m"
    d3 = subprocess.run([exe, "dump", f3], capture_output=True, text=True)
    assert d3.returncode == 0 and "var x (n) type6[3] wsum=17" in d3.stdout and "att units char[1] 'm'" in d3.stdout
    bad = str(tmp_path / "bad.nc")
    open(bad, "wb").write(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
    r = subprocess.run([exe, "dump", bad], capture_output=True, text=True)
    assert r.returncode == 1 and "hdf5:" in r.stdout              # handed to hdf5.hpp, which refuses the image loudly


# ---- the C++ HDF5 reader (icebin_amd/host/hdf5.hpp) against the Python one (icebin_amd/hdf5.py) -------------------------------
_NC_CODE = {"int8": 1, "int16": 3, "int32": 4, "float32": 5, "float64": 6, "uint8": 7, "uint16": 8, "uint32": 9, "int64": 10, "uint64": 11}


def _dump_value(x):
    """tests/cpp/test_ncio.cpp's dump_array() for what the Python readers return"""
    if isinstance(x, (list, tuple)) and all(isinstance(s, str) for s in x):
        x = ",".join(x)                                         # a list of strings is ONE comma-joined char attribute in C++
    if isinstance(x, str):
        return "char[%d] '%s'" % (len(x.encode()), x)
    a = np.ascontiguousarray(x).reshape(-1)
    if a.dtype.kind == "S":
        b = a.tobytes()
        return "char[%d] '%s'" % (len(b), b.decode())
    w = (np.arange(a.size) % 7 + 1).astype(np.float64)
    s = float(np.cumsum(a.astype(np.float64) * w)[-1]) if a.size else 0.0    # cumsum adds in order, like the C++ loop
    return "type%d[%d] wsum=%s" % (_NC_CODE[a.dtype.name], a.size, "%.17g" % s)


def _dump_dataset(ds):
    out = ["dim %s %d" % kv for kv in ds.dims.items()]
    out += ["gatt %s %s" % (k, _dump_value(v)) for k, v in ds.attrs.items()]
    for name, v in ds.variables.items():
        out.append("var %s (%s) %s" % (name, ",".join(v.dims), _dump_value(v.data)))
        out += ["  att %s %s" % (k, _dump_value(x)) for k, x in v.attrs.items()]
    return "\n".join(out) + "\n"


def _cpp_dump(path):
    import subprocess
    return subprocess.run([_ncio_exe(), "dump", str(path)], capture_output=True, text=True)


def test_cpp_hdf5_reader_agrees_with_the_python_reader(tmp_path):
    from icebin_amd import hdf5
    from test_hdf5 import _handmade_file_with_string_list_attributes
    # (1) the hand-assembled image: superblock 0, version-1 header, a list of variable-length strings in a global heap
    p1 = tmp_path / "handmade.nc"
    p1.write_bytes(_handmade_file_with_string_list_attributes())
    r = _cpp_dump(p1)
    assert r.returncode == 0, r.stdout
    assert r.stdout == _dump_dataset(hdf5.read_netcdf4(str(p1))[0])
    assert "gatt sheets char[21] 'greenland,antarctica,'" in r.stdout and "gatt shape type4[2] wsum=153344" in r.stdout
    # (2) scipy's MATLAB 7.3 test file: old-style group (symbol table, local heap), a 512-byte user block, contiguous layout
    import scipy.io
    p2 = os.path.join(os.path.dirname(scipy.io.__file__), "matlab", "tests", "data", "testhdf5_7.4_GLNX86.mat")
    if os.path.exists(p2):
        r = _cpp_dump(p2)
        assert r.returncode == 0 and r.stdout == _dump_dataset(hdf5.read_netcdf4(p2)[0])
        assert "var testdouble (phony_dim_0_9,phony_dim_1_1) type6[9]" in r.stdout
    # (3) the reference's own NetCDF-4 files: superblock 2, version-2 headers, chunked + deflate + shuffle, dimension scales
    ref = "/root/reference/examples/example1"
    if os.path.isdir(ref):
        for fn in sorted(os.listdir(ref)):
            if not fn.endswith(".nc"):
                continue
            path = os.path.join(ref, fn)
            r = _cpp_dump(path)
            assert r.returncode == 0, r.stdout
            assert r.stdout == _dump_dataset(hdf5.read_netcdf4(path)[0]), fn
            raw = open(path, "rb").read()
            # damage: a flipped metadata byte is caught by the checksum; truncation and a broken deflate stream fail loudly
            bad = bytearray(raw)
            bad[raw.index(b"OHDR") + 9] ^= 0x40
            cases = {"flip": bytes(bad), "half": raw[:len(raw) // 2], "tail": raw[:len(raw) - 1000], "head": raw[:200]}
            for key, img in cases.items():
                q = tmp_path / (key + ".nc")
                q.write_bytes(img)
                r = _cpp_dump(q)
                assert r.returncode == 1 and r.stdout.startswith("ERROR ncio:") and "hdf5:" in r.stdout, (fn, key, r.stdout)
                if key == "flip":
                    assert "checksum" in r.stdout
