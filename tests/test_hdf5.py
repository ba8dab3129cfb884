"""The self-contained HDF5 reader (icebin_amd/hdf5.py) behind ncio.Dataset.read for the reference's NetCDF-4 files.

No HDF5 / NetCDF library exists in the image, so the reader is pinned by
  * the published known answers of its checksum (Bob Jenkins' lookup3 self-test values) -- every version-2 metadata
    block of a file read is verified with it, which is independent evidence that blocks are parsed where the writer put them;
  * the reference's own NetCDF-4 example files (examples/example1/*.nc, present in the build container only) against the
    committed fixture tests/golden/nc4_example1.json, plus physical sanity of what comes out (the grid-cell areas of the
    2 x 2.5 degree grid add up to the area of the Earth);
  * an HDF5 file of a different vintage that ships with scipy (MATLAB 7.3: superblock 0 behind a 512-byte user block,
    version-1 object headers, old-style group) with the values scipy's own test suite expects of it.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from icebin_amd import hdf5, ncio

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/examples/example1"
needs_reference = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference's example files exist in the build container only")


def test_lookup3_known_answers():
    # lookup3.c, driver5(): the values Bob Jenkins publishes with the function
    assert hdf5.lookup3(b"") == 0xdeadbeef
    assert hdf5.lookup3(b"", 0xdeadbeef) == 0xbd5b7dde
    assert hdf5.lookup3(b"Four score and seven years ago") == 0x17770551
    assert hdf5.lookup3(b"Four score and seven years ago", 1) == 0xcd628161
    # every tail length 0..12 and the 12-byte block loop are taken
    seen = {hdf5.lookup3(bytes(range(n))) for n in range(40)}
    assert len(seen) == 40


def _fletcher32_loop(data):
    """H5_checksum_fletcher32 restated word by word (end-around-carry reduction every 360 words)."""
    s1 = s2 = 0
    n = len(data) // 2
    i = 0
    while n:
        t = min(n, 360)
        n -= t
        for _ in range(t):
            s1 += (data[i] << 8) | data[i + 1]
            s2 += s1
            i += 2
        s1 = (s1 & 0xffff) + (s1 >> 16)
        s2 = (s2 & 0xffff) + (s2 >> 16)
    if len(data) % 2:
        s1 += data[i] << 8
        s2 += s1
        s1 = (s1 & 0xffff) + (s1 >> 16)
        s2 = (s2 & 0xffff) + (s2 >> 16)
    s1 = (s1 & 0xffff) + (s1 >> 16)
    s2 = (s2 & 0xffff) + (s2 >> 16)
    return (s2 << 16) | s1


def test_fletcher32_is_the_librarys_loop():
    rng = np.random.default_rng(3)
    for n in (0, 1, 2, 3, 7, 720, 721, 4096, 100001):
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert hdf5.fletcher32(b) == _fletcher32_loop(b), n
    for b in (b"\xff" * 4096, b"\xff\xff", b"\0" * 100, b"\xff" * 131070):     # sums that are multiples of 65535
        assert hdf5.fletcher32(b) == _fletcher32_loop(b)


@needs_reference
def test_reference_example_files_against_the_committed_fixture():
    golden = json.load(open(os.path.join(HERE, "golden", "nc4_example1.json")))
    assert len(golden) == 2
    for fname, g in golden.items():
        path = os.path.join(REF, fname)
        raw = open(path, "rb").read()
        assert hashlib.sha256(raw).hexdigest() == g["sha256"], "the example file itself changed"
        ds, f = hdf5.read_netcdf4(path)
        assert f.sb_version == g["superblock"] and f.checked["lookup3"] == g["lookup3_blocks_verified"] > 0
        assert dict(ds.dims) == g["dims"]
        assert set(ds.variables) == set(g["variables"])
        for name, gv in g["variables"].items():
            v = ds.variables[name]
            a = np.ascontiguousarray(v.data)
            assert list(v.dims) == gv["dims"] and a.dtype.name == gv["dtype"] and list(a.shape) == gv["shape"], name
            assert hashlib.sha256(a.astype(a.dtype.newbyteorder("<")).tobytes()).hexdigest() == gv["sha256"], name
            assert {k: (x if isinstance(x, (str, list)) else np.asarray(x).tolist()) for k, x in v.attrs.items()} == gv["attrs"], name
        for k, x in g["attrs"].items():
            got = ds.attrs[k]
            assert (got if isinstance(got, (str, list)) else np.asarray(got).tolist()) == x


@needs_reference
def test_reference_example_files_make_physical_sense():
    # what no parser bug would produce by accident: the ModelE 2 x 2.5 degree grid, whose cell areas add up to the sphere
    ds = ncio.Dataset.read(os.path.join(REF, "130516-Regrid_examples_e4f40-hc40-g5_JUL1956.ijhce4f40-hc40.nc"))     # (dispatch on the signature)
    assert dict(ds.dims) == {"shnhgm": 3, "lon": 144, "lat": 90, "nhc": 41}
    lon, lat = ds.variables["lon"].data, ds.variables["lat"].data
    assert np.allclose(np.diff(lon), 2.5) and lon[0] == -178.75 and lon[-1] == 178.75
    assert lat[0] == -90 and lat[-1] == 90 and np.all(np.diff(lat) > 0)
    area = ds.variables["axyp"].data.astype(np.float64).sum()
    assert abs(area / (4 * np.pi * 6.371e6 ** 2) - 1) < 1e-3
    frac = ds.variables["frac"].data
    assert frac.shape == (41, 90, 144) and ds.variables["frac"].dims == ("nhc", "lat", "lon")
    assert frac.min() >= 0 and frac.max() <= 1
    assert ds.variables["frac"].units == "1" and ds.variables["tsurf"].long_name == "surface air temperature"
    # hidden HDF5 / NetCDF-4 bookkeeping attributes do not leak
    assert not {"CLASS", "NAME", "DIMENSION_LIST", "REFERENCE_LIST", "_Netcdf4Dimid"} & set(ds.variables["lon"].attrs)


@needs_reference
def test_a_flipped_metadata_byte_is_caught_by_the_checksum():
    path = os.path.join(REF, "130516-Regrid_examples_e4f40-hc40-g5_JUL1956.aije4f40-hc40-prec.nc")
    raw = bytearray(open(path, "rb").read())
    at = raw.index(b"OHDR") + 9                     # inside the root object header
    raw[at] ^= 0x40
    with pytest.raises(hdf5.H5Error, match="checksum"):
        hdf5.read_netcdf4(bytes(raw))


def test_matlab73_file_of_scipy_old_style_structures():
    import scipy.io
    path = os.path.join(os.path.dirname(scipy.io.__file__), "matlab", "tests", "data", "testhdf5_7.4_GLNX86.mat")
    if not os.path.exists(path):
        pytest.skip("scipy's test data is not installed")
    f = hdf5.File(path)
    assert (f.sb_version, f.base) == (0, 512)       # superblock 0 behind MATLAB's 512-byte user block
    objs = dict(f.walk())
    assert list(objs) == ["testdouble"]
    o = objs["testdouble"]
    assert o.shape == (9, 1) and o.dtype.np == np.dtype("<f8") and o.layout[0] == "contiguous"
    # scipy/io/matlab/tests/test_mio.py: 'double': arange(0, 2 * pi + pi / 4 / 2, pi / 4)
    assert np.allclose(o.read().reshape(-1), np.arange(9) * np.pi / 4, rtol=0, atol=1e-15)
    assert bytes(np.asarray(o.attrs["MATLAB_class"]).reshape(-1)[0]) == b"double"


@needs_reference
def test_truncated_and_damaged_files_raise_h5error():
    path = os.path.join(REF, "130516-Regrid_examples_e4f40-hc40-g5_JUL1956.aije4f40-hc40-prec.nc")
    raw = open(path, "rb").read()
    for cut in (len(raw) // 2, len(raw) - 1000, 200, 60):
        with pytest.raises(hdf5.H5Error):
            hdf5.read_netcdf4(raw[:cut])
    # a damaged chunk (the deflate stream of the data, not the checksummed metadata)
    ds, f = hdf5.read_netcdf4(raw)
    prec = [o for n, o in f.walk() if n == "prec"][0]
    addr = next(f._chunks(prec.layout[1], 2))[1]
    bad = bytearray(raw)
    bad[addr + 10:addr + 60] = b"\xff" * 50
    with pytest.raises(hdf5.H5Error):
        hdf5.read_netcdf4(bytes(bad))


def _handmade_file_with_string_list_attributes():
    """A minimal HDF5 image assembled by hand from the format specification: superblock 0, a version-1 object header (root
    group without members) carrying three version-1 attribute messages -- a LIST of variable-length strings (what
    `m.info:sheets` / `BvA.info:dim_names` are in the reference's nc4 files), a fixed-length string and an int32 array -- and the
    global heap collection the variable-length strings live in."""
    import struct
    U = 0xFFFFFFFFFFFFFFFF

    def pad8(b):
        return b + b"\0" * (-len(b) % 8)

    names = [b"greenland", b"antarctica", b""]
    gcol_at = 1024
    heap = b""
    for i, nm in enumerate(names, 1):
        heap += struct.pack("<HHIQ", i, 1, 0, len(nm)) + pad8(nm)
    heap += struct.pack("<HHIQ", 0, 0, 0, 0)
    gcol = b"GCOL" + bytes([1, 0, 0, 0]) + struct.pack("<Q", 16 + len(heap)) + heap

    def attr(name, dtype, space, data):
        nm = name + b"\0"
        return bytes([1, 0]) + struct.pack("<HHH", len(nm), len(dtype), len(space)) + pad8(nm) + pad8(dtype) + pad8(space) + data

    vstr = bytes([0x19, 0x01, 0, 0]) + struct.pack("<I", 16) + bytes([0x13, 0, 0, 0]) + struct.pack("<I", 1)
    space1 = lambda n: bytes([1, 1, 0, 0, 0, 0, 0, 0]) + struct.pack("<Q", n)
    scalar = bytes([1, 0, 0, 0, 0, 0, 0, 0])
    vdata = b"".join(struct.pack("<IQI", len(nm), gcol_at, i) for i, nm in enumerate(names, 1))
    fixed = bytes([0x13, 0, 0, 0]) + struct.pack("<I", 8)
    i32 = bytes([0x10, 0x08, 0, 0]) + struct.pack("<I", 4) + struct.pack("<HH", 0, 32)
    msgs = [attr(b"sheets", vstr, space1(3), vdata), attr(b"type", fixed, scalar, b"EIGEN\0\0\0"),
            attr(b"shape", i32, space1(2), struct.pack("<ii", 122, 76611))]
    body = b"".join(struct.pack("<HHBBH", 0x0C, len(pad8(m)), 0, 0, 0) + pad8(m) for m in msgs)
    ohdr = bytes([1, 0]) + struct.pack("<HII", len(msgs), 1, len(body)) + b"\0" * 4 + body
    root_at = 96
    sb = (hdf5.SIGNATURE + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HHI", 4, 16, 0) + struct.pack("<QQQQ", 0, U, gcol_at + len(gcol), U) +
          struct.pack("<QQII", 0, root_at, 0, 0) + b"\0" * 16)
    assert len(sb) == root_at
    img = sb + ohdr
    assert len(img) <= gcol_at
    return img + b"\0" * (gcol_at - len(img)) + gcol


def test_handmade_image_string_list_attributes():
    ds, f = hdf5.read_netcdf4(_handmade_file_with_string_list_attributes())
    assert f.sb_version == 0 and not ds.variables and not ds.dims
    assert ds.attrs["sheets"] == ["greenland", "antarctica", ""]          # a Python list, not a comma-joined string
    assert ncio._strlist(ds.attrs["sheets"]) == ["greenland", "antarctica", ""]
    assert ds.attrs["type"] == "EIGEN"
    assert np.array_equal(ds.attrs["shape"], [122, 76611]) and ds.attrs["shape"].dtype == np.int32


def test_not_hdf5_and_truncated_files_fail_loudly(tmp_path):
    with pytest.raises(hdf5.H5Error, match="not an HDF5 file"):
        hdf5.File(b"CDF\x05" + b"\0" * 600)
    p = str(tmp_path / "h5.nc")
    open(p, "wb").write(hdf5.SIGNATURE + b"\x07" + b"\0" * 64)
    with pytest.raises(hdf5.H5Error, match="superblock version 7"):
        ncio.Dataset.read(p)
