"""Writes tests/golden/nc4_example1.json: what icebin_amd.hdf5 reads out of the reference's own NetCDF-4 example files
(/root/reference/examples/example1/*.nc -- ModelE diagnostics on the 2 x 2.5 degree grid with 40 elevation classes, the
inputs of examples/example1/runme.py).  Runs only where /root/reference exists (the build container); the files themselves
are not copied -- the fixture holds dimensions, variables, shapes, attributes and a SHA-256 / min / max / float64 sum per
variable, so the test can tell a reader regression from a changed file.

usage: python tests/golden/make_nc4_fixture.py
"""
import glob
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from icebin_amd import hdf5  # noqa: E402

SRC = "/root/reference/examples/example1"


def describe(path):
    ds, f = hdf5.read_netcdf4(path)
    raw = open(path, "rb").read()
    out = {"bytes": len(raw), "sha256": hashlib.sha256(raw).hexdigest(), "superblock": f.sb_version,
           "lookup3_blocks_verified": f.checked["lookup3"], "dims": dict(ds.dims),
           "attrs": {k: (v if isinstance(v, (str, list)) else np.asarray(v).tolist()) for k, v in ds.attrs.items()},
           "variables": {}}
    for name, v in ds.variables.items():
        a = np.ascontiguousarray(v.data)
        le = a.astype(a.dtype.newbyteorder("<"))
        fin = a[np.isfinite(a)] if a.dtype.kind == "f" else a
        out["variables"][name] = {
            "dims": list(v.dims), "dtype": a.dtype.name, "shape": list(a.shape),
            "sha256": hashlib.sha256(le.tobytes()).hexdigest(),
            "min": float(fin.min()), "max": float(fin.max()), "sum": float(fin.astype(np.float64).sum()),
            "attrs": {k: (x if isinstance(x, (str, list)) else np.asarray(x).tolist()) for k, x in v.attrs.items()}}
    return out


if __name__ == "__main__":
    files = sorted(glob.glob(os.path.join(SRC, "*.nc")))
    if not files:
        sys.exit("no reference example files under %s" % SRC)
    fixture = {os.path.basename(p): describe(p) for p in files}
    with open(os.path.join(HERE, "nc4_example1.json"), "w") as fh:
        json.dump(fixture, fh, indent=1, sort_keys=True)
    print("wrote", len(fixture), "files,", sum(len(v["variables"]) for v in fixture.values()), "variables")
