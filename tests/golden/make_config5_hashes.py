"""Writes tests/golden/config5_oracle_hashes.json: SHA-256 of every array of the ORACLE's matrices at BASELINE config 5's
full size (Antarctica 1 km <-> 1/2 deg, `a1h`; Greenland 1 km <-> 1/2 deg, `g1h`), run in the build container.

Why hashes: one oracle build of the Antarctic sheet takes 16-30 s and 6 GB of host memory on one core; the whole table
(10 matrices x the (scale, correctA) branches) is minutes of CPU work the GPU suite should not repeat on every run.  The
GPU test (`test_config5_full_size_assembly_is_the_oracle_bitwise`) builds the same matrices through the C-ABI on the same
seeded inputs and compares the hashes: equal SHA-256 over dims / row / col / val / wM / Mw == bit-exact parity with
`oracle/icebin_oracle.c` (which restates RegridMatrices_Dynamic.cpp:50-332).  Inputs are icebin_amd.synthetic (seeded,
deterministic); `inputs` in the JSON pins them too, so a changed generator fails loudly instead of comparing apples to pears.

Run:  python tests/golden/make_config5_hashes.py [config ...]      (about 15 minutes for both sheets)
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from icebin_amd import synthetic as syn          # noqa: E402  (inputs only: no regridding arithmetic)
from oracle import oracle as orc                 # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "config5_oracle_hashes.json")
ALL = ("AvI", "IvA", "AvX", "XvA", "EvI", "IvE", "EvX", "XvE", "EvA", "AvE")
BRANCHES = ((True, False), (True, True), (False, False), (False, True))         # (scale, correctA)


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.view(np.uint8).reshape(-1).data).hexdigest()


def weighted_hashes(nrow, ncol, nnz, dim0, dim1, row, col, val, wM, Mw):
    """The one definition of what is hashed, shared with the GPU test (imported from there)."""
    return dict(nrow=int(nrow), ncol=int(ncol), nnz=int(nnz),
                dim0=sha(np.asarray(dim0, np.int64)), dim1=sha(np.asarray(dim1, np.int64)),
                row=sha(np.asarray(row, np.int32)), col=sha(np.asarray(col, np.int32)),
                val=sha(np.asarray(val, np.float64)), wM=sha(np.asarray(wM, np.float64)), Mw=sha(np.asarray(Mw, np.float64)))


def input_hashes(g, em):
    return dict(nI=int(g["nI"]), nX=int(len(g["ex_area"])), ex_indices=sha(np.asarray(g["ex_indices"], np.int32)),
                ex_area=sha(g["ex_area"]), elevmask=sha(em), A_native_area=sha(g["A_native_area"]), hcdefs=sha(g["hcdefs"]))


def main(configs):
    doc = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for config in configs:
        g = syn.make_grids(config)
        em = syn.dome_elevmask(g)
        rg = orc.Regridder(g)
        ent = doc.setdefault(config, {})
        ent["inputs"] = input_hashes(g, em)
        mats = ent.setdefault("matrices", {})
        for name in ALL:
            for scale, correctA in BRANCHES:
                key = "%s scale=%d correctA=%d" % (name, scale, correctA)
                t0 = time.time()
                o = rg.matrix_d(name, em, scale=scale, correctA=correctA)
                mats[key] = weighted_hashes(o.nrow, o.ncol, o.nnz, o.dims[0], o.dims[1], o.row, o.col, o.val, o.wM, o.Mw)
                mats[key]["oracle_seconds"] = round(time.time() - t0, 1)
                print(config, key, mats[key]["nnz"], "%.1f s" % (time.time() - t0), flush=True)
                del o
                json.dump(doc, open(OUT, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1:] or ["g1h", "a1h"])
