"""Derive the config-1 elevation/mask fixture from the reference's own test data.

Reads /root/reference/tests/test_conserv/elev_mask.cdl (an ncdump text dump of
PISM 20 km Greenland `mask`, `thk`, `topg`, shape [time=1, x=76, y=141]) and
stores the three arrays, bit-for-bit as printed, in elev_mask_g20.npz.  This is
DATA held by the reference's test, not source code.
Run once in the build container; the .npz is committed, the CDL is not.
"""
import re
import sys
import numpy as np

SRC = "/root/reference/tests/test_conserv/elev_mask.cdl"


def parse(path):
    text = open(path).read()
    data = text[text.index("data:"):]
    out = {}
    for name, dtype in (("mask", np.int8), ("thk", np.float64), ("topg", np.float64)):
        m = re.search(r"\b%s =\s*(.*?);" % name, data, re.S)
        toks = [t for t in re.split(r"[,\s]+", m.group(1)) if t]
        out[name] = np.array([float(t) for t in toks]).astype(dtype).reshape(76, 141)
    return out


if __name__ == "__main__":
    arrs = parse(SRC)
    vals, cnt = np.unique(arrs["mask"], return_counts=True)
    print(dict(zip(vals.tolist(), cnt.tolist())))
    np.savez_compressed(sys.argv[1] if len(sys.argv) > 1 else "tests/golden/elev_mask_g20.npz", **arrs)
