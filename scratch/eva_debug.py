"""Which (variant, matrix, scale, correctA) build dies?  Progress goes to stdout unbuffered."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import icebin_amd
from icebin_amd import synthetic as syn
import faulthandler; faulthandler.enable()
for variant in ["sorted", "shuffled", "zero_area", "negative_area", "elev_class"]:
    kw = {}
    if variant == "shuffled": kw["order"] = "shuffled"
    if variant == "zero_area": kw["zero_area_every"] = 7
    g = syn.make_grids("g50", **kw)
    if variant == "negative_area":
        g["ex_area"] = g["ex_area"].copy(); g["ex_area"][::11] *= -1.0
    if variant == "elev_class": g["interp_style"] = 1
    em = syn.dome_elevmask(g)
    mm = icebin_amd.from_synthetic(g)
    rm = mm.regrid_matrices("greenland", em)
    for name in ("AvI", "IvA", "AvX", "XvA", "EvI", "IvE", "EvX", "XvE", "EvA", "AvE"):
        for scale in (True, False):
            for correctA in (True, False):
                print(variant, name, scale, correctA, flush=True)
                w = rm.matrix_d(name, scale=scale, correctA=correctA)
                print("   ok", w.built_fast(), w.nnz, flush=True)
