"""icebin_amd: MI355X-native implementation of IceBin's conservative-regridding hot path.

Python surface mirrors pylib/_icebin.pyx (GCMRegridder, RegridMatrices) and
ibmisc.linear_Weighted; all compute goes through libicebin_hip.so (include/icebin_hip.h).
"""
from ._capi import IcebinHipError, device_count  # noqa: F401
from .linear import SparseSet, compute_E1vE0c, coo_multiply, linear_Weighted, nc_read_weighted, set_tuning  # noqa: F401
from .regrid import GCMRegridder, RegridMatrices, from_synthetic  # noqa: F401
