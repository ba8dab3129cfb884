# The reference's pylib/icebin/__init__.py:18 does exactly this: the package is the compiled extension.
try:    # one HIP runtime per process (see icebin_amd/_capi.py): load torch's copy first when torch is installed
    from icebin_amd import _capi as _c
    _c._preload_hip_runtime()
except ImportError:
    pass
from _icebin import *  # noqa: F401,F403,E402
from _icebin import GCMRegridder, RegridMatrices, linear_Weighted, coo_multiply  # noqa: F401,E402
