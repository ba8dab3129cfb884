"""ctypes binding of libicebin_hip.so (include/icebin_hip.h).  No torch types here."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ICEBIN_HIP_LIB") or os.path.join(_HERE, "lib", "libicebin_hip.so")   # override: experiment builds only

IBH_OK, IBH_EINVAL, IBH_ENODEVICE, IBH_EHIP, IBH_ERANGE, IBH_ENOTIMPL, IBH_ENOKEY = 0, -1, -2, -3, -4, -5, -6


class IcebinHipError(RuntimeError):
    """Raised for every non-zero status, like Cython's `except +` turns the reference's
    everytrace::Exception into RuntimeError (pylib/cicebin.pxd:76-118)."""

    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class RegridderDesc(C.Structure):
    _fields_ = [
        ("nX", C.c_int64), ("ex_indices", C.c_void_p), ("ex_area", C.c_void_p), ("nI", C.c_int64),
        ("nA", C.c_int64), ("nA_dense", C.c_int32), ("A_to_sparse", C.c_void_p),
        ("A_native_area", C.c_void_p), ("A_proj_area", C.c_void_p),
        ("nhc", C.c_int32), ("hcdefs", C.c_void_p), ("hc_stride_A", C.c_int64), ("hc_stride_HC", C.c_int64),
        ("interp_style", C.c_int32), ("I_centroid_xy", C.c_void_p),
    ]


class ExgridDesc(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("xedges", C.c_void_p), ("yedges", C.c_void_p), ("x_fastest", C.c_int32),
                ("npoly", C.c_int32), ("polyptr", C.c_void_p), ("vx", C.c_void_p), ("vy", C.c_void_p), ("iA", C.c_void_p)]


class DeviceView(C.Structure):
    _fields_ = [("nrow", C.c_int32), ("ncol", C.c_int32), ("nnz", C.c_int64), ("rowptr", C.c_void_p),
                ("colind", C.c_void_p), ("val", C.c_void_p), ("wM", C.c_void_p), ("Mw", C.c_void_p)]


_SIGS = {
    "ibh_version": (C.c_int, []),
    "ibh_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "ibh_set_device": (C.c_int, [C.c_int]),
    "ibh_sparse_set_create": (C.c_int, [C.c_int64, C.POINTER(C.c_void_p)]),
    "ibh_sparse_set_create_identity": (C.c_int, [C.c_int64, C.POINTER(C.c_void_p)]),
    "ibh_sparse_set_from_array": (C.c_int, [C.c_int64, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "ibh_sparse_set_destroy": (C.c_int, [C.c_void_p]),
    "ibh_sparse_set_sparse_extent": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "ibh_sparse_set_dense_extent": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "ibh_sparse_set_to_sparse": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ibh_sparse_set_to_dense": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int32)]),
    "ibh_sparse_set_add_dense": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int32)]),
    "ibh_exgrid_generate": (C.c_int, [C.POINTER(ExgridDesc), C.POINTER(C.c_void_p)]),
    "ibh_exgrid_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "ibh_exgrid_get": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ibh_exgrid_destroy": (C.c_int, [C.c_void_p]),
    "ibh_regridder_create": (C.c_int, [C.POINTER(RegridderDesc), C.POINTER(C.c_void_p)]),
    "ibh_regridder_destroy": (C.c_int, [C.c_void_p]),
    "ibh_regridder_sizes": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "ibh_regridder_wA": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_void_p]),
    "ibh_regrid_matrices_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                            C.POINTER(C.c_void_p)]),
    "ibh_regrid_matrices_create_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                                   C.c_void_p, C.POINTER(C.c_void_p)]),
    "ibh_regrid_matrices_destroy": (C.c_int, [C.c_void_p]),
    "ibh_regrid_matrices_matrix_d": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                              C.c_void_p, C.POINTER(C.c_void_p)]),
    "ibh_regrid_matrices_matrix_batch": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "ibh_regrid_matrices_matrix": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "ibh_weighted_from_coo": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "ibh_weighted_from_csr": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "ibh_e1ve0_compute": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]),
    "ibh_weighted_destroy": (C.c_int, [C.c_void_p]),
    "ibh_weighted_shape": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "ibh_weighted_flags": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ibh_weighted_dim": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "ibh_weighted_dim_to_sparse": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "ibh_weighted_get_wM": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ibh_weighted_get_Mw": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ibh_weighted_get_coo": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ibh_weighted_get_csr": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ibh_weighted_apply_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int64,
                                         C.c_double, C.c_int]),
    "ibh_weighted_apply_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int64,
                                           C.c_double, C.c_int, C.c_void_p]),
    "ibh_weighted_apply_many_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p,
                                                C.c_int64, C.c_double, C.c_int, C.c_void_p]),
    "ibh_weighted_matvec_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.c_int,
                                            C.c_void_p]),
    "ibh_coo_matvec": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                C.c_void_p]),
    "ibh_weighted_reserve": (C.c_int, [C.c_void_p, C.c_int32]),
    "ibh_weighted_prepare": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "ibh_weighted_pair_prepare": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "ibh_weighted_apply_chain_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int64,
                                               C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_double, C.c_void_p]),
    "ibh_weighted_apply_pair_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int64,
                                               C.c_void_p, C.c_int64, C.c_double, C.c_void_p]),
    "ibh_comm_unique_id": (C.c_int, [C.c_char_p]),
    "ibh_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]),
    "ibh_comm_create_custom": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "ibh_comm_destroy": (C.c_int, [C.c_void_p]),
    "ibh_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ibh_comm_wait": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ibh_comm_set_custom_gatherv": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ibh_comm_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ibh_comm_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "ibh_regrid_matrices_matrix_d_sharded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                       C.POINTER(C.c_void_p)]),
    "ibh_weighted_apply_many_sharded_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p,
                                                        C.c_int64, C.c_double, C.c_void_p]),
    "ibh_weighted_apply_sharded_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int64,
                                                   C.c_double, C.c_int32, C.c_void_p]),
    "ibh_weighted_apply_transformed_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p,
                                                       C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_double,
                                                       C.c_void_p]),
    "ibh_weighted_apply_weight_host": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]),
    "ibh_weighted_device_view_get": (C.c_int, [C.c_void_p, C.POINTER(DeviceView)]),
    "ibh_weighted_set_kernel": (C.c_int, [C.c_void_p, C.c_char_p]),
    "ibh_weighted_last_kernel": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "ibh_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "ibh_event_destroy": (C.c_int, [C.c_void_p]),
    "ibh_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "ibh_set_launch_events": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ibh_weighted_built_fast": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "ibh_weighted_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "ibh_weighted_last_launch": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "ibh_set_tuning": (C.c_int, [C.c_char_p, C.c_int]),
    "ibh_release_cached_memory": (C.c_int, []),
    "ibh_selftest_sort": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int)]),
}

_lib = None


def _preload_hip_runtime():
    """One HIP/HSA runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so.7 /
    libhsa-runtime64; if this library pulled in /opt/rocm's copy as well, whichever runtime came
    second would find no GPU.  So when torch is installed, load ITS runtime first (without importing
    torch): the dynamic linker then binds libicebin_hip.so's libamdhip64.so.7 to the copy already
    loaded, and a later `import torch` shares it.  ICEBIN_HIP_RUNTIME=system skips this."""
    if os.environ.get("ICEBIN_HIP_RUNTIME", "torch") != "torch":
        return
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return      # torch already loaded its runtime
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    rt = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(rt):
        C.CDLL(rt, mode=C.RTLD_GLOBAL)


def lib():
    """Load the HIP library; fails loudly (no fallback) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "icebin_amd: %s is missing. Build it with `python -m icebin_amd.build` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback." % LIB_PATH)
        _preload_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.ibh_last_error.restype = C.c_char_p
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def exported_symbols():
    return ["ibh_last_error"] + list(_SIGS)


def check(rc):
    if rc != 0:
        raise IcebinHipError(rc, lib().ibh_last_error().decode(errors="replace"))


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def device_count():
    n = C.c_int(0)
    check(lib().ibh_device_count(C.byref(n)))
    return n.value


def destroy(fn_name, handle):
    """Handle destructor usable from __del__, also during interpreter shutdown."""
    L = _lib
    if L is None or handle is None or not getattr(handle, "value", None):
        return
    try:
        getattr(L, fn_name)(handle)
    except Exception:
        pass
