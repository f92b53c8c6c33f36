"""ctypes binding of libdpc_render.so (C ABI declared in include/dpc_render.h).

There is no CPU fallback: if the library is missing, or a tensor does not live on a HIP device, the
call raises.  The library is built in-tree by `make -C pytorch-unsup-pc_amd/csrc` (or
`python __graft_entry__.py`).
"""
import ctypes
import os

import torch

_CSRC = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "csrc"))
# DPC_RENDER_LIB points timing experiments at a variant build of the same library (tools/build_variant.sh); there is no
# other implementation to fall back to either way
LIB_PATH = os.environ.get("DPC_RENDER_LIB") or os.path.join(_CSRC, "libdpc_render.so")

ABI_VERSION = 13
DPC_MAX_TAPS = 63
DPC_MAX_POINTS = (1 << 20) - 1
DPC_SMALL_COLS = 12
COL_DQ, COL_DS, COL_DT, COL_DF = 0, 4, 5, 8
DPC_ERR_SHAPE = -2
DPC_ERR_TAPS = -3
DPC_ERR_LDS = -4
DPC_STATUS_BAD_INDEX = 1

# every symbol include/dpc_render.h declares (tests/test_abi.py checks the header against this list)
SYMBOLS = (
    "dpc_abi_version", "dpc_strerror", "dpc_mask_words_per_plane", "dpc_cells_bytes", "dpc_workspace_bytes", "dpc_check_grid", "dpc_locate",
    "dpc_project_fwd", "dpc_project_bwd", "dpc_project_loss_fwd", "dpc_project_loss_bwd", "dpc_transform_fwd", "dpc_transform_bwd",
    "dpc_splat_fwd", "dpc_splat_bwd", "dpc_smooth", "dpc_drc_fwd", "dpc_drc_bwd",
    "dpc_silhouette_loss", "dpc_point_dropout_indices", "dpc_point_dropout_indices_live", "dpc_schedule_update", "dpc_taps_bucket",
    "dpc_project_loss_step",
    "dpc_nearest_workspace_bytes", "dpc_point_cloud_distance", "dpc_profile_enable", "dpc_profile_disable", "dpc_profile_count", "dpc_profile_get", "dpc_profile_pair_overhead",
)


class DpcParams(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("N", ctypes.c_int32), ("D", ctypes.c_int32), ("H", ctypes.c_int32),
                ("W", ctypes.c_int32), ("taps_xy", ctypes.c_int32), ("taps_z", ctypes.c_int32),
                ("camera_distance", ctypes.c_float), ("focal_length", ctypes.c_float),
                ("clip_val", ctypes.c_float), ("max_depth", ctypes.c_float), ("point_replicas", ctypes.c_int32),
                ("N_src", ctypes.c_int32), ("point_index", ctypes.c_void_p), ("status", ctypes.c_void_p),
                ("n_live", ctypes.c_void_p), ("dev_taps_xy", ctypes.c_void_p), ("dev_taps_z", ctypes.c_void_p)]


class DpcError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        super().__init__("%s failed: %s (code %d)" % (where, strerror(code), code))


_lib = None


def lib():
    """The loaded library; raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "dpc.render: %s is missing -- the HIP extension has not been built. Run `make -C %s` "
                "(needs /opt/rocm/bin/hipcc) or `python -c 'import __graft_entry__ as g; g.build()'`. "
                "There is no CPU fallback." % (LIB_PATH, _CSRC))
        L = ctypes.CDLL(LIB_PATH)
        vp, pp = ctypes.c_void_p, ctypes.POINTER(DpcParams)
        L.dpc_abi_version.restype = ctypes.c_int
        L.dpc_strerror.restype = ctypes.c_char_p
        L.dpc_strerror.argtypes = [ctypes.c_int]
        L.dpc_mask_words_per_plane.restype = ctypes.c_size_t
        L.dpc_mask_words_per_plane.argtypes = [pp]
        L.dpc_cells_bytes.restype = ctypes.c_size_t
        L.dpc_cells_bytes.argtypes = [pp]
        L.dpc_workspace_bytes.restype = ctypes.c_size_t
        L.dpc_workspace_bytes.argtypes = [pp]
        L.dpc_check_grid.restype = ctypes.c_int
        L.dpc_check_grid.argtypes = [pp, ctypes.c_int]
        for name, nptr in (("dpc_project_fwd", 16), ("dpc_project_bwd", 17), ("dpc_transform_fwd", 6),
                           ("dpc_transform_bwd", 8), ("dpc_drc_fwd", 5), ("dpc_drc_bwd", 6), ("dpc_locate", 7)):
            fn = getattr(L, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [pp] + [vp] * nptr
        L.dpc_project_loss_fwd.restype = ctypes.c_int
        L.dpc_project_loss_fwd.argtypes = [pp] + [vp] * 8 + [ctypes.c_int] + [vp] * 12 + [ctypes.POINTER(ctypes.c_int), vp]
        L.dpc_project_loss_bwd.restype = ctypes.c_int
        L.dpc_project_loss_bwd.argtypes = [pp] + [vp] * 13 + [ctypes.c_int] + [vp] * 2 + [ctypes.c_int] + [vp] * 4
        L.dpc_splat_fwd.restype = ctypes.c_int
        L.dpc_splat_fwd.argtypes = [pp, vp, ctypes.c_int, vp, vp, vp]
        L.dpc_splat_bwd.restype = ctypes.c_int
        L.dpc_splat_bwd.argtypes = [pp, vp, ctypes.c_int, vp, vp, vp]
        L.dpc_profile_enable.restype = ctypes.c_int
        L.dpc_profile_enable.argtypes = [ctypes.c_int]
        L.dpc_profile_disable.restype = ctypes.c_int
        L.dpc_profile_count.restype = ctypes.c_int
        L.dpc_profile_get.restype = ctypes.c_int
        L.dpc_profile_get.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_float)]
        L.dpc_profile_pair_overhead.restype = ctypes.c_int
        L.dpc_profile_pair_overhead.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
        L.dpc_silhouette_loss.restype = ctypes.c_int
        L.dpc_silhouette_loss.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
        L.dpc_nearest_workspace_bytes.restype = ctypes.c_size_t
        L.dpc_nearest_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.dpc_point_cloud_distance.restype = ctypes.c_int
        L.dpc_point_cloud_distance.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp]
        L.dpc_smooth.restype = ctypes.c_int
        L.dpc_smooth.argtypes = [pp, vp, vp, ctypes.c_int, vp, vp, vp, vp]
        L.dpc_point_dropout_indices.restype = ctypes.c_int
        L.dpc_point_dropout_indices.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp]
        L.dpc_point_dropout_indices_live.restype = ctypes.c_int
        L.dpc_point_dropout_indices_live.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
        L.dpc_schedule_update.restype = ctypes.c_int
        L.dpc_schedule_update.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
        L.dpc_taps_bucket.restype = ctypes.c_int
        L.dpc_taps_bucket.argtypes = [vp, ctypes.c_int]
        L.dpc_project_loss_step.restype = ctypes.c_int
        L.dpc_project_loss_step.argtypes = [pp] + [vp] * 8 + [ctypes.c_int] + [vp] * 15
        if L.dpc_abi_version() != ABI_VERSION:
            raise RuntimeError("dpc.render: libdpc_render.so ABI %d, expected %d -- rebuild it (make -C %s)"
                               % (L.dpc_abi_version(), ABI_VERSION, _CSRC))
        _lib = L
    return _lib


def strerror(code):
    return lib().dpc_strerror(code).decode()


def check(code, where):
    if code != 0:
        raise DpcError(code, where)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def host_floats(arr):
    """float32 host array -> (ctypes pointer, keep-alive)."""
    import numpy as np

    a = np.ascontiguousarray(arr, dtype=np.float32)
    return a.ctypes.data_as(ctypes.c_void_p), a


def stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_device(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("dpc.render runs on MI355X only: got a %s tensor; move inputs to a HIP device "
                               "(there is no CPU path in this package)" % t.device)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("dpc.render: tensors on different devices (%s vs %s)" % (dev, t.device))
    return dev


def event_pair_overhead_ms(device, pairs=200):
    """What an empty event pair reads on the current stream (the floor in every profile_kernels figure)."""
    ms = ctypes.c_float()
    torch.cuda.synchronize(device)
    check(lib().dpc_profile_pair_overhead(stream_ptr(device), pairs, ctypes.byref(ms)), "dpc_profile_pair_overhead")
    return ms.value


def profile_kernels(fn, device, capacity=4096):
    """Run fn() with the library's per-kernel event timing on; returns {kernel_name: [ms, ...]}."""
    L = lib()
    check(L.dpc_profile_enable(capacity), "dpc_profile_enable")
    try:
        fn()
        torch.cuda.synchronize(device)
    finally:
        L.dpc_profile_disable()
    out = {}
    name, ms = ctypes.c_char_p(), ctypes.c_float()
    for i in range(L.dpc_profile_count()):
        check(L.dpc_profile_get(i, ctypes.byref(name), ctypes.byref(ms)), "dpc_profile_get")
        out.setdefault(name.value.decode(), []).append(ms.value)
    return out
