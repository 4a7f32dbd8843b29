"""ctypes binding of libtsdf_hip.so (the C ABI declared in include/tsdf.h).

There is deliberately NO fallback: if the HIP library is missing or fails to load,
importing the voxelizer raises.  The product path never routes through a CPU
implementation (the oracle under oracle/ is test infrastructure only).
"""
from __future__ import annotations

import contextlib
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_DEFAULT_LIB = os.path.join(_HERE, "libtsdf_hip.so")


def _lib_path() -> str:
    """The product binary.  TSDF_HIP_LIB may name another build of the same ABI (diagnostic variants live in
    <repo>/build/), but only together with TSDF_ALLOW_LIB_OVERRIDE=1: a stale variable must not silently swap the
    library the tests and the bench believe they are measuring."""
    over = os.environ.get("TSDF_HIP_LIB")
    if not over or os.path.abspath(over) == _DEFAULT_LIB:
        return _DEFAULT_LIB
    if os.environ.get("TSDF_ALLOW_LIB_OVERRIDE") != "1":
        raise ImportError(f"TSDF_HIP_LIB={over} asks for a library other than {_DEFAULT_LIB}; set "
                          "TSDF_ALLOW_LIB_OVERRIDE=1 as well if that is intended (experiments only)")
    return over


LIB_PATH = _lib_path()

TSDF_LAYOUT_CZYX = 0
TSDF_LAYOUT_CXYZ = 1
LAYOUTS = {"czyx": TSDF_LAYOUT_CZYX, "cxyz": TSDF_LAYOUT_CXYZ}

TSDF_OK = 0
TSDF_FRAME_OK = 0
TSDF_FRAME_DEGENERATE = 1
TSDF_FRAME_BAD_HEADER = 2


class TsdfError(RuntimeError):
    """A tsdf_* entry point returned a negative status."""


class TsdfCam(ctypes.Structure):
    """``tsdf_cam`` of include/tsdf.h (defaults: pre/tsdf_numba.py:8-10)."""

    _fields_ = [
        ("focal", ctypes.c_double),
        ("cx", ctypes.c_double),
        ("cy", ctypes.c_double),
        ("invalid_eps", ctypes.c_float),
        ("trunc_voxels", ctypes.c_float),
    ]


class TsdfLabels(ctypes.Structure):
    """``tsdf_labels`` of include/tsdf.h: device pointers for the fused label normalisation."""

    _fields_ = [
        ("d_gt", ctypes.c_void_p),
        ("n_joints", ctypes.c_int),
        ("clamp", ctypes.c_int),
        ("d_out_gt_nor", ctypes.c_void_p),
        ("d_out_gt_aug", ctypes.c_void_p),
    ]


ABI_VERSION = 7
INLINE_INDEX_MAX = 32   # TSDF_INLINE_INDEX_MAX of include/tsdf.h

# The debug build of the same sources (make -C csrc debug: -DTSDF_DEBUG_HOOKS): everything the product exports plus the
# test hooks of include/tsdf_debug.h.  Never loaded unless somebody asks for a hook.
DEBUG_LIB_PATH = os.path.join(os.path.dirname(_HERE), "build", "libtsdf_hip_debug.so")

_lib = None
_debug_lib = None


def _bind(L, path: str):
    """Declare the argument / result types of every entry point of include/tsdf.h on a loaded library."""
    vp = ctypes.c_void_p
    cam_p = ctypes.POINTER(TsdfCam)
    L.tsdf_version.restype = ctypes.c_int
    L.tsdf_version.argtypes = []
    if L.tsdf_version() != ABI_VERSION:
        raise ImportError(f"{path} has ABI version {L.tsdf_version()}, this package needs {ABI_VERSION}: rebuild it")
    L.tsdf_strerror.restype = ctypes.c_char_p
    L.tsdf_strerror.argtypes = [ctypes.c_int]
    L.tsdf_resolution_supported.restype = ctypes.c_int
    L.tsdf_resolution_supported.argtypes = [ctypes.c_int]
    L.tsdf_default_cam.restype = None
    L.tsdf_default_cam.argtypes = [cam_p]
    L.tsdf_voxelize_hip.restype = ctypes.c_int
    L.tsdf_voxelize_hip.argtypes = [vp, ctypes.c_int64, vp, vp, ctypes.c_int, ctypes.c_int, cam_p, ctypes.c_int, vp,
                                    vp, vp, vp, vp]
    L.tsdf_voxelize_grid_hip.restype = ctypes.c_int
    L.tsdf_voxelize_grid_hip.argtypes = [vp, ctypes.c_int64, vp, vp, ctypes.c_int, ctypes.c_int, cam_p, ctypes.c_int, vp,
                                         vp, vp, vp]
    L.tsdf_voxelize_aug_hip.restype = ctypes.c_int
    L.tsdf_voxelize_aug_hip.argtypes = [vp, ctypes.c_int64, vp, vp, ctypes.c_int, ctypes.c_int, cam_p, ctypes.c_int, vp,
                                        vp, vp, vp, vp, vp]
    L.tsdf_aabb_hip.restype = ctypes.c_int
    L.tsdf_aabb_hip.argtypes = [vp, ctypes.c_int64, vp, vp, ctypes.c_int, ctypes.c_int, cam_p, vp, vp, vp, vp, vp]
    lab_p = ctypes.POINTER(TsdfLabels)
    L.tsdf_voxelize_labels_hip.restype = ctypes.c_int
    L.tsdf_voxelize_labels_hip.argtypes = L.tsdf_voxelize_hip.argtypes + [lab_p]
    L.tsdf_voxelize_aug_labels_hip.restype = ctypes.c_int
    L.tsdf_voxelize_aug_labels_hip.argtypes = L.tsdf_voxelize_aug_hip.argtypes + [lab_p]
    L.tsdf_voxelize_indexed_hip.restype = ctypes.c_int
    L.tsdf_voxelize_indexed_hip.argtypes = [vp, ctypes.c_int64, vp, vp, ctypes.c_int64, vp, ctypes.c_int, ctypes.c_int,
                                            cam_p, ctypes.c_int, vp, vp, vp, vp, vp, lab_p]
    L.tsdf_voxelize_indexed_host_hip.restype = ctypes.c_int
    L.tsdf_voxelize_indexed_host_hip.argtypes = L.tsdf_voxelize_indexed_hip.argtypes
    L.tsdf_voxelize_indexed_aug_hip.restype = ctypes.c_int
    L.tsdf_voxelize_indexed_aug_hip.argtypes = [vp, ctypes.c_int64, vp, vp, ctypes.c_int64, vp, ctypes.c_int, ctypes.c_int,
                                                cam_p, ctypes.c_int, vp, vp, vp, vp, vp, vp, lab_p]
    L.tsdf_host_gather_frames.restype = ctypes.c_int
    L.tsdf_host_gather_frames.argtypes = [vp, vp, ctypes.c_int64, vp, ctypes.c_int64, vp, ctypes.c_int64, vp, ctypes.c_int]
    L.tsdf_host_gather_frames_n.restype = ctypes.c_int
    L.tsdf_host_gather_frames_n.argtypes = [vp, ctypes.c_int64, vp, ctypes.c_int64, vp, ctypes.c_int64, vp, ctypes.c_int64, vp,
                                            ctypes.c_int]
    L.tsdf_normalize_joints_hip.restype = ctypes.c_int
    L.tsdf_normalize_joints_hip.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]
    L.tsdf_denormalize_joints_hip.restype = ctypes.c_int
    L.tsdf_denormalize_joints_hip.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, vp, vp]
    L.tsdf_stream_release.restype = ctypes.c_int
    L.tsdf_stream_release.argtypes = [vp]
    L.tsdf_describe_launch.restype = ctypes.c_int
    L.tsdf_describe_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
    return L


def load():
    """Load libtsdf_hip.so once; raise loudly if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C handposeestimation-with-3d-cnns_amd/csrc`. There is no CPU fallback."
        )
    _lib = _bind(ctypes.CDLL(LIB_PATH), LIB_PATH)
    return _lib


def load_debug():
    """The debug build (include/tsdf_debug.h): the whole ABI plus tsdf_debug_pixmap_hip / tsdf_debug_set_queue_word, and
    TSDF_XCHG_POLLS read from the environment.  A library image of its own, with its own device-side state."""
    global _debug_lib
    if _debug_lib is not None:
        return _debug_lib
    if not os.path.exists(DEBUG_LIB_PATH):
        raise ImportError(f"{DEBUG_LIB_PATH} not found: build it with `make -C handposeestimation-with-3d-cnns_amd/csrc debug` "
                          "(__graft_entry__.build() does)")
    L = _bind(ctypes.CDLL(DEBUG_LIB_PATH), DEBUG_LIB_PATH)
    vp = ctypes.c_void_p
    L.tsdf_debug_pixmap_hip.restype = ctypes.c_int
    L.tsdf_debug_pixmap_hip.argtypes = [vp, ctypes.c_int64, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(TsdfCam),
                                        ctypes.c_int, vp, vp, vp, vp, vp]
    L.tsdf_debug_set_queue_word.restype = ctypes.c_int
    L.tsdf_debug_set_queue_word.argtypes = [vp, ctypes.c_uint64]
    _debug_lib = L
    return L


@contextlib.contextmanager
def using_debug_library():
    """Inside the block every call of this package goes through the debug build instead of the product (tests that need a
    hook and the launches it acts on in ONE library image).  Not thread-safe; objects that cached the library at
    construction (dataset loaders) keep theirs."""
    global _lib
    prev = _lib
    _lib = load_debug()
    try:
        yield _lib
    finally:
        _lib = prev


def check(status: int, what: str):
    if status != TSDF_OK:
        msg = load().tsdf_strerror(status).decode()
        raise TsdfError(f"{what}: {msg} (status {status})")


def default_cam() -> TsdfCam:
    c = TsdfCam()
    load().tsdf_default_cam(ctypes.byref(c))
    return c
