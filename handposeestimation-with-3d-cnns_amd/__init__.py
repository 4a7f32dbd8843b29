"""MI355X-native projective-TSDF voxelizer (drop-in for the reference's
pre/tsdf_numba.py / pre/process.py voxelization path).

The compute path is the hand-written HIP library ``libtsdf_hip.so`` (csrc/tsdf_hip.hip)
behind the C ABI of include/tsdf.h.  Nothing here falls back to a CPU implementation.
"""
from . import _lib  # noqa: F401
from ._lib import TsdfCam, TsdfError, default_cam  # noqa: F401
from .voxelize import AabbBatch, TsdfBatch, aabb, voxelize  # noqa: F401

__all__ = ["voxelize", "aabb", "TsdfBatch", "AabbBatch", "TsdfCam", "TsdfError", "default_cam"]
