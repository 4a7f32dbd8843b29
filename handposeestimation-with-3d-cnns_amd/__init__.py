"""MI355X-native projective-TSDF voxelizer (drop-in for the reference's
pre/tsdf_numba.py / pre/tsdf_for.py / pre/process.py voxelization path).

The compute path is the hand-written HIP library ``libtsdf_hip.so`` (csrc/tsdf_hip.hip)
behind the C ABI of include/tsdf.h.  Nothing here falls back to a CPU implementation.

Batched API (torch tensors on the GPU):  voxelize, voxelize_grid, voxelize_aug (fused 3-D augmentation), aabb
Reference-signature shims:               tsdf_numba.cal_tsdf_cuda, tsdf_for.tsdf_f / tsdf_cal,
                                         process.DataProcess
Host side:                               packing (MSRA .bin reader / batch packer), shard, synth,
                                         dataset (on-the-fly MSRADepthDataset / VoxelLoader / ResidentLoader, label normalisation)
"""
from . import _lib  # noqa: F401
from ._lib import TsdfCam, TsdfError, default_cam  # noqa: F401
from .voxelize import (AabbBatch, TsdfBatch, aabb, denormalize_joints, normalize_joints, release_stream,  # noqa: F401
                       voxel_pixels, voxelize, voxelize_aug, voxelize_grid, voxelize_indexed, voxelize_labels)
from . import augment, dataset, export, packing, shard, synth  # noqa: F401
from .dataset import MSRA_Dataset, MSRADepthDataset, ResidentLoader, VoxelBatch, VoxelLoader  # noqa: F401
from .tsdf_numba import cal_tsdf_cuda  # noqa: F401
from .tsdf_for import tsdf_cal, tsdf_f  # noqa: F401
from .process import DataProcess  # noqa: F401

__all__ = ["voxelize", "voxelize_labels", "voxelize_indexed", "ResidentLoader", "voxel_pixels", "release_stream", "voxelize_grid", "voxelize_aug", "augment", "aabb", "TsdfBatch", "AabbBatch", "TsdfCam", "TsdfError",
           "default_cam", "cal_tsdf_cuda", "tsdf_f", "tsdf_cal", "DataProcess", "packing", "shard",
           "synth", "dataset", "MSRADepthDataset", "MSRA_Dataset", "VoxelLoader", "VoxelBatch", "normalize_joints", "denormalize_joints"]
