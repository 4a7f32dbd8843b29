"""Drop-in for the reference module ``pre/tsdf_for.py`` (the CPU triple loop), served by the GPU.

    tsdf_v, max_lenth, mid_point = tsdf_f(data, point_cloud)       # pre/tsdf_for.py:6-20
    tsdf_v = tsdf_cal(data, vox_ori, voxel_len, truncation)         # pre/tsdf_for.py:44-122

Contract kept: ``data = {'header': int32[6], 'depth': float32[N]}``; the grid placement comes from
the min/max of the point cloud that is passed in (pre/tsdf_for.py:9-16,23-41), in float32; the
volume is ``float64[3,32,32,32]`` indexed ``[c, x, y, z]`` (pre/tsdf_for.py:59,118-120).
The 32 x 32 x 32 Python loop (~0.1 s per frame) is replaced by the HIP kernel with the
caller-supplied placement (``tsdf_voxelize_grid_hip``).  There is no CPU fallback.
"""
from __future__ import annotations

import numpy as np
import torch


fFocal_msra = 241.42  # pre/tsdf_for.py:3


def max_min_point(point_cloud):
    """pre/tsdf_for.py:23-41: per-axis max/min as float32; zeros are dropped from z only."""
    pc = np.asarray(point_cloud)
    z = pc[:, 2]
    z = z[z != 0]
    point_max = np.array([pc[:, 0].max(), pc[:, 1].max(), z.max()], dtype=np.float32)
    point_min = np.array([pc[:, 0].min(), pc[:, 1].min(), z.min()], dtype=np.float32)
    return point_max, point_min


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("tsdf_f / tsdf_cal need a HIP device: this voxelizer has no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def tsdf_cal(data, vox_ori, voxel_len, truncation, voxel_res: int = 32):
    """pre/tsdf_for.py:44-122 on the GPU -> float64[3,R,R,R] in [c,x,y,z].  One page-locked block up (offsets, header,
    grid placement, depth), one back (the volume): tsdf_numba._Bufs."""
    from . import _lib
    from .tsdf_numba import _bufs
    header = np.ascontiguousarray(data["header"], dtype=np.int32).reshape(6)
    depth = np.ascontiguousarray(data["depth"], dtype=np.float32).reshape(-1)
    dev = _device()
    L = _lib.load()
    if not L.tsdf_resolution_supported(int(voxel_res)):
        raise ValueError(f"unsupported grid resolution {voxel_res} (multiple of 4 in 4..128)")
    b = _bufs(dev, depth.size, int(voxel_res))
    npx = depth.size
    b.h_off[0], b.h_off[1] = 0, npx
    b.h_hdr[:] = header
    b.h_grid[:] = 0.0
    b.h_grid[:3] = np.asarray(vox_ori, dtype=np.float32)
    b.h_grid[3] = np.float32(voxel_len)
    b.h_grid[4] = np.float32(truncation)
    b.h_depth[:npx] = depth
    nin = b.IN_HEAD + 4 * npx
    stream = torch.cuda.current_stream(dev)
    b.d_in[:nin].copy_(b.h_in[:nin], non_blocking=True)
    rc = L.tsdf_voxelize_grid_hip(b.p_depth, npx, b.p_off, b.p_hdr, 1, int(voxel_res), None, _lib.TSDF_LAYOUT_CXYZ,
                                  stream.cuda_stream, b.p_grid, b.p_tsdf, b.p_status)
    _lib.check(rc, "tsdf_voxelize_grid_hip")
    b.h_out.copy_(b.d_out, non_blocking=True)
    stream.synchronize()
    R = int(voxel_res)
    return b.h_out_np[:b.nvol].reshape(3, R, R, R).astype(np.float64)


def tsdf_f(data, point_cloud):
    """pre/tsdf_for.py:6-20: placement from the point cloud (float32 glue), then the volume."""
    voxel_res = 32
    point_max, point_min = max_min_point(point_cloud)
    mid_point = (point_max + point_min) / 2
    len_pixel = point_max - point_min
    max_lenth = np.max(len_pixel)
    voxel_len = max_lenth / voxel_res
    truncation = voxel_len * 3
    vox_ori = mid_point - max_lenth / 2 + voxel_len / 2
    tsdf_v = tsdf_cal(data, vox_ori, voxel_len, truncation, voxel_res)
    return tsdf_v, max_lenth, mid_point
