"""Batched on-device voxelization on torch tensors (thin host layer over the C ABI).

``voxelize`` is the batched form of the reference's ``cal_tsdf_cuda``
(pre/tsdf_numba.py:119-161): depth crops in, ``(tsdf, max_l, mid_p)`` out — the
triple ``3D_CNN/dataset.py:73-79`` hands to the network — but for n frames in one
fused HIP launch on the caller's stream, with the results left on the GPU.
torch is used for device memory and the stream handle only.
"""
from __future__ import annotations

import ctypes
from typing import NamedTuple, Optional

import torch

from . import _lib


class TsdfBatch(NamedTuple):
    tsdf: torch.Tensor    # float32[n,3,R,R,R]
    max_l: torch.Tensor   # float32[n]
    mid_p: torch.Tensor   # float32[n,3]
    status: torch.Tensor  # int32[n]  (_lib.TSDF_FRAME_*)


def _dev_check(name, t, dtype, device=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU (there is no CPU path); got device {t.device}")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if device is not None and t.device != device:
        raise ValueError(f"{name} is on {t.device}, expected {device}")


def voxelize(depth: torch.Tensor, offsets: torch.Tensor, headers: torch.Tensor, res: int = 32,
             layout: str = "czyx", cam: Optional[_lib.TsdfCam] = None,
             out: Optional[TsdfBatch] = None) -> TsdfBatch:
    """Voxelize n packed depth crops on the GPU.

    depth   float32[sum N_i]  crops packed back to back (payload of the MSRA .bin files)
    offsets int64[n+1]        element offsets of each crop in ``depth``
    headers int32[n,6]        W, H, left, top, right, bottom per frame
    res     grid resolution R (reference: 32)
    layout  "czyx" (numba kernel layout, default) or "cxyz" (CPU-loop layout)
    out     optional preallocated TsdfBatch to write into (no allocation, graph-capturable)

    Enqueues on ``torch.cuda.current_stream()`` and returns without synchronising.
    """
    L = _lib.load()
    if layout not in _lib.LAYOUTS:
        raise ValueError("layout must be 'czyx' or 'cxyz'")
    _dev_check("depth", depth, torch.float32)
    dev = depth.device
    _dev_check("offsets", offsets, torch.int64, dev)
    _dev_check("headers", headers, torch.int32, dev)
    if headers.dim() != 2 or headers.shape[1] != 6:
        raise ValueError("headers must have shape [n, 6]")
    n = headers.shape[0]
    if offsets.numel() != n + 1:
        raise ValueError("offsets must have n+1 entries")
    if not L.tsdf_resolution_supported(int(res)):
        raise ValueError(f"unsupported grid resolution {res} (multiple of 4 in 4..128)")
    R = int(res)
    if out is None:
        out = TsdfBatch(
            torch.empty((n, 3, R, R, R), dtype=torch.float32, device=dev),
            torch.empty((n,), dtype=torch.float32, device=dev),
            torch.empty((n, 3), dtype=torch.float32, device=dev),
            torch.empty((n,), dtype=torch.int32, device=dev),
        )
    else:
        _dev_check("out.tsdf", out.tsdf, torch.float32, dev)
        _dev_check("out.max_l", out.max_l, torch.float32, dev)
        _dev_check("out.mid_p", out.mid_p, torch.float32, dev)
        _dev_check("out.status", out.status, torch.int32, dev)
        if tuple(out.tsdf.shape) != (n, 3, R, R, R) or out.max_l.numel() != n or \
                tuple(out.mid_p.shape) != (n, 3) or out.status.numel() != n:
            raise ValueError("out tensors have the wrong shape")
    if n == 0:
        return out
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = L.tsdf_voxelize_hip(depth.data_ptr(), depth.numel(), offsets.data_ptr(), headers.data_ptr(), n, R,
                                 ctypes.byref(cam) if cam is not None else None,
                                 _lib.LAYOUTS[layout], stream, out.tsdf.data_ptr(),
                                 out.max_l.data_ptr(), out.mid_p.data_ptr(), out.status.data_ptr())
    _lib.check(rc, "tsdf_voxelize_hip")
    return out


class AabbBatch(NamedTuple):
    aabb: torch.Tensor    # float32[n,6] min xyz, max xyz
    grid: torch.Tensor    # float32[n,8] mid_p[3], max_l, voxel_len, trunc_dis, 0, 0
    ori: torch.Tensor     # float32[n,3] vox_ori
    status: torch.Tensor  # int32[n]


def aabb(depth: torch.Tensor, offsets: torch.Tensor, headers: torch.Tensor, res: int = 32,
         cam: Optional[_lib.TsdfCam] = None) -> AabbBatch:
    """Phase 1 + glue only: min_max_kernel + host numpy of pre/tsdf_numba.py:75-116,135-147."""
    L = _lib.load()
    _dev_check("depth", depth, torch.float32)
    dev = depth.device
    _dev_check("offsets", offsets, torch.int64, dev)
    _dev_check("headers", headers, torch.int32, dev)
    n = headers.shape[0]
    if offsets.numel() != n + 1:
        raise ValueError("offsets must have n+1 entries")
    ab = torch.empty((n, 6), dtype=torch.float32, device=dev)
    grid = torch.empty((n, 8), dtype=torch.float32, device=dev)
    ori = torch.empty((n, 3), dtype=torch.float32, device=dev)
    st = torch.empty((n,), dtype=torch.int32, device=dev)
    if n:
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            rc = L.tsdf_aabb_hip(depth.data_ptr(), depth.numel(), offsets.data_ptr(), headers.data_ptr(), n, int(res),
                                 ctypes.byref(cam) if cam is not None else None, stream,
                                 ab.data_ptr(), grid.data_ptr(), ori.data_ptr(), st.data_ptr())
        _lib.check(rc, "tsdf_aabb_hip")
    return AabbBatch(ab, grid, ori, st)


def voxelize_grid(depth: torch.Tensor, offsets: torch.Tensor, headers: torch.Tensor, grid: torch.Tensor,
                  res: int = 32, layout: str = "czyx", cam: Optional[_lib.TsdfCam] = None):
    """Phase 2 with caller-supplied grid placement: the batched form of the reference's
    ``tsdf_cal(data, vox_ori, voxel_len, truncation)`` (pre/tsdf_for.py:44-122).

    grid  float32[n,8] on the GPU: vox_ori[3], voxel_len, trunc_dis, 3 pad words per frame.
    Returns (tsdf float32[n,3,R,R,R], status int32[n]).
    """
    L = _lib.load()
    if layout not in _lib.LAYOUTS:
        raise ValueError("layout must be 'czyx' or 'cxyz'")
    _dev_check("depth", depth, torch.float32)
    dev = depth.device
    _dev_check("offsets", offsets, torch.int64, dev)
    _dev_check("headers", headers, torch.int32, dev)
    _dev_check("grid", grid, torch.float32, dev)
    n = headers.shape[0]
    if offsets.numel() != n + 1 or tuple(grid.shape) != (n, 8):
        raise ValueError("offsets must have n+1 entries and grid shape [n, 8]")
    if not L.tsdf_resolution_supported(int(res)):
        raise ValueError(f"unsupported grid resolution {res} (multiple of 4 in 4..128)")
    R = int(res)
    tsdf = torch.empty((n, 3, R, R, R), dtype=torch.float32, device=dev)
    st = torch.empty((n,), dtype=torch.int32, device=dev)
    if n:
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            rc = L.tsdf_voxelize_grid_hip(depth.data_ptr(), depth.numel(), offsets.data_ptr(), headers.data_ptr(), n, R,
                                          ctypes.byref(cam) if cam is not None else None,
                                          _lib.LAYOUTS[layout], stream, grid.data_ptr(), tsdf.data_ptr(),
                                          st.data_ptr())
        _lib.check(rc, "tsdf_voxelize_grid_hip")
    return tsdf, st


def voxelize_aug(depth: torch.Tensor, offsets: torch.Tensor, headers: torch.Tensor, xforms: torch.Tensor,
                 res: int = 64, layout: str = "czyx", cam: Optional[_lib.TsdfCam] = None,
                 out: Optional[TsdfBatch] = None) -> TsdfBatch:
    """Voxelization with a per-frame 3-D affine augmentation fused into the kernel
    (BASELINE.json configs[4]; see ``tsdf_voxelize_aug_hip`` in include/tsdf.h for the contract).

    xforms  float64[n,24] on the GPU: forward map rows {A_i0,A_i1,A_i2,b_i} then the inverse map
            (``augment.random_affines`` / ``augment.pack_affine`` build them).
    Returns the same TsdfBatch as :func:`voxelize`; ``max_l`` / ``mid_p`` are in the mapped frame.
    """
    L = _lib.load()
    if layout not in _lib.LAYOUTS:
        raise ValueError("layout must be 'czyx' or 'cxyz'")
    _dev_check("depth", depth, torch.float32)
    dev = depth.device
    _dev_check("offsets", offsets, torch.int64, dev)
    _dev_check("headers", headers, torch.int32, dev)
    _dev_check("xforms", xforms, torch.float64, dev)
    n = headers.shape[0]
    if offsets.numel() != n + 1 or tuple(xforms.shape) != (n, 24):
        raise ValueError("offsets must have n+1 entries and xforms shape [n, 24]")
    if not L.tsdf_resolution_supported(int(res)):
        raise ValueError(f"unsupported grid resolution {res} (multiple of 4 in 4..128)")
    R = int(res)
    if out is None:
        out = TsdfBatch(
            torch.empty((n, 3, R, R, R), dtype=torch.float32, device=dev),
            torch.empty((n,), dtype=torch.float32, device=dev),
            torch.empty((n, 3), dtype=torch.float32, device=dev),
            torch.empty((n,), dtype=torch.int32, device=dev),
        )
    elif tuple(out.tsdf.shape) != (n, 3, R, R, R):
        raise ValueError("out tensors have the wrong shape")
    if n:
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            rc = L.tsdf_voxelize_aug_hip(depth.data_ptr(), depth.numel(), offsets.data_ptr(), headers.data_ptr(), n, R,
                                         ctypes.byref(cam) if cam is not None else None,
                                         _lib.LAYOUTS[layout], stream, xforms.data_ptr(), out.tsdf.data_ptr(),
                                         out.max_l.data_ptr(), out.mid_p.data_ptr(), out.status.data_ptr())
        _lib.check(rc, "tsdf_voxelize_aug_hip")
    return out
