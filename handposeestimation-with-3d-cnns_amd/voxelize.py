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


_get_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _raw_stream(dev) -> int:
    """hipStream_t of torch's current stream on ``dev`` as an integer (without building a torch Stream object: at
    batch 16 the host side of a call is longer than the kernel, tools/exp_latency_parts.py)."""
    if _get_raw_stream is not None:
        return _get_raw_stream(dev.index)
    return torch.cuda.current_stream(dev).cuda_stream


class _Current:
    """``with _current(dev):`` — makes ``dev`` the current device, at no cost when it already is."""

    __slots__ = ("guard",)

    def __init__(self, dev):
        self.guard = None if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)

    def __enter__(self):
        if self.guard is not None:
            self.guard.__enter__()

    def __exit__(self, *exc):
        if self.guard is not None:
            self.guard.__exit__(*exc)
        return False


def _dev_check(name, t, dtype, device=None, host_ok=False):
    """``host_ok``: the per-frame metadata (offsets, headers, gt) may also be PAGE-LOCKED host memory, which the
    GPU reads over the link (include/tsdf.h) — a pinned CPU tensor then passes; pageable memory never does."""
    if t.__class__ is torch.Tensor and t.is_cuda and t.dtype is dtype and t.is_contiguous() and \
            (device is None or t.device == device):
        return   # the common case, in one expression: this runs seven times per call
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if host_ok and not t.is_cuda and t.is_pinned():
        device = None
    elif not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU" + (" or in page-locked host memory" if host_ok else "") +
                         f" (there is no CPU path); got device {t.device}")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if device is not None and t.device != device:
        raise ValueError(f"{name} is on {t.device}, expected {device}")


def _check_inputs(L, depth, offsets, headers, res, layout):
    """Shared validation of the packed-frame inputs; returns (device, n, R)."""
    if layout not in _lib.LAYOUTS:
        raise ValueError("layout must be 'czyx' or 'cxyz'")
    _dev_check("depth", depth, torch.float32)
    dev = depth.device
    _dev_check("offsets", offsets, torch.int64, dev, host_ok=True)
    _dev_check("headers", headers, torch.int32, dev, host_ok=True)
    if headers.dim() != 2 or headers.shape[1] != 6:
        raise ValueError("headers must have shape [n, 6]")
    n = headers.shape[0]
    if offsets.numel() != n + 1:
        raise ValueError("offsets must have n+1 entries")
    if not L.tsdf_resolution_supported(int(res)):
        raise ValueError(f"unsupported grid resolution {res} (multiple of 4 in 4..128)")
    return dev, n, int(res)


def _make_out(out, n, R, dev) -> "TsdfBatch":
    """Allocate the outputs, or validate caller-supplied ones (device, dtype, contiguity, shape): their
    data pointers go straight to the kernel."""
    if out is None:
        return TsdfBatch(
            torch.empty((n, 3, R, R, R), dtype=torch.float32, device=dev),
            torch.empty((n,), dtype=torch.float32, device=dev),
            torch.empty((n, 3), dtype=torch.float32, device=dev),
            torch.empty((n,), dtype=torch.int32, device=dev),
        )
    _dev_check("out.tsdf", out.tsdf, torch.float32, dev)
    _dev_check("out.max_l", out.max_l, torch.float32, dev)
    _dev_check("out.mid_p", out.mid_p, torch.float32, dev)
    _dev_check("out.status", out.status, torch.int32, dev)
    if tuple(out.tsdf.shape) != (n, 3, R, R, R) or out.max_l.numel() != n or \
            tuple(out.mid_p.shape) != (n, 3) or out.status.numel() != n:
        raise ValueError("out tensors have the wrong shape")
    return out


def _labels_struct(gt, n, dev, clamp, out_gt_nor=None, want_aug=False):
    """Validate the label tensors and build the ``tsdf_labels`` struct (kept alive by the caller)."""
    _dev_check("gt", gt, torch.float32, dev, host_ok=True)
    if gt.shape[0] != n or gt.numel() % (3 * max(n, 1)) != 0 and n > 0:
        raise ValueError("gt must have shape [n, 3*J] or [n, J, 3]")
    nc = gt.numel() // n if n else 63
    if nc % 3 or not 1 <= nc // 3 <= 170:
        raise ValueError("gt must hold 1..170 joints of 3 coordinates per frame")
    if out_gt_nor is None:
        out_gt_nor = torch.empty(gt.shape, dtype=torch.float32, device=dev)
    else:
        _dev_check("out_gt_nor", out_gt_nor, torch.float32, dev)
        if out_gt_nor.shape != gt.shape:
            raise ValueError("out_gt_nor must have gt's shape")
    gt_aug = torch.empty(gt.shape, dtype=torch.float32, device=dev) if want_aug else None
    lab = _lib.TsdfLabels(gt.data_ptr(), nc // 3, 1 if clamp else 0, out_gt_nor.data_ptr(),
                          gt_aug.data_ptr() if gt_aug is not None else None)
    return lab, out_gt_nor, gt_aug


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
    dev, n, R = _check_inputs(L, depth, offsets, headers, res, layout)
    out = _make_out(out, n, R, dev)
    if n == 0:
        return out
    with _Current(dev):
        stream = _raw_stream(dev)
        rc = L.tsdf_voxelize_hip(depth.data_ptr(), depth.numel(), offsets.data_ptr(), headers.data_ptr(), n, R,
                                 ctypes.byref(cam) if cam is not None else None,
                                 _lib.LAYOUTS[layout], stream, out.tsdf.data_ptr(),
                                 out.max_l.data_ptr(), out.mid_p.data_ptr(), out.status.data_ptr())
    _lib.check(rc, "tsdf_voxelize_hip")
    return out


def voxelize_labels(depth: torch.Tensor, offsets: torch.Tensor, headers: torch.Tensor, gt: torch.Tensor,
                    res: int = 32, layout: str = "czyx", cam: Optional[_lib.TsdfCam] = None, clamp: bool = True,
                    out: Optional[TsdfBatch] = None, out_gt_nor: Optional[torch.Tensor] = None,
                    gt_copy: bool = False):
    """:func:`voxelize` plus the label normalisation of the same launch: ``(gt - mid_p) / max_l + 0.5`` per
    joint coordinate (pre/joint_nor.py:8-18), clamped to [0,1] as 3D_CNN/train.py:241-242 does (``clamp``).
    gt float32[n,63] (or [n,J,3]) on the GPU.  Returns ``(TsdfBatch, gt_nor)``; frames whose status is not 0
    get 0.5 everywhere.

    ``offsets``, ``headers`` and ``gt`` may be pinned CPU tensors (read by the kernel over the link; they must not
    change before the launch has finished); ``gt_copy=True`` then also returns the joints as a device tensor,
    written by the same launch: ``(TsdfBatch, gt_nor, gt_on_device)``."""
    L = _lib.load()
    dev, n, R = _check_inputs(L, depth, offsets, headers, res, layout)
    out = _make_out(out, n, R, dev)
    lab, gt_nor, gt_dev = _labels_struct(gt, n, dev, clamp, out_gt_nor, want_aug=gt_copy)
    if n == 0:
        return (out, gt_nor, gt_dev) if gt_copy else (out, gt_nor)
    with _Current(dev):
        stream = _raw_stream(dev)
        rc = L.tsdf_voxelize_labels_hip(depth.data_ptr(), depth.numel(), offsets.data_ptr(), headers.data_ptr(), n, R,
                                        ctypes.byref(cam) if cam is not None else None, _lib.LAYOUTS[layout], stream,
                                        out.tsdf.data_ptr(), out.max_l.data_ptr(), out.mid_p.data_ptr(),
                                        out.status.data_ptr(), ctypes.byref(lab))
    _lib.check(rc, "tsdf_voxelize_labels_hip")
    return (out, gt_nor, gt_dev) if gt_copy else (out, gt_nor)


def voxelize_indexed(depth: torch.Tensor, offsets: torch.Tensor, headers: torch.Tensor, index: torch.Tensor,
                     gt: Optional[torch.Tensor] = None, res: int = 32, layout: str = "czyx",
                     cam: Optional[_lib.TsdfCam] = None, clamp: bool = True, out: Optional[TsdfBatch] = None,
                     gt_copy: bool = False, xforms: Optional[torch.Tensor] = None,
                     out_gt_nor: Optional[torch.Tensor] = None, out_gt: Optional[torch.Tensor] = None):
    """A batch drawn by index from a pack that lives on the GPU (``tsdf_voxelize_indexed_hip``): ``depth`` /
    ``offsets[N+1]`` / ``headers[N,6]`` (and ``gt[N,3J]``) describe the whole pack, uploaded once; frame i of the batch is
    pack frame ``index[i]`` (int64[n], any order — a shuffled minibatch; device or pinned host memory; or, for
    n <= 32 without ``xforms``, an ordinary CPU tensor, which is read during the call and travels to the GPU inside the
    kernel arguments: ``tsdf_voxelize_indexed_host_hip``).  Outputs are in batch order.  Bit-identical to :func:`voxelize_labels` on the gathered frames.  Returns ``TsdfBatch`` without ``gt``,
    else ``(TsdfBatch, gt_nor)`` or, with ``gt_copy=True``, ``(TsdfBatch, gt_nor, gt_of_the_batch)``.
    ``xforms`` float64[n,24] on the GPU (one map per batch position, as for :func:`voxelize_aug`) adds the fused 3-D
    augmentation: the labels are then mapped with it, ``gt_of_the_batch`` is ``T(joints)``.
    ``out`` / ``out_gt_nor`` / ``out_gt`` (the latter implies ``gt_copy``): preallocated outputs of exactly the batch's
    shapes — a loader's ring buffers; nothing is allocated then."""
    L = _lib.load()
    by_value = isinstance(index, torch.Tensor) and not index.is_cuda and not index.is_pinned() and \
        index.dtype is torch.int64 and index.dim() == 1 and index.is_contiguous() and \
        index.numel() <= _lib.INLINE_INDEX_MAX and xforms is None
    if not by_value:   # (a small index in ordinary host memory travels inside the kernel arguments instead)
        _dev_check("index", index, torch.int64, depth.device if isinstance(depth, torch.Tensor) else None, host_ok=True)
    if index.dim() != 1:
        raise ValueError("index must have shape [n]")
    dev, n_pack, R = _check_inputs(L, depth, offsets, headers, res, layout)
    n = index.numel()
    out = _make_out(out, n, R, dev)
    lab = gt_nor = gt_dev = None
    if xforms is not None:
        _dev_check("xforms", xforms, torch.float64, dev, host_ok=True)   # (page-locked host memory: read over the link)
        if xforms.numel() != 24 * n:
            raise ValueError("xforms must have shape [n, 24] (forward rows then inverse rows)")
    if gt is not None:
        _dev_check("gt", gt, torch.float32, dev, host_ok=True)
        if gt.dim() < 2 or gt.shape[0] != n_pack:
            raise ValueError("gt must hold the labels of every frame of the pack: [N, 3*J] or [N, J, 3]")
        nc = gt.numel() // n_pack if n_pack else 63
        if nc % 3 or not 1 <= nc // 3 <= 170:
            raise ValueError("gt must hold 1..170 joints of 3 coordinates per frame")
        shape = (n,) + tuple(gt.shape[1:])
        for name, t in (("out_gt_nor", out_gt_nor), ("out_gt", out_gt)):
            if t is not None:
                _dev_check(name, t, torch.float32, dev)
                if tuple(t.shape) != shape:
                    raise ValueError(f"{name} must have shape {shape}")
        gt_nor = out_gt_nor if out_gt_nor is not None else torch.empty(shape, dtype=torch.float32, device=dev)
        gt_copy = gt_copy or out_gt is not None
        gt_dev = out_gt if out_gt is not None else (torch.empty_like(gt_nor) if gt_copy else None)
        lab = _lib.TsdfLabels(gt.data_ptr(), nc // 3, 1 if clamp else 0, gt_nor.data_ptr(),
                              gt_dev.data_ptr() if gt_dev is not None else None)
    if n:
        with _Current(dev):
            head = (depth.data_ptr(), depth.numel(), offsets.data_ptr(), headers.data_ptr(), n_pack, index.data_ptr(), n, R,
                    ctypes.byref(cam) if cam is not None else None, _lib.LAYOUTS[layout], _raw_stream(dev))
            tail = (out.tsdf.data_ptr(), out.max_l.data_ptr(), out.mid_p.data_ptr(), out.status.data_ptr(),
                    ctypes.byref(lab) if lab is not None else None)
            if by_value:
                rc = L.tsdf_voxelize_indexed_host_hip(*head, *tail)
            elif xforms is None:
                rc = L.tsdf_voxelize_indexed_hip(*head, *tail)
            else:
                rc = L.tsdf_voxelize_indexed_aug_hip(*head, xforms.data_ptr(), *tail)
        _lib.check(rc, "tsdf_voxelize_indexed_hip")
    if gt is None:
        return out
    return (out, gt_nor, gt_dev) if gt_copy else (out, gt_nor)


def normalize_joints(gt: torch.Tensor, max_l: torch.Tensor, mid_p: torch.Tensor, clamp: bool = True,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Labels into the voxel cube's [0,1] frame on their own (``tsdf_normalize_joints_hip``):
    ``(gt - mid_p) / max_l + 0.5`` (pre/joint_nor.py:8-18), clamped as 3D_CNN/train.py:241-242 unless
    ``clamp=False``; frames with ``max_l == 0`` give 0.5.  gt [n,63] or [n,21,3]; result has gt's shape."""
    return _norm_call(gt, max_l, mid_p, clamp, False, out)


def denormalize_joints(pred: torch.Tensor, max_l: torch.Tensor, mid_p: torch.Tensor,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Inverse of :func:`normalize_joints` for network outputs: ``(pred - 0.5) * max_l + mid_p``
    (3D_CNN/train.py:263-266)."""
    return _norm_call(pred, max_l, mid_p, False, True, out)


def _norm_call(x, max_l, mid_p, clamp, inverse, out):
    L = _lib.load()
    _dev_check("joints", x, torch.float32)
    dev = x.device
    n = x.shape[0]
    _dev_check("max_l", max_l, torch.float32, dev)
    _dev_check("mid_p", mid_p, torch.float32, dev)
    if max_l.numel() != n or tuple(mid_p.shape) != (n, 3):
        raise ValueError("max_l must be [n] and mid_p [n,3]")
    nc = x.numel() // n if n else 63
    if n and (x.numel() != n * nc or nc % 3 or not 1 <= nc // 3 <= 170):
        raise ValueError("joints must have shape [n, 3*J] or [n, J, 3]")
    if out is None:
        out = torch.empty_like(x)
    else:
        _dev_check("out", out, torch.float32, dev)
        if out.shape != x.shape:
            raise ValueError("out must have the input's shape")
    if n:
        with _Current(dev):
            stream = _raw_stream(dev)
            if inverse:
                rc = L.tsdf_denormalize_joints_hip(x.data_ptr(), max_l.data_ptr(), mid_p.data_ptr(), n, nc // 3,
                                                   stream, out.data_ptr())
            else:
                rc = L.tsdf_normalize_joints_hip(x.data_ptr(), max_l.data_ptr(), mid_p.data_ptr(), n, nc // 3,
                                                 1 if clamp else 0, stream, out.data_ptr())
        _lib.check(rc, "tsdf_(de)normalize_joints_hip")
    return out


def release_stream(stream=None) -> None:
    """Tell the library that ``stream`` (default: the current one) is going away (``tsdf_stream_release``).  The
    stream is synchronised first: its work-queue word and mailboxes must not be handed to another stream while one of
    its launches is still running (include/tsdf.h, "stream ownership")."""
    L = _lib.load()
    s = stream if stream is not None else torch.cuda.current_stream()
    s.synchronize()
    L.tsdf_stream_release(s.cuda_stream)


def voxel_pixels(depth: torch.Tensor, offsets: torch.Tensor, headers: torch.Tensor, res: int = 32,
                 layout: str = "czyx", grid: Optional[torch.Tensor] = None, cam: Optional[_lib.TsdfCam] = None):
    """Diagnostic (``tsdf_debug_pixmap_hip``, include/tsdf_debug.h): the voxelizer together with the pixel every voxel
    gathers.  Runs in the DEBUG build of the library (``_lib.load_debug()``: the product does not carry the hook).
    Returns ``(tsdf, pixmap int32[n,R,R,R] indexed [z,y,x], status)``; pixmap values as in include/tsdf_debug.h."""
    L = _lib.load_debug()
    dev, n, R = _check_inputs(L, depth, offsets, headers, res, layout)
    if grid is not None:
        _dev_check("grid", grid, torch.float32, dev)
        if tuple(grid.shape) != (n, 8):
            raise ValueError("grid must have shape [n, 8]")
    tsdf = torch.empty((n, 3, R, R, R), dtype=torch.float32, device=dev)
    pm = torch.empty((n, R, R, R), dtype=torch.int32, device=dev)
    st = torch.empty((n,), dtype=torch.int32, device=dev)
    if n:
        with _Current(dev):
            stream = _raw_stream(dev)
            rc = L.tsdf_debug_pixmap_hip(depth.data_ptr(), depth.numel(), offsets.data_ptr(), headers.data_ptr(), n, R,
                                         ctypes.byref(cam) if cam is not None else None, _lib.LAYOUTS[layout], stream,
                                         grid.data_ptr() if grid is not None else None, tsdf.data_ptr(),
                                         pm.data_ptr(), st.data_ptr())
        _lib.check(rc, "tsdf_debug_pixmap_hip")
    return tsdf, pm, st


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
    _dev_check("offsets", offsets, torch.int64, dev, host_ok=True)
    _dev_check("headers", headers, torch.int32, dev, host_ok=True)
    if headers.dim() != 2 or headers.shape[1] != 6:
        raise ValueError("headers must have shape [n, 6]")
    n = headers.shape[0]
    if offsets.numel() != n + 1:
        raise ValueError("offsets must have n+1 entries")
    if not L.tsdf_resolution_supported(int(res)):
        raise ValueError(f"unsupported grid resolution {res} (multiple of 4 in 4..128)")
    ab = torch.empty((n, 6), dtype=torch.float32, device=dev)
    grid = torch.empty((n, 8), dtype=torch.float32, device=dev)
    ori = torch.empty((n, 3), dtype=torch.float32, device=dev)
    st = torch.empty((n,), dtype=torch.int32, device=dev)
    if n:
        with _Current(dev):
            stream = _raw_stream(dev)
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
    dev, n, R = _check_inputs(L, depth, offsets, headers, res, layout)
    _dev_check("grid", grid, torch.float32, dev)
    if tuple(grid.shape) != (n, 8):
        raise ValueError("grid must have shape [n, 8]")
    tsdf = torch.empty((n, 3, R, R, R), dtype=torch.float32, device=dev)
    st = torch.empty((n,), dtype=torch.int32, device=dev)
    if n:
        with _Current(dev):
            stream = _raw_stream(dev)
            rc = L.tsdf_voxelize_grid_hip(depth.data_ptr(), depth.numel(), offsets.data_ptr(), headers.data_ptr(), n, R,
                                          ctypes.byref(cam) if cam is not None else None,
                                          _lib.LAYOUTS[layout], stream, grid.data_ptr(), tsdf.data_ptr(),
                                          st.data_ptr())
        _lib.check(rc, "tsdf_voxelize_grid_hip")
    return tsdf, st


def voxelize_aug(depth: torch.Tensor, offsets: torch.Tensor, headers: torch.Tensor, xforms: torch.Tensor,
                 res: int = 64, layout: str = "czyx", cam: Optional[_lib.TsdfCam] = None,
                 out: Optional[TsdfBatch] = None, gt: Optional[torch.Tensor] = None, clamp: bool = True):
    """Voxelization with a per-frame 3-D affine augmentation fused into the kernel
    (BASELINE.json configs[4]; see ``tsdf_voxelize_aug_hip`` in include/tsdf.h for the contract).

    xforms  float64[n,24] on the GPU: forward map rows {A_i0,A_i1,A_i2,b_i} then the inverse map
            (``augment.random_affines`` / ``augment.pack_affine`` build them).
    gt      optional float32[n,63] labels: they are mapped with the frame's forward map and normalised in the
            augmented grid by the same launch (``tsdf_voxelize_aug_labels_hip``).
    Returns the same TsdfBatch as :func:`voxelize` (``max_l`` / ``mid_p`` in the mapped frame), or
    ``(TsdfBatch, gt_nor, gt_aug)`` when ``gt`` is given.
    """
    L = _lib.load()
    dev, n, R = _check_inputs(L, depth, offsets, headers, res, layout)
    _dev_check("xforms", xforms, torch.float64, dev)
    if tuple(xforms.shape) != (n, 24):
        raise ValueError("xforms must have shape [n, 24]")
    out = _make_out(out, n, R, dev)
    lab = gt_nor = gt_aug = None
    if gt is not None:
        lab, gt_nor, gt_aug = _labels_struct(gt, n, dev, clamp, want_aug=True)
    if n:
        with _Current(dev):
            stream = _raw_stream(dev)
            args = (depth.data_ptr(), depth.numel(), offsets.data_ptr(), headers.data_ptr(), n, R,
                    ctypes.byref(cam) if cam is not None else None, _lib.LAYOUTS[layout], stream, xforms.data_ptr(),
                    out.tsdf.data_ptr(), out.max_l.data_ptr(), out.mid_p.data_ptr(), out.status.data_ptr())
            if lab is None:
                rc = L.tsdf_voxelize_aug_hip(*args)
            else:
                rc = L.tsdf_voxelize_aug_labels_hip(*args, ctypes.byref(lab))
        _lib.check(rc, "tsdf_voxelize_aug_hip")
    return out if gt is None else (out, gt_nor, gt_aug)
