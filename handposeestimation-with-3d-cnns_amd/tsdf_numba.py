"""Drop-in for the reference module ``pre/tsdf_numba.py`` (same entry point, same result tuple).

    from tsdf_numba import cal_tsdf_cuda          # reference
    tsdf, max_l, mid_p = cal_tsdf_cuda(s)

``s`` is the dict the reference builds from one MSRA ``.bin`` file
(``{'header': int32[6], 'data': float32[N]}``, pre/tsdf_numba.py:122-133; the key ``'depth'`` that
pre/time_test.py:18-21 uses is accepted as well).  The result is what pre/tsdf_numba.py:161
returns: ``(tsdf float32[3,R,R,R] indexed [c,z,y,x], max_l numpy.float32, mid_p float32[3])``.

Where the reference launches two numba kernels and crosses PCIe four times per frame
(:133,:137,:151,:158), this calls the fused HIP kernel once.  For throughput use the batched
``voxelize`` (one launch for the whole batch, results left on the GPU); this single-frame form
exists so that existing call sites keep working.  There is no CPU fallback.
"""
from __future__ import annotations

import threading

import numpy as np
import torch

from . import _lib

FOCAL = 241.42     # pre/tsdf_numba.py:8
CENTER_X = 160     # :9
CENTER_Y = 120     # :10
VOXEL_RES = 32     # the missing ``params.VOXEL_RES`` (pre/tsdf_numba.py:3)


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("cal_tsdf_cuda needs a HIP device: this voxelizer has no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


_tls = threading.local()   # per thread and device: one page-locked + one device buffer each way, reused from call to call


class _Bufs:
    """The single frame's round trip in TWO copies: offsets, header and depth go up as one page-locked block, volume,
    max_l, mid_p and status come back as one.  (The reference crosses the link four times per frame with pageable
    memory, pre/tsdf_numba.py:133,137,151,158; the first version of this shim did six small pageable copies and three
    synchronisations: 0.16 ms per frame, most of it waiting.)"""

    IN_HEAD = 128  # bytes: offsets int64[2] at 0, header int32[6] at 16, grid float32[8] at 64 (tsdf_for.tsdf_cal's
                   # placement), depth float32[] from 128 (16-byte aligned)

    def __init__(self, dev, npx, R):
        self.dev, self.cap_px, self.R = dev, max(npx, 160 * 160), R
        nin = self.IN_HEAD + 4 * self.cap_px
        self.h_in = torch.empty(nin, dtype=torch.uint8).pin_memory()
        self.d_in = torch.empty(nin, dtype=torch.uint8, device=dev)
        self.h_in_np = self.h_in.numpy()
        self.h_off = self.h_in_np[0:16].view(np.int64)
        self.h_hdr = self.h_in_np[16:40].view(np.int32)
        self.h_grid = self.h_in_np[64:96].view(np.float32)
        self.h_depth = self.h_in_np[self.IN_HEAD:].view(np.float32)
        self.nvol = 3 * R ** 3
        self.d_out = torch.empty(self.nvol + 8, dtype=torch.float32, device=dev)   # volume, max_l, mid_p[3], status, pad
        self.h_out = torch.empty(self.nvol + 8, dtype=torch.float32).pin_memory()
        self.h_out_np = self.h_out.numpy()
        base_in, base_out = self.d_in.data_ptr(), self.d_out.data_ptr()
        self.p_off, self.p_hdr, self.p_grid, self.p_depth = base_in, base_in + 16, base_in + 64, base_in + self.IN_HEAD
        self.p_tsdf, self.p_max_l = base_out, base_out + 4 * self.nvol
        self.p_mid, self.p_status = self.p_max_l + 4, self.p_max_l + 16


def _bufs(dev, npx: int, R: int) -> _Bufs:
    key = (dev.index, R)
    if not hasattr(_tls, "bufs"):
        _tls.bufs = {}
    b = _tls.bufs.get(key)
    if b is None or b.cap_px < npx:
        b = _tls.bufs[key] = _Bufs(dev, npx, R)
    return b


def cal_tsdf_cuda(s, voxel_res: int = VOXEL_RES):
    """One frame -> (tsdf, max_l, mid_p), or ``None`` for a frame the reference gives up on.

    The reference catches the numeric ``RuntimeWarning`` a degenerate frame raises, prints its
    intermediate values and returns ``None`` (pre/tsdf_numba.py:162-171); a frame without any valid
    pixel, or whose valid pixels span no volume, takes that path here too (one printed line).
    """
    header = np.ascontiguousarray(s["header"], dtype=np.int32).reshape(6)
    data = s["data"] if "data" in s else s["depth"]
    data = np.ascontiguousarray(data, dtype=np.float32).reshape(-1)
    dev = _device()
    L = _lib.load()
    if not L.tsdf_resolution_supported(int(voxel_res)):
        raise ValueError(f"unsupported grid resolution {voxel_res} (multiple of 4 in 4..128)")
    b = _bufs(dev, data.size, int(voxel_res))
    npx = data.size
    b.h_off[0], b.h_off[1] = 0, npx
    b.h_hdr[:] = header
    b.h_depth[:npx] = data
    nin = b.IN_HEAD + 4 * npx
    stream = torch.cuda.current_stream(dev)
    b.d_in[:nin].copy_(b.h_in[:nin], non_blocking=True)                        # H2D, one copy     (:133,:151)
    rc = L.tsdf_voxelize_hip(b.p_depth, npx, b.p_off, b.p_hdr, 1, int(voxel_res), None, _lib.TSDF_LAYOUT_CZYX,
                             stream.cuda_stream, b.p_tsdf, b.p_max_l, b.p_mid, b.p_status)
    _lib.check(rc, "tsdf_voxelize_hip")
    b.h_out.copy_(b.d_out, non_blocking=True)                                   # D2H, one copy     (:137,:158)
    stream.synchronize()
    status = int(b.h_out_np[b.nvol + 4:b.nvol + 5].view(np.int32)[0])
    if status != _lib.TSDF_FRAME_OK:
        what = "no valid pixel / zero extent" if status == _lib.TSDF_FRAME_DEGENERATE else "bad header"
        print("warning caught: ", what, "for bbox", header[2:6].tolist())
        return None
    R = int(voxel_res)
    tsdf = b.h_out_np[:b.nvol].reshape(3, R, R, R).copy()
    max_l = np.float32(b.h_out_np[b.nvol])
    mid_p = b.h_out_np[b.nvol + 1:b.nvol + 4].copy()
    return tsdf, max_l, mid_p
