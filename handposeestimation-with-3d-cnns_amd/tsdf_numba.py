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

import numpy as np
import torch

from . import _lib
from .voxelize import voxelize

FOCAL = 241.42     # pre/tsdf_numba.py:8
CENTER_X = 160     # :9
CENTER_Y = 120     # :10
VOXEL_RES = 32     # the missing ``params.VOXEL_RES`` (pre/tsdf_numba.py:3)


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("cal_tsdf_cuda needs a HIP device: this voxelizer has no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


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
    depth = torch.from_numpy(data).to(dev)
    offsets = torch.tensor([0, data.size], dtype=torch.int64, device=dev)
    headers = torch.from_numpy(header[None]).to(dev)
    out = voxelize(depth, offsets, headers, res=voxel_res, layout="czyx")
    status = int(out.status.item())  # synchronises, like the reference's copy_to_host (:158)
    if status != _lib.TSDF_FRAME_OK:
        what = "no valid pixel / zero extent" if status == _lib.TSDF_FRAME_DEGENERATE else "bad header"
        print("warning caught: ", what, "for bbox", header[2:6].tolist())
        return None
    tsdf = out.tsdf[0].cpu().numpy()
    max_l = np.float32(out.max_l[0].item())
    mid_p = out.mid_p[0].cpu().numpy()
    return tsdf, max_l, mid_p
