"""Vectorised numpy restatement of SURVEY.md Appendix A — TEST INFRASTRUCTURE ONLY.

An independent second statement of the numba path's arithmetic, used to cross-check
``tsdf_oracle.c`` at sizes where the reference's own Python triple loop
(/root/reference pre/tsdf_for.py:62-120, ~0.1 s per frame) is too slow, and by
``tools/make_goldens.py`` for the AABB expectation (the reference's ``min_max_kernel``,
pre/tsdf_numba.py:75-116, cannot be executed: no usable numba, no ``params`` module).

numpy evaluates each binary operation separately and rounds it to the array dtype, so
float64 arrays give exactly "float64 intermediates, unfused multiply-add".
"""
from __future__ import annotations

import numpy as np

FOCAL = 241.42  # pre/tsdf_numba.py:8 (Python float -> float64)
CENTER_X = 160  # :9
CENTER_Y = 120  # :10


def aabb(depth, header):
    """A.1: pre/tsdf_numba.py:84-96,140-141 -> (n_valid, min_p f32[3], max_p f32[3])."""
    l, t, r, b = (int(v) for v in header[2:6])
    bw, bh = r - l, b - t
    d = np.asarray(depth, dtype=np.float32).reshape(bh, bw)
    valid = np.abs(d) >= np.float32(1)  # :87 (NaN is invalid: include/tsdf.h)
    if not valid.any():
        return 0, None, None
    x = (np.arange(bw, dtype=np.int64) + l)[None, :]  # :84
    y = (np.arange(bh, dtype=np.int64) + t)[:, None]  # :85
    q = d.astype(np.float64) / FOCAL  # :91
    cam_x = (q * (x - CENTER_X)).astype(np.float32)  # :92, f32 smem store :95
    cam_y = ((-q) * (y - CENTER_Y)).astype(np.float32)  # :93
    cam_z = -d  # :94
    mn = np.array([cam_x[valid].min(), cam_y[valid].min(), cam_z[valid].min()], np.float32)
    mx = np.array([cam_x[valid].max(), cam_y[valid].max(), cam_z[valid].max()], np.float32)
    return int(valid.sum()), mn, mx


def glue(min_p, max_p, R=32):
    """A.2: pre/tsdf_numba.py:142-147, float32 throughout."""
    min_p = np.asarray(min_p, np.float32)
    max_p = np.asarray(max_p, np.float32)
    mid_p = (min_p + max_p) / 2
    len_e = max_p - min_p
    max_l = np.max(len_e)
    voxel_len = max_l / R
    trunc_dis = voxel_len * 3
    vox_ori = mid_p - max_l / 2 + voxel_len / 2
    assert mid_p.dtype == np.float32 and vox_ori.dtype == np.float32
    assert np.float32(voxel_len).dtype == np.float32
    return mid_p, np.float32(max_l), np.float32(voxel_len), np.float32(trunc_dis), vox_ori


def voxels(depth, header, ori, voxel_len, trunc_dis, R=32):
    """A.3: pre/tsdf_numba.py:15-72 -> (tsdf f32[3,R,R,R] in [c,z,y,x], pixmap int32[R,R,R])."""
    l, t, r, b = (int(v) for v in header[2:6])
    bw = r - l
    depth = np.asarray(depth, dtype=np.float32)
    ori = np.asarray(ori, np.float32).astype(np.float64)
    vl = np.float64(np.float32(voxel_len))
    tr = np.float64(np.float32(trunc_dis))
    idx = np.arange(R, dtype=np.float64)
    v_x = (ori[0] + idx * vl)[None, None, :]  # :26
    v_y = (ori[1] + idx * vl)[None, :, None]  # :27
    v_z = (ori[2] + idx * vl)[:, None, None]  # :28
    with np.errstate(all="ignore"):
        q = -FOCAL / v_z  # :30
        px_f = v_x * q + CENTER_X  # :31
        py_f = (-v_y) * q + CENTER_Y  # :32
        pix_x = np.trunc(np.nan_to_num(px_f, nan=0.0, posinf=2**31 - 1, neginf=-(2**31))).astype(np.int64)
        pix_y = np.trunc(np.nan_to_num(py_f, nan=0.0, posinf=2**31 - 1, neginf=-(2**31))).astype(np.int64)
        pix_x, pix_y = np.broadcast_arrays(pix_x, pix_y)
        inb = (pix_x >= l) & (pix_x < r) & (pix_y >= t) & (pix_y < b)  # :36
        gidx = np.where(inb, (pix_y - t) * bw + pix_x - l, 0)  # :38
        pd = depth[gidx]  # :39
        ok = inb & (np.abs(pd) >= np.float32(1))  # :40 (NaN is invalid)
        pd64 = pd.astype(np.float64)
        q2 = pd64 / FOCAL  # :43
        w_x = (pix_x - CENTER_X) * q2  # :44
        w_y = -(pix_y - CENTER_Y) * q2  # :45
        w_z = -pd64  # :46
        v_xb, v_yb, v_zb = np.broadcast_arrays(v_x, v_y, v_z)
        tx = np.abs(v_xb - w_x) / tr  # :47
        ty = np.abs(v_yb - w_y) / tr  # :48
        tz = np.abs(v_zb - w_z) / tr  # :49
        dist = np.sqrt(tx * tx + ty * ty + tz * tz)  # :51-52
        far = dist > 1  # :54
        out = np.stack([tx, ty, tz])
        out = np.where(far[None], 1.0, out)
        out = np.where(out > 1.0, 1.0, out)  # :58-60
        out = np.where((w_z > v_zb)[None], -out, out)  # :65-68
        out = np.where(ok[None], out, 0.0)  # :33-41
    pixmap = np.where(ok, gidx, np.where(inb, -2 - gidx, -1)).astype(np.int32)
    return out.astype(np.float32), pixmap


def frame(depth, header, R=32):
    """cal_tsdf_cuda (pre/tsdf_numba.py:119-161) for one frame -> (tsdf, max_l, mid_p)."""
    nv, mn, mx = aabb(depth, header)
    if nv == 0:
        return np.zeros((3, R, R, R), np.float32), np.float32(0), np.zeros(3, np.float32)
    mid_p, max_l, vl, tr, ori = glue(mn, mx, R)
    if not (max_l > 0 and np.isfinite(max_l)) or not np.isfinite(mid_p).all():
        return np.zeros((3, R, R, R), np.float32), np.float32(0), (mid_p if np.isfinite(mid_p).all() else np.zeros(3, np.float32))
    out, _ = voxels(depth, header, ori, vl, tr, R)
    return out, max_l, mid_p


def normalize_joints(gt, max_l, mid_p, clamp=True):
    """pre/joint_nor.py:8-18 + the clamp of 3D_CNN/train.py:241-242, float32; degenerate frames -> 0.5."""
    gt = np.asarray(gt, np.float32)
    n = gt.shape[0]
    j = gt.reshape(n, -1, 3)
    max_l = np.asarray(max_l, np.float32).reshape(n)
    mid_p = np.asarray(mid_p, np.float32).reshape(n, 3)
    with np.errstate(all="ignore"):
        out = (j - mid_p[:, None, :]) / max_l[:, None, None] + np.float32(0.5)
    if clamp:
        out = np.where(out < 0, np.float32(0), out)
        out = np.where(out > 1, np.float32(1), out)
    out = np.where((max_l > 0)[:, None, None], out, np.float32(0.5)).astype(np.float32)
    return out.reshape(gt.shape)
