"""Parity oracle for the TSDF hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package, and only as the checker / the reported CPU
baseline.  The product package never imports it.

``oracle/tsdf_oracle.c`` is the C restatement (follows /root/reference
pre/tsdf_numba.py:15-72,84-96,140-147 with numba's inferred types, SURVEY.md
Appendix A); this module is its ctypes loader.  ``oracle/tsdf_oracle_np.py`` is an
independent numpy restatement of the same contract used to cross-check the C file.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtsdf_oracle.so")
# TSDF_ORACLE_SO: another build of the same file (tests/test_tiers_cpu.py runs the golden tests against the
# AddressSanitizer/UBSan build, `make -C oracle asan`, in a child process)
_SO_OVERRIDE = os.environ.get("TSDF_ORACLE_SO")


class TsdfCam(ctypes.Structure):
    """Mirror of ``tsdf_cam`` (include/tsdf.h)."""

    _fields_ = [
        ("focal", ctypes.c_double),
        ("cx", ctypes.c_double),
        ("cy", ctypes.c_double),
        ("invalid_eps", ctypes.c_float),
        ("trunc_voxels", ctypes.c_float),
    ]


def build(force: bool = False) -> str:
    """Compile libtsdf_oracle.so with gcc (oracle/Makefile) if it is missing or stale."""
    src = os.path.join(_HERE, "tsdf_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "tsdf.h")
    stale = (
        force
        or not os.path.exists(_SO)
        or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    )
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libtsdf_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if _SO_OVERRIDE:
            L = ctypes.CDLL(_SO_OVERRIDE)
        else:
            build()
            L = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int32)
        lp = ctypes.POINTER(ctypes.c_int64)
        cp = ctypes.POINTER(TsdfCam)
        L.tsdf_oracle_aabb.restype = ctypes.c_long
        L.tsdf_oracle_aabb.argtypes = [fp, ip, cp, fp, fp]
        L.tsdf_oracle_glue.restype = None
        L.tsdf_oracle_glue.argtypes = [fp, fp, ctypes.c_int, cp, fp, fp]
        L.tsdf_oracle_voxels.restype = None
        L.tsdf_oracle_voxels.argtypes = [fp, ip, fp, ctypes.c_float, ctypes.c_float, ctypes.c_int, cp,
                                         ctypes.c_int, fp, ip]
        L.tsdf_oracle_voxelize.restype = ctypes.c_int
        L.tsdf_oracle_voxelize.argtypes = [fp, lp, ip, ctypes.c_int, ctypes.c_int, cp, ctypes.c_int,
                                           ctypes.c_int, fp, fp, fp, ip, fp, fp, fp]
        dp = ctypes.POINTER(ctypes.c_double)
        L.tsdf_oracle_voxelize_aug.restype = ctypes.c_int
        L.tsdf_oracle_voxelize_aug.argtypes = [fp, lp, ip, ctypes.c_int, ctypes.c_int, cp, ctypes.c_int,
                                               ctypes.c_int, dp, fp, fp, fp, ip]
        L.tsdf_oracle_normalize_joints.restype = None
        L.tsdf_oracle_normalize_joints.argtypes = [fp, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp]
        L.tsdf_oracle_transform_joints.restype = None
        L.tsdf_oracle_transform_joints.argtypes = [fp, dp, ctypes.c_int, ctypes.c_int, fp]
        _lib = L
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct)) if a is not None else None


def aabb(depth, header):
    """A.1 -> (n_valid, min_p f32[3], max_p f32[3])."""
    depth = np.ascontiguousarray(depth, dtype=np.float32)
    header = np.ascontiguousarray(header, dtype=np.int32)
    mn = np.zeros(3, np.float32)
    mx = np.zeros(3, np.float32)
    nv = lib().tsdf_oracle_aabb(_p(depth, ctypes.c_float), _p(header, ctypes.c_int32), None,
                                _p(mn, ctypes.c_float), _p(mx, ctypes.c_float))
    return int(nv), mn, mx


def glue(min_p, max_p, R=32):
    """A.2 -> (grid f32[8] = mid_p[3],max_l,voxel_len,trunc,0,0 ; ori f32[3])."""
    mn = np.ascontiguousarray(min_p, dtype=np.float32)
    mx = np.ascontiguousarray(max_p, dtype=np.float32)
    grid = np.zeros(8, np.float32)
    ori = np.zeros(3, np.float32)
    lib().tsdf_oracle_glue(_p(mn, ctypes.c_float), _p(mx, ctypes.c_float), R, None,
                           _p(grid, ctypes.c_float), _p(ori, ctypes.c_float))
    return grid, ori


def voxels(depth, header, ori, voxel_len, trunc_dis, R=32, layout=0, want_pixmap=False, cam=None):
    """A.3 with explicit grid parameters -> tsdf f32[3,R,R,R] (and pixmap int32[R,R,R]); ``cam`` as in :func:`voxelize`."""
    depth = np.ascontiguousarray(depth, dtype=np.float32)
    header = np.ascontiguousarray(header, dtype=np.int32)
    ori = np.ascontiguousarray(ori, dtype=np.float32)
    out = np.empty((3, R, R, R), np.float32)
    pm = np.empty((R, R, R), np.int32) if want_pixmap else None
    lib().tsdf_oracle_voxels(_p(depth, ctypes.c_float), _p(header, ctypes.c_int32),
                             _p(ori, ctypes.c_float), float(np.float32(voxel_len)),
                             float(np.float32(trunc_dis)), R, _cam(cam), layout,
                             _p(out, ctypes.c_float), _p(pm, ctypes.c_int32))
    return (out, pm) if want_pixmap else out


def _cam(cam):
    """None, a TsdfCam, or (focal, cx, cy, invalid_eps, trunc_voxels) -> argument for the C entry points."""
    if cam is None or isinstance(cam, TsdfCam):
        return ctypes.byref(cam) if cam is not None else None
    return ctypes.byref(TsdfCam(*[float(v) for v in cam]))


def voxelize(depth, offsets, headers, R=32, layout=0, n_threads=1, want_tsdf=True, extras=False, cam=None):
    """Batch form, same argument meaning as ``tsdf_voxelize_hip`` but on host arrays.

    Returns dict(tsdf, max_l, mid_p, status[, aabb, grid, ori], threads).
    """
    depth = np.ascontiguousarray(depth, dtype=np.float32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    headers = np.ascontiguousarray(headers, dtype=np.int32).reshape(-1, 6)
    n = headers.shape[0]
    out = np.empty((n, 3, R, R, R), np.float32) if want_tsdf else None
    max_l = np.empty(n, np.float32)
    mid_p = np.empty((n, 3), np.float32)
    status = np.empty(n, np.int32)
    ab = np.empty((n, 6), np.float32) if extras else None
    grid = np.empty((n, 8), np.float32) if extras else None
    ori = np.empty((n, 3), np.float32) if extras else None
    used = lib().tsdf_oracle_voxelize(
        _p(depth, ctypes.c_float), _p(offsets, ctypes.c_int64), _p(headers, ctypes.c_int32), n, R,
        _cam(cam), layout, n_threads, _p(out, ctypes.c_float), _p(max_l, ctypes.c_float),
        _p(mid_p, ctypes.c_float), _p(status, ctypes.c_int32), _p(ab, ctypes.c_float),
        _p(grid, ctypes.c_float), _p(ori, ctypes.c_float))
    res = dict(tsdf=out, max_l=max_l, mid_p=mid_p, status=status, threads=int(used))
    if extras:
        res.update(aabb=ab, grid=grid, ori=ori)
    return res


def voxelize_aug(depth, offsets, headers, xforms, R=32, layout=0, n_threads=1, cam=None):
    """Augmented form (re-specified, parity unpinned; see tsdf_oracle.c): xforms float64[n,24] =
    forward affine rows {A_i0,A_i1,A_i2,b_i} then the inverse.  Returns dict(tsdf,max_l,mid_p,status)."""
    depth = np.ascontiguousarray(depth, dtype=np.float32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    headers = np.ascontiguousarray(headers, dtype=np.int32).reshape(-1, 6)
    xforms = np.ascontiguousarray(xforms, dtype=np.float64).reshape(-1, 24)
    n = headers.shape[0]
    assert xforms.shape[0] == n
    out = np.empty((n, 3, R, R, R), np.float32)
    max_l = np.empty(n, np.float32)
    mid_p = np.empty((n, 3), np.float32)
    status = np.empty(n, np.int32)
    lib().tsdf_oracle_voxelize_aug(
        _p(depth, ctypes.c_float), _p(offsets, ctypes.c_int64), _p(headers, ctypes.c_int32), n, R, _cam(cam),
        layout, n_threads, _p(xforms, ctypes.c_double), _p(out, ctypes.c_float), _p(max_l, ctypes.c_float),
        _p(mid_p, ctypes.c_float), _p(status, ctypes.c_int32))
    return dict(tsdf=out, max_l=max_l, mid_p=mid_p, status=status)


def normalize_joints(gt, max_l, mid_p, clamp=True):
    """pre/joint_nor.py:8-18 + clamp (3D_CNN/train.py:241-242); gt float32[n,3J] -> same shape."""
    gt = np.ascontiguousarray(gt, dtype=np.float32)
    n = gt.shape[0]
    J = gt.reshape(n, -1).shape[1] // 3
    max_l = np.ascontiguousarray(max_l, dtype=np.float32).reshape(n)
    mid_p = np.ascontiguousarray(mid_p, dtype=np.float32).reshape(n, 3)
    out = np.empty_like(gt)
    lib().tsdf_oracle_normalize_joints(_p(gt, ctypes.c_float), _p(max_l, ctypes.c_float), _p(mid_p, ctypes.c_float),
                                       n, J, 1 if clamp else 0, _p(out, ctypes.c_float))
    return out


def transform_joints(gt, xforms):
    """T(joint) with each frame's forward map (augmented labels), float32 result of gt's shape."""
    gt = np.ascontiguousarray(gt, dtype=np.float32)
    n = gt.shape[0]
    J = gt.reshape(n, -1).shape[1] // 3
    xforms = np.ascontiguousarray(xforms, dtype=np.float64).reshape(n, 24)
    out = np.empty_like(gt)
    lib().tsdf_oracle_transform_joints(_p(gt, ctypes.c_float), _p(xforms, ctypes.c_double), n, J, _p(out, ctypes.c_float))
    return out
