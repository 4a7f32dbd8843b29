"""Offline preprocessing in the reference's on-disk schema — one GPU launch per gesture.

The reference driver (pre/read_MSRA.py:37-140) walks ``<DB>/<subject>/<gesture>/``, voxelizes every
frame on the CPU (~0.2 s each) and writes, per subject directory under ``./result``:

    TSDF/<gesture>.npz          tsdf[n,3,32,32,32], max_l[n], mid_p[n,3]     (:119-120)
    ground_truth/<gesture>.npy  gt[n,63]                                      (:121-122)
    num/<gesture>.npy           n (scalar)                                    (:123-124)
    Point_Cloud/<gesture>.npy   [n,6000,3] random resample of the point cloud (:117-118)
    ../data_num-<subject>.npy   frames of the subject (scalar)                (:139)

which is what ``3D_CNN/dataset.py:93-131`` reads back.  ``preprocess_tree`` writes the same files with the
voxelization done by the HIP kernel: a gesture (~500 frames) is packed (``packing.pack_bin_files``), uploaded
once and voxelized in one launch.

Differences from the reference writer, all deliberate and switchable:
  * grid placement uses ALL valid pixels (the numba path), not the random 6000-point
    resample ``DataProcess.process()`` uses — the files are reproducible;
  * arrays are float32 (the reader casts to float32 anyway, 3D_CNN/dataset.py:101-103); ``dtype=np.float64``
    gives the reference's ``np.empty`` default back;
  * layout is ``[c,x,y,z]`` — what the reference writer produced, since it went through the CPU loop
    (pre/tsdf_for.py:118-120); pass ``layout="czyx"`` for the numba layout;
  * a ``status`` array is added to the npz (0 ok / 1 degenerate / 2 bad header): the reference crashes or
    writes garbage for such frames;
  * ``gt_3d=True`` stores the labels as ``[n,21,3]`` WITH z NEGATED — the only form the reference reader handles
    (3D_CNN/dataset.py:107-109 negates z of 3-D label arrays, pairing with the commented-out writer lines
    pre/read_MSRA.py:81-82 that pre-negate it; for the ``[n,63]`` array its own writer saves, ``g_t`` is undefined).
    After the reader's flip the labels are back in the camera frame (z = -depth) that ``mid_p`` lives in.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Optional, Sequence

import numpy as np

from . import packing

_SUBDIRS = ("Point_Cloud", "TSDF", "ground_truth", "num")


def _default_voxelize(pk: packing.PackedFrames, res: int, layout: str, device):
    """Upload + one launch; returns host arrays (tsdf, max_l, mid_p, status)."""
    import torch

    from .voxelize import voxelize

    depth, offsets, headers = pk.to_torch(device, pin=True, non_blocking=True)
    out = voxelize(depth, offsets, headers, res=res, layout=layout)
    torch.cuda.synchronize(depth.device)
    return (out.tsdf.cpu().numpy(), out.max_l.cpu().numpy(), out.mid_p.cpu().numpy(),
            out.status.cpu().numpy())


def resample_point_clouds(pk: packing.PackedFrames, points_num: int = 6000,
                          rng: Optional[np.random.Generator] = None) -> np.ndarray:
    """``DataProcess.point_cloud`` + ``set_length`` (pre/process.py:30-84) for every frame of a pack:
    ``float64[n, points_num, 3]``.  Frames without any non-zero point give zeros."""
    from .process import DataProcess

    rng = rng if rng is not None else np.random.default_rng()
    n = len(pk)
    out = np.zeros((n, points_num, 3), np.float64)
    for i in range(n):
        header, depth = pk.frame(i)
        pts = DataProcess({"header": header, "depth": depth}, None, points_num).point_cloud()
        m = pts.shape[0]
        if m == 0:
            continue
        if m < points_num:  # keep every point once, fill up with replacement (pre/process.py:74-79)
            idx = np.arange(points_num)
            idx[m:] = rng.integers(0, m, size=points_num - m)
        else:
            idx = rng.integers(0, m, size=points_num)
        out[i] = pts[idx]
    return out


def write_gesture(sub_dir: str, gesture: str, tsdf: np.ndarray, max_l: np.ndarray, mid_p: np.ndarray,
                  ground_truth: np.ndarray, status: Optional[np.ndarray] = None,
                  point_cloud: Optional[np.ndarray] = None, gt_3d: bool = False) -> None:
    """The four per-gesture files of pre/read_MSRA.py:117-124 (directories are created as needed)."""
    for d in _SUBDIRS:
        os.makedirs(os.path.join(sub_dir, d), exist_ok=True)
    n = int(tsdf.shape[0])
    extra = {} if status is None else {"status": np.asarray(status, np.int32)}
    np.savez(os.path.join(sub_dir, "TSDF", "%s.npz" % gesture), tsdf=tsdf, max_l=max_l, mid_p=mid_p, **extra)
    gt = np.asarray(ground_truth, np.float32).reshape(n, -1)
    if gt_3d:
        gt = gt.reshape(n, 21, 3).copy()
        gt[:, :, 2] = -gt[:, :, 2]  # the reference reader flips it back (3D_CNN/dataset.py:107-109)
    np.save(os.path.join(sub_dir, "ground_truth", "%s.npy" % gesture), gt)
    np.save(os.path.join(sub_dir, "num", "%s.npy" % gesture), n)
    if point_cloud is not None:
        np.save(os.path.join(sub_dir, "Point_Cloud", "%s.npy" % gesture), point_cloud)


def preprocess_tree(db_dir: str, save_dir: str, *, res: int = 32, layout: str = "cxyz", dtype=np.float32,
                    points_num: int = 6000, point_clouds: bool = True, gt_3d: bool = False,
                    subjects: Optional[Sequence[str]] = None, gestures: Optional[Sequence[str]] = None,
                    device="cuda", rng: Optional[np.random.Generator] = None,
                    voxelize_fn: Optional[Callable] = None, verbose: bool = False) -> Dict[str, int]:
    """Replacement for ``read_MSRA.main()`` (pre/read_MSRA.py:37-140, AUG=False): voxelize a whole MSRA tree
    into ``save_dir`` in the reference's schema.  Returns ``{subject: frames}``.

    ``voxelize_fn(pack, res, layout, device) -> (tsdf, max_l, mid_p, status)`` may replace the HIP call
    (tests use it to check the file handling without a GPU); by default the HIP voxelizer runs and a
    missing library or device is an error."""
    vox = voxelize_fn if voxelize_fn is not None else _default_voxelize
    os.makedirs(save_dir, exist_ok=True)
    subs = list(subjects) if subjects is not None else sorted(
        d for d in os.listdir(db_dir) if os.path.isdir(os.path.join(db_dir, d)))
    totals: Dict[str, int] = {}
    for sub in subs:
        sub_in, sub_out = os.path.join(db_dir, sub), os.path.join(save_dir, sub)
        ges_list = list(gestures) if gestures is not None else sorted(
            g for g in os.listdir(sub_in) if os.path.isdir(os.path.join(sub_in, g)))
        total = 0
        for ges in ges_list:
            g_dir = os.path.join(sub_in, ges)
            bin_num, gt = packing.read_joint(g_dir)
            pk = packing.pack_bin_files(packing.gesture_bin_paths(g_dir, bin_num))
            tsdf, max_l, mid_p, status = vox(pk, res, layout, device)
            pc = resample_point_clouds(pk, points_num, rng) if point_clouds else None
            write_gesture(sub_out, ges, np.asarray(tsdf, dtype), np.asarray(max_l, dtype),
                          np.asarray(mid_p, dtype), gt, status, pc, gt_3d)
            total += bin_num
            if verbose:
                print("%s-%s files saved." % (sub, ges))
        np.save(os.path.join(save_dir, "data_num-%s.npy" % sub), total)
        totals[sub] = total
    return totals
