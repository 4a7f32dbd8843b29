"""Drop-in for the reference class ``DataProcess`` (pre/process.py:4-28) — voxelization on the GPU.

    DataProcess(data, ground_truth, point_num=6000, aug=False).process()
        -> [point_clouds, tsdf, max_l, mid_p]                         # pre/process.py:13-28

``point_cloud`` (pre/process.py:30-68) and ``set_length`` (:70-84) are vectorised numpy here (the
reference loops over rows, columns and then every pixel in Python); ``tsdf_f`` / ``tsdf_cal``
(:86-200) run on the GPU.  ``process()`` keeps the reference's behaviour of placing the grid on
the AABB of the random ``point_num``-point resample (:16-17), so its TSDF is RNG-dependent exactly
like the reference's; call ``tsdf_f(self.point_cloud())`` (or the batched ``voxelize``) for the
deterministic all-pixel placement the numba path uses.

``aug=True`` (pre/process.py:19-24): the reference's ``data_aug`` raises AxisError on its own input and, where a
variant of it runs, only moves the grid while the TSDF still samples the un-augmented depth image (SURVEY.md
App. B#8-9) — there is no behaviour to keep, so it is RE-SPECIFIED (parity unpinned, like BASELINE configs[4]):
the same three draws in the same order from numpy's legacy generator (``np.random.seed(k)`` reproduces them:
``augment.reference_draw``), the stretch / rotation conventions pinned by tests/golden/aug_ref.npz, but ``rot_z`` is
used for R_z and the cloud is stretched and rotated about a real centre (the un-augmented grid centre).  The
result list has the reference's nine entries in its order (:23-24); ``tsdf_aug`` is the FUSED augmented
voxelization of the depth image (``voxelize_aug``: every valid pixel mapped, grid placed on the mapped cloud,
distances taken in the mapped frame), ``ground_truth_aug`` the joints under the same map.
"""
from __future__ import annotations

import numpy as np

from . import augment as _aug
from . import tsdf_for as _tf


class DataProcess(object):
    def __init__(self, data, ground_truth, point_num=6000, aug=False):
        self.fFocal_msra = 241.42
        self.data = data
        self.ground_truth = ground_truth
        self.point_num = point_num
        self.aug = aug

    def process(self):
        hand_points = self.point_cloud()
        point_clouds = self.set_length(hand_points)
        tsdf, max_l, mid_p = self.tsdf_f(point_clouds)
        if not self.aug:
            return [point_clouds, tsdf, max_l, mid_p]
        hand_aug, ground_truth_aug = self.data_aug(hand_points, centre=mid_p)
        point_clouds_aug = self.set_length(hand_aug)
        tsdf_aug, max_l_aug, mid_p_aug = self.tsdf_aug()
        return [point_clouds, tsdf, max_l, mid_p,
                point_clouds_aug, tsdf_aug, max_l_aug, mid_p_aug, ground_truth_aug]     # pre/process.py:23-24

    def data_aug(self, point_clouds, centre=None):
        """pre/process.py:202-261, re-specified (module docstring): ``(point_clouds_aug, ground_truth_aug)``.
        Draws from ``np.random`` in the reference's order; the map is kept in ``self.xform`` (float64[24]) so that
        :meth:`tsdf_aug` voxelizes with the same one."""
        pts = np.asarray(point_clouds, np.float64)
        if centre is None:
            pmax, pmin = self.max_min_point(pts)
            centre = (pmax + pmin) / 2
        rs = np.random.mtrand._rand        # the legacy global generator np.random.uniform / randint use
        self.xform = _aug.random_affines(np.asarray(centre, np.float64)[None, :], rng=rs)[0][0]
        joints = np.asarray(self.ground_truth, np.float64).reshape(1, -1, 3)
        ground_truth_aug = _aug.apply_affine(joints, self.xform[None]).reshape(-1, 63)
        hand_aug = _aug.apply_affine(pts[None], self.xform[None])[0]
        return hand_aug, ground_truth_aug

    def tsdf_aug(self, voxel_res: int = 32):
        """The augmented volume of this frame under ``self.xform``: ``(tsdf_aug float64[3,R,R,R] in [c,x,y,z],
        max_l_aug, mid_p_aug)`` — the same types as :meth:`tsdf_f` returns."""
        import torch

        from .voxelize import voxelize_aug
        header = np.ascontiguousarray(self.data["header"], dtype=np.int32).reshape(6)
        depth = np.ascontiguousarray(self.data["depth"], dtype=np.float32).reshape(-1)
        dev = _tf._device()
        out = voxelize_aug(torch.from_numpy(depth).to(dev), torch.tensor([0, depth.size], dtype=torch.int64, device=dev),
                           torch.from_numpy(header[None]).to(dev),
                           torch.from_numpy(np.ascontiguousarray(self.xform[None])).to(dev), res=voxel_res, layout="cxyz")
        return (out.tsdf[0].cpu().numpy().astype(np.float64), np.float32(out.max_l[0].item()),
                out.mid_p[0].cpu().numpy())

    def point_cloud(self):
        """pre/process.py:30-68: back-project every bbox pixel, keep the non-zero points.

        x and y are formed in float32 ((w + left - W/2) * d, then / focal in float64 on store),
        z = -depth; a point is kept if any coordinate is non-zero (:62-64)."""
        header = self.data["header"]
        depth = np.asarray(self.data["depth"])
        img_width, img_height = header[0], header[1]
        bb_left, bb_top, bb_right, bb_bottom = header[2], header[3], header[4], header[5]
        bb_height, bb_width = int(bb_bottom - bb_top), int(bb_right - bb_left)
        d = depth.reshape(bb_height, bb_width)
        # same operand types as the reference (float32 index vectors, numpy-scalar header fields), so
        # numpy's own promotion rules decide the working precision exactly as they do there
        w_matrix = np.arange(bb_width, dtype=np.float32)
        h_matrix = np.arange(bb_height, dtype=np.float32)
        x = np.multiply((w_matrix + bb_left - (img_width / 2))[None, :], d) / self.fFocal_msra
        y = -np.multiply((h_matrix + bb_top - (img_height / 2))[:, None], d) / self.fFocal_msra
        pts = np.zeros((bb_height * bb_width, 3))
        pts[:, 0] = x.reshape(-1)
        pts[:, 1] = y.reshape(-1)
        pts[:, 2] = -depth.reshape(-1)
        return pts[np.any(pts != 0, axis=1)]

    def set_length(self, hand_points):
        """pre/process.py:70-84: resample (with replacement) to exactly point_num points."""
        n = hand_points.shape[0]
        if n < self.point_num:
            idx = np.arange(0, self.point_num, 1, dtype=np.int32)
            idx[n:] = np.random.randint(0, n, size=self.point_num - n)
        else:
            idx = np.random.randint(0, n, size=self.point_num)
        return hand_points[idx, :]

    def max_min_point(self, point_cloud):
        return _tf.max_min_point(point_cloud)

    def tsdf_f(self, point_cloud):
        """pre/process.py:86-100."""
        return _tf.tsdf_f(self.data, point_cloud)

    def tsdf_cal(self, vox_ori, voxel_len, truncation):
        """pre/process.py:122-200."""
        return _tf.tsdf_cal(self.data, vox_ori, voxel_len, truncation)
