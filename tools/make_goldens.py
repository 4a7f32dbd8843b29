#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING the reference's own code.

Runs only in the build container (it imports /root/reference/pre/tsdf_for.py and
pre/process.py, which need nothing but numpy).  The reference never travels to the
GPU box; only the small .npz fixtures written here do.  Re-run with
    python tools/make_goldens.py
and commit the result; tests/test_oracle_golden.py consumes it.

What each fixture pins:
  loop32  pre/tsdf_for.py::tsdf_f -> tsdf_cal exactly as it runs today (numpy 2:
          float32 scalar arithmetic), fed a 2-point cloud {min_p, max_p} so that its
          own glue (tsdf_for.py:11-16) is evaluated on the full-pixel AABB.  Pins the
          grid placement (max_l, mid_p) and the per-voxel formula.
  loop64  the SAME reference function tsdf_cal fed float64-typed copies of the float32
          parameters and depth.  Every operation then runs in float64, which is what
          numba infers for pre/tsdf_numba.py:15-72 (FOCAL is a Python float) — this is
          the numba typing evaluated by the reference's own code.  The parity target.
  pc_*    pre/process.py::DataProcess.point_cloud + max_min_point on ALL valid points
          (no 6000-point resample): the CPU analogue of min_max_kernel.  It computes
          x,y in float32, so it may differ from the numba typing by 1 ulp (App. A.1);
          stored as a witness with that tolerance.
  aug_ref pre/process.py::DataProcess.data_aug run on [1,M,3] clouds with np.random seeded: pins the stretch /
          rotation conventions and the draw order that handposeestimation-with-3d-cnns_amd/augment.py restates.
  joint_nor_ref  pre/joint_nor.py::normalize (:8-18) RUN on [n,21,3] float32 labels (the one shape it does not raise
          on: for the [n,63] arrays its callers hold, `joint_nor[i] = ...` cannot broadcast (21,3) into (63,)) with the
          goldens' own max_l / mid_p plus seeded extra frames: pins the label formula of SURVEY.md 8(f)#4.
  aabb_*  the numba-typing AABB (pre/tsdf_numba.py:84-96,140-141) from
          oracle/tsdf_oracle_np.py — a restatement, not a run (min_max_kernel cannot
          be executed here: no usable numba, no params.py).
"""
from __future__ import annotations

import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_PRE = "/root/reference/pre"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF_PRE)

import tsdf_for  # noqa: E402  (the reference)
import process as ref_process  # noqa: E402  (the reference)
import joint_nor as ref_joint_nor  # noqa: E402  (the reference)

synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
from oracle import tsdf_oracle_np as onp  # noqa: E402


def special_frames():
    """Hand-built edge cases on top of the seeded generator."""
    out = []
    # bbox touching the image border, blob cut by the bbox edge
    h, d = synth.synth_frame(100, "full")
    l, t, r, b = 0, 0, 140, 120
    crop = d.reshape(240, 320)[t:b, l:r].copy()
    out.append(("border_cut", np.array([320, 240, l, t, r, b], np.int32), crop.reshape(-1)))
    # small odd-sized bbox (17 wide): rows are not 16-byte aligned
    h, d = synth.synth_frame(101, "full")
    img = d.reshape(240, 320)
    ys, xs = np.nonzero(img)
    cy, cx = int(ys.mean()), int(xs.mean())
    l, t = cx - 8, cy - 12
    r, b = l + 17, t + 25
    crop = img[t:b, l:r].copy()
    out.append(("small_odd", np.array([320, 240, l, t, r, b], np.int32), crop.reshape(-1)))
    return out


def run_reference(header, depth):
    nv, mn, mx = onp.aabb(depth, header)
    assert nv > 0
    data32 = {"header": header, "depth": depth}

    captured = {}
    orig = tsdf_for.tsdf_cal

    def spy(data, vox_ori, voxel_len, truncation):
        captured["p"] = (np.array(vox_ori), voxel_len, truncation)
        return orig(data, vox_ori, voxel_len, truncation)

    tsdf_for.tsdf_cal = spy
    try:
        pc2 = np.stack([mn, mx]).astype(np.float32)
        loop32, max_l, mid_p = tsdf_for.tsdf_f(data32, pc2)  # reference glue + loop, as it runs today
    finally:
        tsdf_for.tsdf_cal = orig
    vox_ori, voxel_len, truncation = captured["p"]
    assert vox_ori.dtype == np.float32 and np.asarray(voxel_len).dtype == np.float32

    # numba typing through the reference's own loop: float64-typed copies of float32 values
    data64 = {"header": header, "depth": depth.astype(np.float64)}
    loop64 = orig(data64, vox_ori.astype(np.float64), np.float64(voxel_len), np.float64(truncation))

    # CPU analogue of the AABB on all valid points (process.py:30-68,102-120)
    dp = ref_process.DataProcess({"header": header, "depth": depth}, np.zeros(63, np.float32))
    pts = dp.point_cloud()
    pmax, pmin = dp.max_min_point(pts)

    # loop layout [c,x,y,z] -> numba layout [c,z,y,x]
    l32 = np.ascontiguousarray(loop32.transpose(0, 3, 2, 1)).astype(np.float32)
    l64 = np.ascontiguousarray(loop64.transpose(0, 3, 2, 1)).astype(np.float32)
    assert np.array_equal(l32.astype(np.float64), loop32.transpose(0, 3, 2, 1)), "loop32 holds f32 values"
    return dict(
        header=header, depth=depth,
        aabb_min=mn, aabb_max=mx, n_valid=np.int64(nv),
        pc_min=pmin, pc_max=pmax, pc_n=np.int64(pts.shape[0]),
        max_l=np.float32(max_l), mid_p=np.asarray(mid_p, np.float32),
        vox_ori=vox_ori, voxel_len=np.float32(voxel_len), trunc=np.float32(truncation),
        loop32=l32, loop64=l64,
        n_flip=np.int64((np.abs(l32 - l64) > 1e-5).any(axis=0).sum()),
    )


def run_reference_aug(outdir):
    """aug_ref.npz: the reference's own DataProcess.data_aug (pre/process.py:202-261) run on [1,M,3] clouds —
    the one input shape it does not raise AxisError on — with np.random seeded.  Stores the
    inputs, the seeds and what the reference returned; nothing else (the draws are not observable from outside:
    tests re-draw them with augment.reference_draw and must reproduce these outputs)."""
    rng = np.random.default_rng(4242)
    seeds = np.array([0, 1, 7, 12345, 20261004], np.int64)
    M = 64
    pcs = rng.normal(0, 60, (len(seeds), 1, M, 3))
    pcs[..., 2] -= 400.0  # in front of the camera, z = -depth
    gts = rng.normal(0, 70, (len(seeds), 63))
    gts[:, 2::3] -= 400.0
    pc_aug = np.empty_like(pcs)
    gt_aug = np.empty((len(seeds), 63))
    for i, sd in enumerate(seeds):
        dp = ref_process.DataProcess({"header": np.array([320, 240, 0, 0, 4, 4], np.int32),
                                      "depth": np.zeros(16, np.float32)}, gts[i].copy(), aug=True)
        np.random.seed(int(sd))
        pa, ga = dp.data_aug(pcs[i].copy())
        pc_aug[i] = pa
        gt_aug[i] = ga.reshape(63)
    np.savez_compressed(os.path.join(outdir, "aug_ref.npz"), seeds=seeds, pc=pcs, gt=gts, pc_aug=pc_aug, gt_aug=gt_aug)
    print(f"aug_ref: {len(seeds)} seeded data_aug runs on [1,{M},3] clouds")


def run_reference_joint_nor(outdir, golden_scales):
    """joint_nor_ref.npz: the reference's own normalize (pre/joint_nor.py:8-18) run on float32 [n,21,3] labels.
    Frames 0..7 use the max_l / mid_p the reference's tsdf_f produced for the volume goldens (same order as
    MANIFEST.txt), the rest seeded scales; some joints lie outside the cube so that the clamp of
    3D_CNN/train.py:241-242 (applied by the test to these reference values) has something to do.  The reference
    stores its float32 results in a float64 array; they are kept as float64 here, untouched."""
    rng = np.random.default_rng(777)
    n_extra = 24
    max_l = np.concatenate([np.array([g[0] for g in golden_scales], np.float32),
                            rng.uniform(80, 400, n_extra).astype(np.float32)])
    mid_p = np.concatenate([np.stack([g[1] for g in golden_scales]).astype(np.float32),
                            (rng.normal(0, 80, (n_extra, 3)) + [0, 0, -420]).astype(np.float32)])
    n = len(max_l)
    gt = (mid_p[:, None, :] + rng.normal(0, 0.35, (n, 21, 3)) * max_l[:, None, None]).astype(np.float32)
    want = ref_joint_nor.normalize(gt.copy(), max_l, mid_p)
    assert want.shape == (n, 21, 3) and want.dtype == np.float64
    assert (want < 0).any() and (want > 1).any()
    np.savez_compressed(os.path.join(outdir, "joint_nor_ref.npz"), gt=gt, max_l=max_l, mid_p=mid_p, joint_nor=want)
    # the [n,63] form its callers hold (pre/joint_nor.py:42-45 feeds np.load of ground_truth/*.npy) does not run:
    try:
        ref_joint_nor.normalize(gt.reshape(n, 63).copy(), max_l, mid_p)
        raise AssertionError("normalize accepted [n,63]")
    except ValueError:
        pass
    print(f"joint_nor_ref: {n} frames x 21 joints through pre/joint_nor.py::normalize "
          f"({int((want < 0).sum())} coordinates < 0, {int((want > 1).sum())} > 1)")


def main():
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    run_reference_aug(outdir)
    frames = []
    for s in (0, 1, 2):
        h, d = synth.synth_frame(s, "full")
        frames.append((f"full_{s}", h, d))
    for s in (10, 11, 12):
        h, d = synth.synth_frame(s, "crop")
        frames.append((f"crop_{s}", h, d))
    frames += special_frames()
    names = []
    scales = []
    for name, h, d in frames:
        g = run_reference(h, d)
        scales.append((g["max_l"], g["mid_p"]))
        np.savez_compressed(os.path.join(outdir, f"{name}.npz"), **g)
        names.append(name)
        print(f"{name}: bbox {h[4]-h[2]}x{h[5]-h[3]} valid {int(g['n_valid'])} max_l {float(g['max_l']):.4f} "
              f"flips(f32 vs f64 loop) {int(g['n_flip'])} "
              f"|loop32-loop64|max(non-flip) "
              f"{float(np.abs(g['loop32']-g['loop64'])[np.abs(g['loop32']-g['loop64'])<=1e-5].max()):.2e}")
    run_reference_joint_nor(outdir, scales)
    with open(os.path.join(outdir, "MANIFEST.txt"), "w") as f:
        f.write("# written by tools/make_goldens.py from /root/reference/pre/{tsdf_for,process,joint_nor}.py\n")
        f.write("# numpy %s\n" % np.__version__)
        for n in names:
            f.write(n + "\n")


if __name__ == "__main__":
    import argparse

    argparse.ArgumentParser(description="Regenerate tests/golden/*.npz by running the reference's own code "
                                        "(needs /root/reference; no options).").parse_args()
    main()
