"""Seeded synthetic MSRA-style depth frames (no dataset ships with the reference).

The frame format is the payload of an MSRA ``.bin`` file as the reference reads it
(``pre/read_MSRA.py:155-164``): a 6 x int32 header ``[W, H, left, top, right, bottom]``
and ``(right-left)*(bottom-top)`` float32 depths in millimetres, row-major over the
bounding box, 0 = background.  The generator follows SURVEY.md section 8(d): a
hemispherical blob (the "hand") at 300-600 mm in front of a 320x240 camera, 1 mm
Gaussian noise and 1 % dropped foreground pixels (holes).

Two distributions:
  * ``full``  - bbox = whole image (BASELINE.json configs[1], N = 76,800 px);
  * ``crop``  - MSRA-like bounding boxes, side U(90,160) px, placed inside the image.

Everything is numpy so the same frames can be produced here (goldens), in the tests
and on the GPU box (bench) from the seed alone.
"""
from __future__ import annotations

import numpy as np

IMG_W = 320
IMG_H = 240


def synth_frame(seed: int, kind: str = "full"):
    """One frame -> (header int32[6], depth float32[b_w*b_h])."""
    rng = np.random.default_rng(1234 + int(seed))
    if kind == "full":
        l, t, r, b = 0, 0, IMG_W, IMG_H
    elif kind == "crop":
        bw = int(rng.integers(90, 161))
        bh = int(rng.integers(90, 161))
        l = int(rng.integers(0, IMG_W - bw + 1))
        t = int(rng.integers(0, IMG_H - bh + 1))
        r, b = l + bw, t + bh
    else:
        raise ValueError("kind must be 'full' or 'crop'")
    bw, bh = r - l, b - t
    # Blob centre / radius in image pixels, kept inside the bbox.
    rad_hi = min(90.0, 0.48 * min(bw, bh))
    rad_lo = min(50.0, 0.6 * rad_hi)
    rad = float(rng.uniform(rad_lo, rad_hi))
    ecc = float(rng.uniform(0.75, 1.0))  # ellipse: shrink the y radius
    cx = float(rng.uniform(l + rad, r - rad))
    cy = float(rng.uniform(t + rad * ecc, b - rad * ecc))
    base = float(rng.uniform(300.0, 600.0))
    bulge = float(rng.uniform(20.0, 60.0))
    xs = np.arange(l, r, dtype=np.float64)[None, :]
    ys = np.arange(t, b, dtype=np.float64)[:, None]
    rr = ((xs - cx) / rad) ** 2 + ((ys - cy) / (rad * ecc)) ** 2
    inside = rr < 1.0
    depth = base - bulge * np.sqrt(np.clip(1.0 - rr, 0.0, 1.0))
    depth = depth + rng.normal(0.0, 1.0, size=depth.shape)
    holes = rng.random(depth.shape) < 0.01
    depth = np.where(inside & ~holes, depth, 0.0).astype(np.float32)
    header = np.array([IMG_W, IMG_H, l, t, r, b], dtype=np.int32)
    return header, np.ascontiguousarray(depth.reshape(-1))


def synth_variant(seed: int, bbox=(0, 0, IMG_W, IMG_H), base: float = 450.0, rad: float = 60.0, centre=None,
                  keep: float = 0.99, bulge: float = 40.0, ecc: float = 0.9, noise: float = 1.0, sign: str = "pos"):
    """One frame outside the two benchmark distributions -> (header int32[6], depth float32[b_w*b_h]): the same blob
    model as synth_frame with every parameter explicit, for the parity fixtures and the fuzzer.

    bbox    (left, top, right, bottom) inside the 320x240 image;
    base    depth of the blob's rim in mm (150 = a hand at the lens, 1500 = across the room);
    rad     x radius of the blob in pixels (``ecc`` shrinks the y radius); ``centre`` = (cx, cy) in image pixels, default
            a seeded position that keeps the blob inside the bbox (a centre near a bbox edge cuts the blob);
    keep    probability that a foreground pixel keeps its depth (0.01 = a very sparse crop, 1.0 = no holes);
    sign    "pos": depths as measured; "neg": all negated; "halves": the left half of the valid pixels negated;
            "checker": negated where (x + y) is odd.  The reference accepts any |d| >= 1 (pre/tsdf_numba.py:43,
            pre/tsdf_for.py:90) and maps d to z = -d, so a mixed-sign crop puts z = 0 INSIDE the grid: q = -F / v_z
            changes sign, and grows without bound, across it.
    """
    rng = np.random.default_rng(987654 + int(seed))
    l, t, r, b = (int(v) for v in bbox)
    if not (0 <= l < r <= IMG_W and 0 <= t < b <= IMG_H):
        raise ValueError("bbox must lie inside the 320x240 image")
    if centre is None:
        rx, ry = min(rad, 0.5 * (r - l)), min(rad * ecc, 0.5 * (b - t))
        cx = float(rng.uniform(l + rx, r - rx)) if r - l > 2 * rx else 0.5 * (l + r)
        cy = float(rng.uniform(t + ry, b - ry)) if b - t > 2 * ry else 0.5 * (t + b)
    else:
        cx, cy = float(centre[0]), float(centre[1])
    xs = np.arange(l, r, dtype=np.float64)[None, :]
    ys = np.arange(t, b, dtype=np.float64)[:, None]
    rr = ((xs - cx) / rad) ** 2 + ((ys - cy) / (rad * ecc)) ** 2
    depth = base - bulge * np.sqrt(np.clip(1.0 - rr, 0.0, 1.0))
    depth = depth + rng.normal(0.0, noise, size=depth.shape)
    kept = rng.random(depth.shape) < keep
    depth = np.where((rr < 1.0) & kept, depth, 0.0)
    if sign == "neg":
        depth = -depth
    elif sign == "halves":
        depth = np.where(xs + 0 * ys < cx, -depth, depth)
    elif sign == "checker":
        depth = np.where(((xs + ys).astype(np.int64) & 1) == 1, -depth, depth)
    elif sign != "pos":
        raise ValueError("sign must be pos, neg, halves or checker")
    header = np.array([IMG_W, IMG_H, l, t, r, b], dtype=np.int32)
    return header, np.ascontiguousarray(depth.astype(np.float32).reshape(-1))


def synth_batch(n: int, kind: str = "full", seed0: int = 0, threads: int = 1):
    """n frames packed back to back.

    Returns (depth float32[sum N_i], offsets int64[n+1], headers int32[n,6]) - exactly
    the three input arrays of ``tsdf_voxelize_hip`` (include/tsdf.h).  ``threads`` > 1
    generates the frames on a thread pool (numpy releases the GIL in the fills); every
    frame has its own seeded generator, so the result does not depend on it.
    """
    if threads > 1 and n > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(threads) as ex:
            frames = list(ex.map(lambda i: synth_frame(seed0 + i, kind), range(n)))
    else:
        frames = [synth_frame(seed0 + i, kind) for i in range(n)]
    headers = np.empty((n, 6), dtype=np.int32)
    offsets = np.zeros(n + 1, dtype=np.int64)
    for i, (h, d) in enumerate(frames):
        headers[i] = h
        offsets[i + 1] = offsets[i] + d.size
    depth = np.concatenate([d for _, d in frames]) if frames else np.zeros(0, dtype=np.float32)
    return depth, offsets, headers


def synth_msra_tree(root: str, n_sub: int = 4, n_ges: int = 5, n_frames: int = 2, seed: int = 0,
                    kind: str = "crop") -> int:
    """A small MSRA-shaped tree under ``root`` — ``<root>/P<s>/<g>/{joint.txt, 000000_depth.bin, ...}``
    (pre/read_MSRA.py:46-50,79,99) — made of seeded synthetic frames and labels, byte-identical wherever it is
    generated (the build container writes the reference-run fixtures from it, the tests regenerate it on the GPU box).
    Gesture directories are named ``1..n_ges`` (sorted as strings, like the reference's ``sorted(os.listdir())``);
    labels are written with 6 decimals, joints scattered about the frame's own hand (z = -depth).  Returns the frame
    count."""
    import os

    rng = np.random.default_rng(977 + int(seed))
    k = 0
    for s in range(n_sub):
        for g in range(n_ges):
            gdir = os.path.join(root, "P%d" % s, "%d" % (g + 1))
            os.makedirs(gdir, exist_ok=True)
            rows = []
            for i in range(n_frames):
                h, d = synth_frame(5000 + 100 * int(seed) + k, kind)
                with open(os.path.join(gdir, "%06d_depth.bin" % i), "wb") as f:
                    h.tofile(f)
                    d.tofile(f)
                valid = d[d != 0]
                zc = -float(valid.mean()) if valid.size else -400.0
                j = rng.normal(0.0, 45.0, (21, 3))
                j[:, 2] += zc
                rows.append(j.reshape(63))
                k += 1
            with open(os.path.join(gdir, "joint.txt"), "w") as f:
                f.write("%d\n" % n_frames)
                for row in rows:
                    f.write(" ".join("%.6f" % v for v in row) + "\n")
    return k
