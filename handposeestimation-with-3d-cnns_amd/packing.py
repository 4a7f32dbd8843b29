"""Frame I/O and batch packing (host side, numpy only).

The wire format the HIP entry point consumes (include/tsdf.h) is three arrays:
``depth float32[sum N_i]`` (bounding-box crops packed back to back), ``offsets int64[n+1]``
and ``headers int32[n,6]``.  This module builds them from MSRA ``.bin`` files — the format
the reference reads one file at a time in ``pre/read_MSRA.py:155-164`` (6 x int32 header
``[W, H, left, top, right, bottom]`` then ``(right-left)*(bottom-top)`` float32 depths) —
and from in-memory frames.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

HEADER_INTS = 6


def read_bin(f_name: str) -> Tuple[np.ndarray, np.ndarray]:
    """One MSRA depth file -> (header int32[6], depth float32[N]).

    Same return value as the reference's ``read_bin`` (pre/read_MSRA.py:155-164), but the file
    is opened in binary mode (the reference opens it as text) and a
    payload that does not match the header's bounding box is an error instead of garbage.
    """
    with open(f_name, "rb") as f:
        header = np.fromfile(f, dtype=np.int32, count=HEADER_INTS)
        depth = np.fromfile(f, dtype=np.float32)
    if header.size != HEADER_INTS:
        raise ValueError(f"{f_name}: truncated header ({header.size} of 6 int32)")
    n = (int(header[4]) - int(header[2])) * (int(header[5]) - int(header[3]))
    if n != depth.size:
        raise ValueError(f"{f_name}: bbox {header[2:6].tolist()} needs {n} depths, file holds {depth.size}")
    return header, depth


def write_bin(f_name: str, header: np.ndarray, depth: np.ndarray) -> None:
    """Inverse of :func:`read_bin` (used to build test fixtures and synthetic datasets)."""
    with open(f_name, "wb") as f:
        np.asarray(header, dtype=np.int32).tofile(f)
        np.asarray(depth, dtype=np.float32).tofile(f)


def read_joint(f_dir: str) -> Tuple[int, np.ndarray]:
    """``joint.txt`` of one gesture -> (frame count, ground truth float32[n,63]).

    Mirrors pre/read_MSRA.py:143-152: first line = number of frames, then one row of
    21 x (x,y,z) millimetre coordinates per frame.
    """
    f_name = os.path.join(f_dir, "joint.txt")
    with open(f_name, "r") as f:
        bin_num = int(f.readline())
    gt = np.loadtxt(f_name, dtype=np.float32, skiprows=1, ndmin=2)
    if gt.shape[0] != bin_num or gt.shape[1] != 63:
        raise ValueError(f"{f_name}: expected [{bin_num},63] joints, got {gt.shape}")
    return bin_num, gt


@dataclass
class PackedFrames:
    """n frames in the layout of ``tsdf_voxelize_hip``: depth, offsets[n+1], headers[n,6]."""

    depth: np.ndarray
    offsets: np.ndarray
    headers: np.ndarray

    def __len__(self) -> int:
        return int(self.headers.shape[0])

    @property
    def pixels(self) -> np.ndarray:
        """Pixels per frame (int64[n])."""
        return np.diff(self.offsets)

    def slice(self, a: int, b: int) -> "PackedFrames":
        """Frames [a, b) as a self-contained packed batch (offsets rebased to 0)."""
        a, b = int(a), int(b)
        off = self.offsets[a:b + 1] - self.offsets[a]
        return PackedFrames(self.depth[self.offsets[a]:self.offsets[b]], off.astype(np.int64),
                            self.headers[a:b])

    def frame(self, i: int) -> Tuple[np.ndarray, np.ndarray]:
        return self.headers[i], self.depth[self.offsets[i]:self.offsets[i + 1]]

    def to_torch(self, device, pin: bool = False, non_blocking: bool = False):
        """(depth, offsets, headers) as torch tensors on ``device`` (one H2D copy each)."""
        import torch

        ts = []
        for a in (self.depth, self.offsets, self.headers):
            t = torch.from_numpy(np.ascontiguousarray(a))
            if pin:
                t = t.pin_memory()
            ts.append(t.to(device, non_blocking=non_blocking))
        return tuple(ts)


def pack_frames(frames: Iterable[Tuple[np.ndarray, np.ndarray]]) -> PackedFrames:
    """[(header, depth), ...] -> PackedFrames.  Validates every header against its payload."""
    hs: List[np.ndarray] = []
    ds: List[np.ndarray] = []
    for i, (h, d) in enumerate(frames):
        h = np.asarray(h, dtype=np.int32).reshape(-1)
        d = np.asarray(d, dtype=np.float32).reshape(-1)
        if h.size != HEADER_INTS:
            raise ValueError(f"frame {i}: header must have 6 int32")
        n = (int(h[4]) - int(h[2])) * (int(h[5]) - int(h[3]))
        if int(h[4]) <= int(h[2]) or int(h[5]) <= int(h[3]) or n != d.size:
            raise ValueError(f"frame {i}: bbox {h[2:6].tolist()} does not match {d.size} depths")
        hs.append(h)
        ds.append(d)
    n = len(hs)
    headers = np.stack(hs) if n else np.zeros((0, HEADER_INTS), np.int32)
    offsets = np.zeros(n + 1, dtype=np.int64)
    if n:
        offsets[1:] = np.cumsum([d.size for d in ds])
    depth = np.concatenate(ds) if n else np.zeros(0, np.float32)
    return PackedFrames(depth, offsets, headers)


def pack_bin_files(paths: Sequence[str]) -> PackedFrames:
    """Read many ``*_depth.bin`` files into one packed batch (the reference reads and processes
    them one by one inside its frame loop, pre/read_MSRA.py:98-106)."""
    return pack_frames(read_bin(p) for p in paths)


def gesture_bin_paths(gesture_dir: str, bin_num: Optional[int] = None) -> List[str]:
    """``000000_depth.bin`` ... in frame order (pre/read_MSRA.py:99)."""
    if bin_num is None:
        bin_num, _ = read_joint(gesture_dir)
    return [os.path.join(gesture_dir, "%06d_depth.bin" % i) for i in range(bin_num)]
