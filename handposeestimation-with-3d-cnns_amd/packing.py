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
import struct
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

HEADER_INTS = 6

# One subject (or any set of frames) as ONE file: what pre/read_MSRA.py:98-106 re-reads from ~8,500 tiny .bin files
# and a joint.txt per gesture, laid out as the arrays the HIP entry consumes, so that every epoch after the first is
# a memory map instead of 76 k open() calls.  Layout (little endian, every array 64-byte aligned):
#   0   magic "TSDFPK01"
#   8   int64 n, int64 n_px, int64 gt_cols (0: no labels), int64 n_groups, 3 x int64 reserved
#   64  headers int32[n,6] | offsets int64[n+1] | gt float32[n,gt_cols] | group_start int64[n_groups+1] |
#       group names (utf-8, '\n'-joined, length-prefixed int64) | depth float32[n_px]
# "groups" are the gestures of a subject in file order (group_start[g] .. group_start[g+1] are its frames).
PACK_MAGIC = b"TSDFPK01"
_ALIGN = 64


def _gather_threads() -> int:
    """Workers of the shuffled-batch gather: the cores this process may use, at most 16 (a one-GPU box's CPU share);
    8 threads moved 64 MB per batch at 43 GB/s — below the link's 56 GB/s, so the gather, not the upload, bounded the
    host-fed shuffled loader."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


_GATHER_THREADS = _gather_threads()


def _native_gather(pk, idx: np.ndarray, out: np.ndarray, off: np.ndarray) -> bool:
    """The shuffled-batch gather by ``tsdf_host_gather_frames`` (threads, one memcpy per frame instead of one numpy slice
    assignment per frame: ~10 us of interpreter each).  False when the library is not built or the arrays are not plain
    float32 / int64 memory — the caller then copies frame by frame in numpy.  Host memory only: not a compute path."""
    if idx.size < 4:
        return False
    try:
        from . import _lib
        L = _lib.load()
    except (ImportError, OSError):
        return False
    src, so = pk.depth, pk.offsets
    if not (isinstance(src, np.ndarray) and src.dtype == np.float32 and src.flags.c_contiguous
            and isinstance(so, np.ndarray) and so.dtype == np.int64 and so.flags.c_contiguous
            and out.dtype == np.float32 and out.flags.c_contiguous and out.flags.writeable):
        return False
    idx = np.ascontiguousarray(idx, np.int64)
    off2 = np.empty_like(off)
    rc = L.tsdf_host_gather_frames_n(src.ctypes.data, src.size, so.ctypes.data, so.size - 1, idx.ctypes.data, idx.size,
                                     out.ctypes.data, out.size, off2.ctypes.data, _GATHER_THREADS)
    if rc != 0:
        raise ValueError("pack offsets do not describe its depth payload (damaged pack?)")
    return bool((off2 == off).all())


def _pad(n: int) -> int:
    return (n + _ALIGN - 1) // _ALIGN * _ALIGN


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
    gt: Optional[np.ndarray] = None            # float32[n, 63] labels (joint.txt rows), when known
    group_start: Optional[np.ndarray] = None   # int64[g+1]: frame ranges of the gestures, when packed from a tree
    group_names: Optional[List[str]] = None

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
                            self.headers[a:b], None if self.gt is None else self.gt[a:b])

    def take(self, idx: np.ndarray, depth_out: Optional[np.ndarray] = None) -> "PackedFrames":
        """Frames ``idx`` (any order, e.g. a shuffled batch) as a packed batch.  ``depth_out`` (float32, large
        enough) receives the depths — e.g. a pinned staging buffer — so the gather is the only copy."""
        idx = np.asarray(idx, dtype=np.int64)
        if idx.size and (np.diff(idx) == 1).all():  # contiguous: one slice, no gather
            a, b = int(idx[0]), int(idx[-1]) + 1
            sub = self.slice(a, b)
            if depth_out is not None:
                depth_out[: sub.depth.size] = sub.depth
                sub.depth = depth_out[: sub.depth.size]
            return sub
        lens = self.offsets[idx + 1] - self.offsets[idx]
        off = np.zeros(idx.size + 1, np.int64)
        np.cumsum(lens, out=off[1:])
        total = int(off[-1])
        out = depth_out[:total] if depth_out is not None else np.empty(total, np.float32)
        if not _native_gather(self, idx, out, off):
            for k, i in enumerate(idx):  # n slice copies (the index arithmetic above is vectorised)
                out[off[k]:off[k + 1]] = self.depth[self.offsets[i]:self.offsets[i + 1]]
        return PackedFrames(out, off, self.headers[idx], None if self.gt is None else self.gt[idx])

    def pin(self) -> "PackedFrames":
        """Move the depth payload into page-locked host memory (once): contiguous batches can then be uploaded by
        DMA straight out of the pack, with no staging copy (``dataset.VoxelLoader``).  Needs torch with a GPU."""
        import torch

        if getattr(self, "_pinned", None) is None:
            t = torch.empty(int(self.depth.size), dtype=torch.float32).pin_memory()
            t.numpy()[:] = self.depth
            self._pinned = t
            self.depth = t.numpy()
        return self

    # ---- one-file blob ----
    def save(self, path: str) -> None:
        """Write the pack as one ``TSDFPK01`` file (layout at the top of this module)."""
        n = len(self)
        headers = np.ascontiguousarray(self.headers, np.int32).reshape(n, HEADER_INTS)
        offsets = np.ascontiguousarray(self.offsets, np.int64)
        depth = np.ascontiguousarray(self.depth, np.float32)
        if offsets.shape != (n + 1,) or offsets[0] != 0 or offsets[-1] != depth.size:
            raise ValueError("offsets do not describe the depth buffer")
        gt = None if self.gt is None else np.ascontiguousarray(self.gt, np.float32)
        if gt is not None:
            gt = gt.reshape(n, gt.size // n if n else (gt.shape[-1] if gt.ndim > 1 else 0))
        gs = np.ascontiguousarray(self.group_start if self.group_start is not None else [0, n], np.int64)
        names = "\n".join(self.group_names or [""] * (gs.size - 1)).encode()
        tmp = path + ".tmp"
        with open(tmp, "wb") as f:
            f.write(PACK_MAGIC)
            f.write(struct.pack("<7q", n, depth.size, 0 if gt is None else gt.shape[1], gs.size - 1, 0, 0, 0))
            for arr in (headers, offsets, gt, gs, np.frombuffer(struct.pack("<q", len(names)) + names, np.uint8), depth):
                f.write(b"\0" * (_pad(f.tell()) - f.tell()))
                if arr is not None:
                    arr.tofile(f)
        os.replace(tmp, path)

    @staticmethod
    def load(path: str, mmap: bool = True) -> "PackedFrames":
        """Open a ``TSDFPK01`` file.  ``mmap=True`` maps the arrays (nothing is read until used, pages are shared
        between loader processes); ``mmap=False`` reads them into memory."""
        with open(path, "rb") as f:
            head = f.read(64)
        if head[:8] != PACK_MAGIC:
            raise ValueError(f"{path}: not a TSDFPK01 pack")
        n, n_px, gt_cols, n_groups = struct.unpack("<4q", head[8:40])
        size = os.path.getsize(path)

        def arr(pos, dtype, shape):
            count = int(np.prod(shape))
            nbytes = count * np.dtype(dtype).itemsize
            if pos + nbytes > size:
                raise ValueError(f"{path}: truncated pack")
            if count == 0:
                a = np.zeros(shape, dtype)
            elif mmap:
                a = np.memmap(path, dtype=dtype, mode="r", offset=pos, shape=shape)
            else:
                a = np.fromfile(path, dtype=dtype, count=count, offset=pos).reshape(shape)
            return a, _pad(pos + nbytes)

        pos = 64
        headers, pos = arr(pos, np.int32, (n, HEADER_INTS))
        offsets, pos = arr(pos, np.int64, (n + 1,))
        gt = None
        if gt_cols:
            gt, pos = arr(pos, np.float32, (n, gt_cols))
        gs, pos = arr(pos, np.int64, (n_groups + 1,))
        ln, _ = arr(pos, np.int64, (1,))
        raw, pos2 = arr(pos + 8, np.uint8, (int(ln[0]),))
        names = bytes(np.asarray(raw)).decode().split("\n") if n_groups else []
        pos = _pad(pos + 8 + int(ln[0]))
        depth, _ = arr(pos, np.float32, (n_px,))
        if n and (int(offsets[0]) != 0 or int(offsets[-1]) != n_px):
            raise ValueError(f"{path}: offsets do not match the depth payload")
        if n:   # every interior offset too, and every header against its payload (vectorised; reads n*32 bytes)
            o = np.asarray(offsets)
            if (np.diff(o) < 0).any():
                raise ValueError(f"{path}: offsets are not non-decreasing (damaged pack)")
            h = np.asarray(headers).astype(np.int64)
            if ((h[:, 4] - h[:, 2]) * (h[:, 5] - h[:, 3]) != np.diff(o)).any() or (h[:, 4] <= h[:, 2]).any() \
                    or (h[:, 5] <= h[:, 3]).any():
                raise ValueError(f"{path}: a header contradicts its payload (damaged pack)")
        return PackedFrames(depth, offsets, headers, gt, np.asarray(gs), names)

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


def _read_bin_raw(path: str) -> np.ndarray:
    return np.fromfile(path, dtype=np.uint8)


def pack_bin_files_fast(paths: Sequence[str], threads: int = 8) -> PackedFrames:
    """:func:`pack_bin_files` for many files: the files are read as raw bytes by a small thread pool (file I/O
    releases the GIL) and headers / payload sizes are validated and laid out with vectorised numpy — no per-frame
    Python arithmetic beyond the copy into the packed buffer."""
    n = len(paths)
    if n == 0:
        return pack_frames([])
    with ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        raws = list(ex.map(_read_bin_raw, paths))
    sizes = np.array([r.size for r in raws], np.int64)
    if (sizes < 4 * HEADER_INTS).any():
        raise ValueError(f"{paths[int(np.argmax(sizes < 4 * HEADER_INTS))]}: truncated header")
    headers = np.stack([r[: 4 * HEADER_INTS].view(np.int32) for r in raws])
    h64 = headers.astype(np.int64)
    bw, bh = h64[:, 4] - h64[:, 2], h64[:, 5] - h64[:, 3]
    npx = bw * bh
    bad = (bw <= 0) | (bh <= 0) | (4 * npx != sizes - 4 * HEADER_INTS)
    if bad.any():
        i = int(np.argmax(bad))
        raise ValueError(f"{paths[i]}: bbox {headers[i, 2:6].tolist()} needs {int(npx[i])} depths, "
                         f"file holds {(int(sizes[i]) - 24) // 4}")
    offsets = np.zeros(n + 1, np.int64)
    np.cumsum(npx, out=offsets[1:])
    depth = np.empty(int(offsets[-1]), np.float32)
    dview = depth.view(np.uint8)
    for i, r in enumerate(raws):
        dview[4 * offsets[i]:4 * offsets[i + 1]] = r[4 * HEADER_INTS:]
    return PackedFrames(depth, offsets, np.ascontiguousarray(headers))


def pack_subject(sub_dir: str, gestures: Optional[Sequence[str]] = None, threads: int = 8) -> PackedFrames:
    """One MSRA subject directory ``<sub>/<gesture>/{joint.txt, 000000_depth.bin, ...}`` -> one pack with labels
    and gesture boundaries (the per-gesture loop of pre/read_MSRA.py:78-106, minus the voxelization)."""
    ges = list(gestures) if gestures is not None else sorted(
        g for g in os.listdir(sub_dir) if os.path.isdir(os.path.join(sub_dir, g)))
    packs, gts, starts = [], [], [0]
    for g in ges:
        g_dir = os.path.join(sub_dir, g)
        bin_num, gt = read_joint(g_dir)
        packs.append(pack_bin_files_fast(gesture_bin_paths(g_dir, bin_num), threads))
        gts.append(gt)
        starts.append(starts[-1] + bin_num)
    if not packs:
        pk = pack_frames([])
        pk.gt, pk.group_start, pk.group_names = np.zeros((0, 63), np.float32), np.zeros(1, np.int64), []
        return pk
    offs = [np.zeros(1, np.int64)]
    base = 0
    for pk in packs:
        offs.append(pk.offsets[1:] + base)
        base += int(pk.offsets[-1])
    return PackedFrames(np.concatenate([pk.depth for pk in packs]), np.concatenate(offs),
                        np.concatenate([pk.headers for pk in packs]), np.concatenate(gts).astype(np.float32),
                        np.asarray(starts, np.int64), ges)


def pack_tree(db_dir: str, out_dir: str, subjects: Optional[Sequence[str]] = None, threads: int = 8,
              overwrite: bool = False) -> Dict[str, str]:
    """Pack every subject of an MSRA tree into ``<out_dir>/<subject>.tsdfpk`` (skipping packs that exist unless
    ``overwrite``).  Returns {subject: path}.  Run once; ``dataset.MSRADepthDataset(packed_dir=...)`` then maps them."""
    os.makedirs(out_dir, exist_ok=True)
    subs = list(subjects) if subjects is not None else sorted(
        d for d in os.listdir(db_dir) if os.path.isdir(os.path.join(db_dir, d)))
    out = {}
    for sub in subs:
        path = os.path.join(out_dir, sub + ".tsdfpk")
        if overwrite or not os.path.exists(path):
            pack_subject(os.path.join(db_dir, sub), threads=threads).save(path)
        out[sub] = path
    return out


def gesture_bin_paths(gesture_dir: str, bin_num: Optional[int] = None) -> List[str]:
    """``000000_depth.bin`` ... in frame order (pre/read_MSRA.py:99)."""
    if bin_num is None:
        bin_num, _ = read_joint(gesture_dir)
    return [os.path.join(gesture_dir, "%06d_depth.bin" % i) for i in range(bin_num)]
