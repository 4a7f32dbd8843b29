"""On-the-fly MSRA dataset: depth crops in, voxel grids out of the GPU — no preprocessed npz files.

The reference voxelizes every frame offline (pre/read_MSRA.py:37-140, ~0.2 s per frame), writes
``result/<subject>/TSDF/<gesture>.npz`` and then loads *all* of it into host RAM
(3D_CNN/dataset.py:35-38,99-117: ~76 k frames x 393 KB).  Here the dataset holds the raw frames
(``header``, ``depth`` crop, ``gt``) — straight from the ``.bin`` files, or, after one packing pass
(``packing.pack_tree``), from one memory-mapped ``.tsdfpk`` file per subject — and the loader packs a
batch, uploads it once and calls the HIP voxelizer on the training stream; what comes out is the tuple
the reference's ``__getitem__`` returns (3D_CNN/dataset.py:73-79), batched and already on the GPU:
``(tsdf[n,3,R,R,R], gt[n,63], max_l[n], mid_p[n,3])``.

Kept from the reference: directory layout ``<root>/<subject>/<gesture>/{joint.txt, 000000_depth.bin..}``
(pre/read_MSRA.py:46-50,79,99), leave-one-subject-out split (3D_CNN/dataset.py:44-53) and the
``small`` subset of 4 subjects x 5 gestures (:26-31).  Fixed: ``opt`` is honoured instead of being
ignored (:20-22).

Label sign convention (the "z flip", 3D_CNN/dataset.py:107-109).  MSRA's joint.txt stores joints in the
camera frame this pipeline uses throughout — the camera looks down -z, a point's z is MINUS its depth
(pre/process.py:46,58; pre/tsdf_numba.py:94) — so ``mid_p`` and the labels agree as they come from the file and
nothing is flipped here.  The reference reader negates z only for 3-D ``[n,21,3]`` label arrays, which pairs with
writer lines that are commented out (pre/read_MSRA.py:81-82: they would have stored z pre-negated); for the
``[n,63]`` arrays its writer really saves, the reader's branch is skipped (and ``g_t`` is then undefined — it
crashes, App. B#11).  ``export.write_gesture(gt_3d=True)`` therefore stores z pre-negated, so that the reference
reader's flip restores the camera-frame sign.
"""
from __future__ import annotations

import os
import sys
import queue
import threading
from typing import Iterator, List, NamedTuple, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.utils.data as data

from . import packing, shard
from .voxelize import (TsdfBatch, denormalize_joints, normalize_joints, voxelize, voxelize_indexed,  # noqa: F401
                       voxelize_labels)


def _subset(size: str) -> Tuple[int, int]:
    if size == "full":
        return 9, 17
    if size == "small":
        return 4, 5
    raise ValueError("size must be 'full' or 'small'")


class MSRADepthDataset(data.Dataset):
    """Raw MSRA frames: ``__getitem__ -> (header int32[6], depth float32[N], gt float32[63])``.

    root_path   the MSRA tree ``<root>/<subject>/<gesture>/...``
    packed_dir  directory of ``<subject>.tsdfpk`` packs (``packing.pack_tree``).  Given, frames come from the
                memory-mapped packs — one ``open()`` per subject instead of one per frame per epoch
                (pre/read_MSRA.py:98-106 re-reads ~8,500 files per subject); missing packs are built on first use
                when ``build_packs`` is true.
    """

    def __init__(self, root_path: str, train: bool = True, test_idx: int = 2, size: str = "full",
                 subjects: Optional[Sequence[str]] = None, packed_dir: Optional[str] = None,
                 build_packs: bool = True):
        n_sub, n_ges = _subset(size)
        self.root_path = root_path
        self.train = train
        self.test_idx = test_idx
        if subjects is not None:
            all_sub = list(subjects)
        elif root_path is not None and os.path.isdir(root_path):
            # subject directories (MSRA: P0..P8) — not a directory of packs that happens to live inside the tree
            skip = os.path.abspath(packed_dir) if packed_dir is not None else None
            all_sub = sorted(d for d in os.listdir(root_path) if os.path.isdir(os.path.join(root_path, d))
                             and os.path.abspath(os.path.join(root_path, d)) != skip)[:n_sub]
        elif packed_dir is not None:
            all_sub = sorted(f[:-7] for f in os.listdir(packed_dir) if f.endswith(".tsdfpk"))[:n_sub]
        else:
            raise ValueError("root_path is not a directory and no packed_dir was given")
        if not 0 <= test_idx < len(all_sub):
            raise ValueError("test_idx out of range")
        chosen = [s for i, s in enumerate(all_sub) if (i != test_idx) == train]
        self.subjects = chosen
        self.paths: List[str] = []
        self.packs: List[packing.PackedFrames] = []
        self._pack_of = np.zeros(0, np.int32)     # frame -> index into self.packs
        self._local = np.zeros(0, np.int64)       # frame -> frame index inside that pack
        gts: List[np.ndarray] = []
        if packed_dir is not None:
            pk_of, loc = [], []
            for sub in chosen:
                path = os.path.join(packed_dir, sub + ".tsdfpk")
                if not os.path.exists(path):
                    if not build_packs:
                        raise FileNotFoundError(path)
                    packing.pack_tree(root_path, packed_dir, subjects=[sub])
                pk = packing.PackedFrames.load(path, mmap=True)
                gs = pk.group_start if pk.group_start is not None else np.array([0, len(pk)])
                end = int(gs[min(n_ges, len(gs) - 1)])   # the first n_ges gestures (3D_CNN/dataset.py:26-31)
                self.packs.append(pk)
                pk_of.append(np.full(end, len(self.packs) - 1, np.int32))
                loc.append(np.arange(end, dtype=np.int64))
                gts.append(np.asarray(pk.gt[:end]) if pk.gt is not None else np.zeros((end, 63), np.float32))
            if pk_of:
                self._pack_of, self._local = np.concatenate(pk_of), np.concatenate(loc)
        else:
            for sub in chosen:
                sub_dir = os.path.join(root_path, sub)
                gestures = sorted(g for g in os.listdir(sub_dir) if os.path.isdir(os.path.join(sub_dir, g)))
                for ges in gestures[:n_ges]:
                    g_dir = os.path.join(sub_dir, ges)
                    bin_num, gt = packing.read_joint(g_dir)
                    self.paths += packing.gesture_bin_paths(g_dir, bin_num)
                    gts.append(gt)
        self.ground_truth = np.concatenate(gts) if gts else np.zeros((0, 63), np.float32)

    @classmethod
    def from_packs(cls, packs: Sequence[packing.PackedFrames]) -> "MSRADepthDataset":
        """A dataset over packs that are already in memory (no directory walk, no split)."""
        self = cls.__new__(cls)
        self.root_path, self.train, self.test_idx, self.subjects, self.paths = None, True, -1, [], []
        self.packs = list(packs)
        self._pack_of = np.concatenate([np.full(len(pk), k, np.int32) for k, pk in enumerate(self.packs)])
        self._local = np.concatenate([np.arange(len(pk), dtype=np.int64) for pk in self.packs])
        self.ground_truth = np.concatenate([np.asarray(pk.gt) if pk.gt is not None else np.zeros((len(pk), 63), np.float32)
                                            for pk in self.packs])
        return self

    @property
    def packed(self) -> bool:
        return bool(self.packs)

    def __len__(self) -> int:
        return len(self._local) if self.packed else len(self.paths)

    def pixels(self) -> Optional[np.ndarray]:
        """Pixels per frame (int64[n]) when known without touching the files (packed datasets)."""
        if not self.packed:
            return None
        out = np.empty(len(self), np.int64)
        for k, pk in enumerate(self.packs):
            m = self._pack_of == k
            out[m] = pk.pixels[self._local[m]]
        return out

    def __getitem__(self, index: int):
        if self.packed:
            pk = self.packs[int(self._pack_of[index])]
            h, d = pk.frame(int(self._local[index]))
            return np.asarray(h), np.asarray(d), self.ground_truth[index]
        header, depth = packing.read_bin(self.paths[index])
        return header, depth, self.ground_truth[index]

    def pin_packs(self, limit_bytes: int = 32 << 30) -> bool:
        """Page-lock the packs' depth payloads (if they fit ``limit_bytes``) so that contiguous batches upload
        without a staging copy.  Returns whether the packs are pinned."""
        if not self.packed or sum(int(pk.depth.size) * 4 for pk in self.packs) > limit_bytes:
            return False
        for pk in self.packs:
            pk.pin()
        return True

    def contiguous_source(self, idx: np.ndarray):
        """(pinned float32 tensor slice, PackedFrames view) when ``idx`` is a run of consecutive frames of one
        pinned pack, else None."""
        idx = np.asarray(idx, np.int64)
        if not self.packed or idx.size == 0 or not (np.diff(idx) == 1).all():
            return None
        ks = self._pack_of[idx]
        pk = self.packs[int(ks[0])]
        if ks[0] != ks[-1] or getattr(pk, "_pinned", None) is None:
            return None
        a, b = int(self._local[idx[0]]), int(self._local[idx[-1]]) + 1
        view = pk.slice(a, b)
        view.gt = self.ground_truth[idx]
        return pk._pinned[int(pk.offsets[a]):int(pk.offsets[b])], view

    def take(self, idx: np.ndarray, depth_out: Optional[np.ndarray] = None) -> packing.PackedFrames:
        """Frames ``idx`` as one packed batch (with labels).  Packed datasets gather straight from the memory
        maps (one slice when the indices are consecutive inside a subject); otherwise the files are read."""
        idx = np.asarray(idx, np.int64)
        if self.packed and idx.size:
            ks = self._pack_of[idx]
            if (ks == ks[0]).all():
                out = self.packs[int(ks[0])].take(self._local[idx], depth_out)
                out.gt = self.ground_truth[idx]
                return out
            pk = packing.pack_frames(self.packs[int(k)].frame(int(l)) for k, l in zip(ks, self._local[idx]))
            if depth_out is not None:
                depth_out[: pk.depth.size] = pk.depth
                pk.depth = depth_out[: pk.depth.size]
            pk.gt = self.ground_truth[idx]
            return pk
        pk = packing.pack_frames(self[int(i)][:2] for i in idx)
        if depth_out is not None:
            depth_out[: pk.depth.size] = pk.depth
            pk.depth = depth_out[: pk.depth.size]
        pk.gt = self.ground_truth[idx] if idx.size else np.zeros((0, 63), np.float32)
        return pk


def collate_frames(batch) -> Tuple[packing.PackedFrames, np.ndarray]:
    """[(header, depth, gt), ...] -> (PackedFrames, gt float32[n,63]) on the host."""
    pk = packing.pack_frames((h, d) for h, d, _ in batch)
    gt = np.stack([g for _, _, g in batch]).astype(np.float32) if batch else np.zeros((0, 63), np.float32)
    return pk, gt


def voxelize_batch(pk: packing.PackedFrames, gt: np.ndarray, device, res: int = 32, pin: bool = True):
    """Upload one packed batch and voxelize it on the current stream of ``device``.

    Returns ``(tsdf, gt, max_l, mid_p, status)`` as GPU tensors — the first four are the reference's
    per-item tuple (3D_CNN/dataset.py:73-79) batched."""
    depth, offsets, headers = pk.to_torch(device, pin=pin, non_blocking=True)
    tgt = torch.from_numpy(np.ascontiguousarray(gt))
    if pin:
        tgt = tgt.pin_memory()
    tgt = tgt.to(device, non_blocking=True)
    out: TsdfBatch = voxelize(depth, offsets, headers, res=res)
    return out.tsdf, tgt, out.max_l, out.mid_p, out.status


class VoxelBatch(NamedTuple):
    """What :class:`VoxelLoader` yields.  The first four fields are the reference's per-item tuple
    (3D_CNN/dataset.py:73-79) batched on the GPU; ``status`` tells degenerate / malformed frames apart (the
    reference returns None for them, tsdf_numba.py:162-171), ``gt_nor`` are the labels in the cube's [0,1] frame
    (pre/joint_nor.py:8-18 + the clamp of 3D_CNN/train.py:241-242), written by the voxelizer's own launch."""

    tsdf: torch.Tensor
    gt: torch.Tensor
    max_l: torch.Tensor
    mid_p: torch.Tensor
    status: torch.Tensor
    gt_nor: Optional[torch.Tensor]


def plan_batches(n: int, batch_size: int, rank: int = 0, world: int = 1, shuffle: bool = False, seed: int = 0,
                 epoch: int = 0, drop_last: bool = False, weights: Optional[np.ndarray] = None,
                 balance: Optional[str] = None) -> List[np.ndarray]:
    """The frame indices of every batch of one epoch for one rank.  Ranks own CONTIGUOUS shards of the frame range
    (``shard.shard_bounds``) — the split BASELINE.json configs[3] names; shuffling permutes inside the rank's shard.
    No collective is involved.

    balance  ``"frames"``: shards of equal frame count (sizes differ by at most one) and THE SAME BATCHES — number AND
             sizes — ON EVERY RANK: what a training loop that steps a gradient collective once per batch needs.  A rank
             whose shard is one frame short repeats one of its frames (the first of its epoch order) at the end of its
             index list BEFORE the batches are cut, as ``torch.utils.data.DistributedSampler`` pads samples — so no rank
             ever sees a one-frame batch (BatchNorm in train mode, full weight in a per-batch all-reduce); with
             ``drop_last`` every rank keeps ``min(shard) // batch_size`` full batches.  ``n < world`` is a ValueError
             (a rank without a frame has nothing to repeat, and a per-batch collective would hang).
             ``"pixels"``: cut points balance ``weights`` (pixels per frame; MSRA boxes vary ~3x in area): equal WORK per
             rank, for voxelization / export jobs that never synchronise per batch.  Frame and batch counts then differ
             between ranks (8 ranks over MSRA-like subjects: 7 to 15 batches of 1024).
             Default: ``"pixels"`` when ``weights`` is given, else ``"frames"``.
    """
    if balance is None:
        balance = "pixels" if weights is not None else "frames"
    if balance not in ("frames", "pixels"):
        raise ValueError("balance must be 'frames' or 'pixels'")
    if balance == "pixels" and weights is None:
        raise ValueError("balance='pixels' needs the per-frame weights")
    bounds = shard.shard_bounds(n, world, weights if balance == "pixels" else None)
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    a, b = bounds[rank]
    idx = np.arange(a, b, dtype=np.int64)
    if shuffle:
        np.random.default_rng((seed, epoch, rank)).shuffle(idx)
    if balance == "frames":
        if n < world:
            raise ValueError(f"balance='frames' needs at least one frame per rank (n={n}, world={world})")
        lens = [e - s for s, e in bounds]
        if not drop_last and idx.size < max(lens):
            idx = np.concatenate([idx, idx[:1]])      # pad at sample level: shard sizes differ by at most one
    batches = [idx[i:i + batch_size] for i in range(0, idx.size, batch_size)]
    if balance == "frames":
        if drop_last:
            batches = batches[: min(lens) // batch_size]
    elif drop_last and batches and batches[-1].size < batch_size:
        batches.pop()
    return batches


class _Staging:
    """One reusable set of pinned host buffers + device input tensors for a batch (two of them alternate)."""

    def __init__(self, max_px: int, bs: int, device, with_gt: bool):
        self.h_depth = torch.empty(max_px, dtype=torch.float32).pin_memory()
        self.h_off = torch.empty(bs + 1, dtype=torch.int64).pin_memory()
        self.h_hdr = torch.empty((bs, 6), dtype=torch.int32).pin_memory()
        self.h_gt = torch.empty((bs, 63), dtype=torch.float32).pin_memory() if with_gt else None
        self.d_depth = torch.empty(max_px, dtype=torch.float32, device=device)
        # offsets / headers / labels stay in the pinned buffers: the voxelizer reads them over the link (include/tsdf.h).
        # ONE copy per batch: a small copy between two big ones can halve the next big one's rate (the runtime's
        # choice of copy engine; tools/exp_loader_meta.py: 630 k vs 880 k crops/s with the same bytes).
        self.copied = torch.cuda.Event()
        self.consumed = torch.cuda.Event()
        self.filled = threading.Event()   # host side: the worker has packed a batch into the pinned buffers
        self.free = threading.Event()     # host side: the H2D copy out of the pinned buffers has been ISSUED (the
        self.free.set()                   # worker then waits for `copied` itself, off the consumer's path)
        self.used = False
        self.n = 0
        self.npx = 0
        self.src = None                   # pinned tensor to upload from (the staging buffer, or a slice of a pinned pack)


class VoxelLoader:
    """Batches of voxel grids produced on the fly.

    A worker thread packs the next batch straight into one of TWO reusable pinned staging sets (no allocation
    and no ``pin_memory()`` per batch) while the GPU voxelizes and trains on the current one; uploads run on a
    side stream, the compute stream waits on an event, so H2D copies overlap the previous batch's kernels
    (BASELINE.json configs[2]).  Frames shard across ranks contiguously (:func:`plan_batches`).
    ``max_pixels`` bounds a batch's pixel count (default: ``batch_size`` full 320x240 frames).
    """

    def __init__(self, dataset: MSRADepthDataset, batch_size: int, device, res: int = 32,
                 shuffle: bool = False, seed: int = 0, drop_last: bool = False,
                 rank: int = 0, world: int = 1, labels: bool = True, clamp: bool = True,
                 max_pixels: Optional[int] = None, layout: str = "czyx", pin_packs: bool = True,
                 balance: str = "frames"):
        """``balance``: how ranks split the frames (:func:`plan_batches`): ``"frames"`` (default) gives every rank the same
        number of batches — required when the consumer synchronises per batch (DDP); ``"pixels"`` equalises the
        voxelization work instead (export jobs)."""
        self.balance = balance
        self.ds, self.bs, self.device, self.res = dataset, int(batch_size), torch.device(device), res
        self.shuffle, self.seed, self.drop_last = shuffle, seed, drop_last
        self.rank, self.world = rank, world
        self.labels, self.clamp, self.layout = labels, clamp, layout
        self.max_px = int(max_pixels) if max_pixels else self.bs * 320 * 240
        self.epoch = 0
        self.pin_packs = pin_packs
        self._sets: Optional[List[_Staging]] = None
        self._copy_stream = None

    def _batches(self) -> List[np.ndarray]:
        return plan_batches(len(self.ds), self.bs, self.rank, self.world, self.shuffle, self.seed, self.epoch,
                            self.drop_last, self.ds.pixels() if self.balance == "pixels" else None, self.balance)

    def __len__(self) -> int:
        return len(self._batches())

    def __iter__(self) -> Iterator[VoxelBatch]:
        batches = self._batches()
        self.epoch += 1
        if self._sets is None:
            self._sets = [_Staging(self.max_px, self.bs, self.device, True) for _ in range(2)]
            if self.pin_packs:
                self.ds.pin_packs()
        sets = self._sets
        for s in sets:
            if s.used:
                s.consumed.synchronize()   # an abandoned previous epoch may still be reading this set
            s.free.set()
            s.filled.clear()
            s.used = False
        err: "queue.Queue" = queue.Queue()
        stop = threading.Event()

        def work():
            try:
                for k, b in enumerate(batches):
                    s = sets[k & 1]
                    while not s.free.wait(0.05):
                        if stop.is_set():
                            return
                    s.free.clear()
                    if s.used:
                        s.consumed.synchronize()   # the kernel that read this set's pinned metadata (and therefore
                                                   # the upload out of its pinned depth buffer) is done
                    direct = self.ds.contiguous_source(b)
                    if direct is not None:            # DMA straight out of the pinned pack: no staging copy
                        s.src, pk = direct
                    else:
                        pk = self.ds.take(b, s.h_depth.numpy())
                        s.src = s.h_depth[: pk.depth.size]
                    if pk.depth.size > self.max_px:
                        raise ValueError(f"batch of {pk.depth.size} pixels exceeds max_pixels={self.max_px}")
                    s.n, s.npx = len(pk), int(pk.depth.size)
                    s.h_off.numpy()[: s.n + 1] = pk.offsets
                    s.h_hdr.numpy()[: s.n] = pk.headers
                    s.h_gt.numpy()[: s.n] = pk.gt
                    s.filled.set()
            except BaseException as e:  # surface I/O errors in the consumer
                err.put(e)
                for s in sets:
                    s.filled.set()

        t = threading.Thread(target=work, daemon=True)
        t.start()
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        copy_stream = self._copy_stream
        try:
            for k in range(len(batches)):
                # the stream the consumer is on NOW: a generator may be resumed under another torch.cuda.stream(...)
                # context than the one it was started in, and the launch below goes to whatever is current
                cur = torch.cuda.current_stream(self.device)
                s = sets[k & 1]
                s.filled.wait()
                if not err.empty():
                    raise err.get()
                s.filled.clear()
                n, npx = s.n, s.npx
                with torch.cuda.stream(copy_stream):
                    if k >= 2:
                        copy_stream.wait_event(s.consumed)   # the kernels that read this device set are done
                    s.d_depth[:npx].copy_(s.src, non_blocking=True)
                    s.copied.record(copy_stream)
                cur.wait_event(s.copied)
                # one launch: volumes, max_l / mid_p, normalised labels and the labels' device copy (which outlives the set)
                out, gt_nor, gt = voxelize_labels(s.d_depth[:npx], s.h_off[: n + 1], s.h_hdr[:n], s.h_gt[:n], res=self.res,
                                                  layout=self.layout, clamp=self.clamp, gt_copy=True)
                if not self.labels:
                    gt_nor = None
                s.consumed.record(cur)
                s.used = True
                s.free.set()   # issued: the worker may reuse the pinned buffers once `consumed` has fired
                yield VoxelBatch(out.tsdf, gt, out.max_l, out.mid_p, out.status, gt_nor)
        finally:
            stop.set()
            t.join()


class ResidentPacks:
    """The packs of a pack-backed dataset, uploaded to the GPU once: ``depth``, ``offsets[G+1]``, ``headers[G,6]``,
    ``gt[G,63]`` over the G frames of all packs back to back, and ``frame[d]`` = which of them dataset item d is."""

    def __init__(self, dataset: MSRADepthDataset, device):
        if not dataset.packed:
            raise ValueError("a resident dataset needs packs (packing.pack_tree + packed_dir=, or from_packs)")
        packs = dataset.packs
        px = np.array([int(pk.depth.size) for pk in packs], np.int64)
        nf = np.array([len(pk) for pk in packs], np.int64)
        pbase = np.concatenate([[0], np.cumsum(px)])
        fbase = np.concatenate([[0], np.cumsum(nf)])
        self.depth = torch.empty(int(pbase[-1]), dtype=torch.float32, device=device)
        import warnings
        with warnings.catch_warnings():      # (a memory-mapped pack is read-only: torch warns, and we only read)
            warnings.simplefilter("ignore", UserWarning)
            for k, pk in enumerate(packs):   # (a memory-mapped pack is read here, once)
                self.depth[int(pbase[k]):int(pbase[k + 1])].copy_(torch.from_numpy(np.ascontiguousarray(pk.depth)))
        off = np.concatenate([np.asarray(pk.offsets[:-1], np.int64) + pbase[k] for k, pk in enumerate(packs)]
                             + [pbase[-1:]])
        hdr = np.concatenate([np.asarray(pk.headers, np.int32).reshape(-1, 6) for pk in packs])
        gt = np.concatenate([np.asarray(pk.gt, np.float32).reshape(len(pk), -1) if pk.gt is not None
                             else np.zeros((len(pk), 63), np.float32) for pk in packs])
        self.offsets = torch.from_numpy(off).to(device)
        self.headers = torch.from_numpy(np.ascontiguousarray(hdr)).to(device)
        self.gt = torch.from_numpy(np.ascontiguousarray(gt)).to(device)
        self.frame = fbase[dataset._pack_of] + dataset._local

    def nbytes(self) -> int:
        return 4 * self.depth.numel()


class ResidentLoader:
    """Batches of voxel grids from a dataset that LIVES ON THE GPU.

    All of MSRA is 76.5 k crops = 4.8 GB; an MI355X has 288 GB.  So instead of feeding crops over the link batch after
    batch (:class:`VoxelLoader`: PCIe-bound, and for shuffled batches bound by the host's gather) the packs are uploaded
    ONCE and every batch is drawn by index on the device (``tsdf_voxelize_indexed_hip``): a training step sends n frame
    indices, nothing else.  Same arguments, same batches (:func:`plan_batches`) and same :class:`VoxelBatch` as
    ``VoxelLoader``; the dataset must be pack-backed (``MSRADepthDataset(packed_dir=...)`` / ``from_packs``).

    ``prefetch=k`` (k > 1) takes the host out of the step for SMALL batches — the reference trains with batch 16
    (3D_CNN/train.py:36), where one launch per batch is bound by Python, not by the GPU: the epoch's permutation is
    uploaded once, ONE launch voxelizes k consecutive batches (k*batch_size frames: the fused kernel's regime) into a
    ring buffer, and the loader yields k prebuilt ``batch_size``-frame views of it.  The batches are the same frames in
    the same order with the same values as with ``prefetch=1``; what changes is their lifetime: a batch's tensors are
    views of a ring of ``ring`` blocks and are overwritten once ``(ring - 1) * k`` further batches have been drawn (on the
    stream that was current when the block was launched — a consumer on that stream can never observe the overwrite;
    clone what you keep).  ``augment=True`` draws its maps per batch exactly as without prefetch.  The ``augment`` draws
    use numpy's ``Generator`` seeded per (seed, epoch, rank, batch): the reference's distributions, not its legacy stream.
    """

    def __init__(self, dataset: MSRADepthDataset, batch_size: int, device, res: int = 32, shuffle: bool = False,
                 seed: int = 0, drop_last: bool = False, rank: int = 0, world: int = 1, labels: bool = True,
                 clamp: bool = True, layout: str = "czyx", augment: bool = False, balance: str = "frames",
                 prefetch: int = 1, ring: int = 2):
        """``augment=True``: every frame of every batch gets a fresh 3-D augmentation with the reference's distributions
        (``augment.random_affines``, pre/process.py:209-216) about its own un-augmented grid centre, fused into the
        voxelizer (BASELINE configs[4]); the yielded ``gt`` are then the mapped joints, ``gt_nor`` their labels.
        ``balance``: see :func:`plan_batches` (``"frames"``: every rank yields the same number of batches)."""
        if not dataset.packed:
            raise ValueError("ResidentLoader needs a pack-backed dataset (packing.pack_tree + packed_dir=, or from_packs)")
        if prefetch < 1 or ring < 2:
            raise ValueError("prefetch must be >= 1 and ring >= 2")
        self.augment = bool(augment)
        self.balance = balance
        self.prefetch, self.ring = int(prefetch), int(ring)
        self.ds, self.bs, self.device, self.res = dataset, int(batch_size), torch.device(device), res
        self.shuffle, self.seed, self.drop_last = shuffle, seed, drop_last
        self.rank, self.world = rank, world
        self.labels, self.clamp, self.layout = labels, clamp, layout
        self.epoch = 0
        self._dev = None   # (depth, offsets, headers, gt) of all packs, on the device
        self._g = None     # dataset frame -> frame of the concatenated packs
        self._idx = None   # two pinned index buffers + the events of the launches that read them
        self._blocks = None  # prefetch > 1: the ring of output blocks and their prebuilt batch views

    def resident_bytes(self) -> int:
        return sum(4 * int(pk.depth.size) for pk in self.ds.packs)

    def _upload(self):
        rp = ResidentPacks(self.ds, self.device)
        self._dev = (rp.depth, rp.offsets, rp.headers, rp.gt)
        self._g = rp.frame
        from . import _lib
        # (batches of up to 32 frames hand their index over by value — tsdf_voxelize_indexed_host_hip —, which an ordinary
        # CPU tensor selects in voxelize_indexed; larger ones, and augmented ones, are read from page-locked memory)
        pin = self.bs > _lib.INLINE_INDEX_MAX or self.augment
        self._idx = [(torch.empty(self.bs, dtype=torch.int64).pin_memory() if pin else torch.empty(self.bs, dtype=torch.int64),
                      torch.cuda.Event()) for _ in range(2)]
        self._used = [False, False]
        self._mid = None
        if self.augment:   # the centres the maps turn about: every frame's own grid centre, one AABB launch over the pack
            from .voxelize import aabb
            d, o, h, _ = self._dev
            self._mid = aabb(d, o, h, res=self.res).grid[:, :3].cpu().numpy().astype(np.float64)
            self._xf = [torch.empty((self.bs, 24), dtype=torch.float64).pin_memory() for _ in range(2)]

    def _batches(self) -> List[np.ndarray]:
        return plan_batches(len(self.ds), self.bs, self.rank, self.world, self.shuffle, self.seed, self.epoch,
                            self.drop_last, self.ds.pixels() if self.balance == "pixels" else None, self.balance)

    def __len__(self) -> int:
        return len(self._batches())

    def __iter__(self) -> Iterator[VoxelBatch]:
        batches = self._batches()
        self.epoch += 1
        if self._dev is None:
            self._upload()
        if self.prefetch > 1:
            return self._iter_blocks(batches, self.epoch)
        return self._iter_single(batches, self.epoch)

    def _iter_single(self, batches, epoch) -> Iterator[VoxelBatch]:
        depth, off, hdr, gt = self._dev
        for k, b in enumerate(batches):
            cur = torch.cuda.current_stream(self.device)   # (per batch: the consumer may have switched streams)
            h_idx, done = self._idx[k & 1]
            if self._used[k & 1]:
                done.synchronize()          # the launch that read this index buffer two batches ago
            n = int(b.size)
            gidx = self._g[b]
            h_idx.numpy()[:n] = gidx
            xf = None
            if self.augment:
                from . import augment as _aug
                h_xf = self._xf[k & 1]
                h_xf.numpy()[:n] = _aug.random_affines(self._mid[gidx], rng=(self.seed, epoch, self.rank, k))[0]
                xf = h_xf[:n].to(self.device, non_blocking=True)   # (read per voxel: device memory, not the link)
            out, gt_nor, g = voxelize_indexed(depth, off, hdr, h_idx[:n], gt, res=self.res, layout=self.layout,
                                              clamp=self.clamp, gt_copy=True, xforms=xf)
            done.record(cur)
            self._used[k & 1] = True
            yield VoxelBatch(out.tsdf, g, out.max_l, out.mid_p, out.status, gt_nor if self.labels else None)

    # ---- prefetch > 1: one launch per block of `prefetch` batches ----
    def _make_blocks(self):
        depth, off, hdr, gt = self._dev
        cap, R = self.bs * self.prefetch, self.res
        lab_shape = (cap,) + tuple(gt.shape[1:])
        blocks = []
        for _ in range(self.ring):
            out = TsdfBatch(torch.empty((cap, 3, R, R, R), dtype=torch.float32, device=self.device),
                            torch.empty((cap,), dtype=torch.float32, device=self.device),
                            torch.empty((cap, 3), dtype=torch.float32, device=self.device),
                            torch.empty((cap,), dtype=torch.int32, device=self.device))
            gt_nor = torch.empty(lab_shape, dtype=torch.float32, device=self.device)
            g = torch.empty(lab_shape, dtype=torch.float32, device=self.device)
            views = [self._view(out, gt_nor, g, j * self.bs, (j + 1) * self.bs) for j in range(self.prefetch)]
            blocks.append((out, gt_nor, g, views))
        return blocks

    def _view(self, out, gt_nor, g, a, b) -> VoxelBatch:
        return VoxelBatch(out.tsdf[a:b], g[a:b], out.max_l[a:b], out.mid_p[a:b], out.status[a:b],
                          gt_nor[a:b] if self.labels else None)

    def _iter_blocks(self, batches, epoch) -> Iterator[VoxelBatch]:
        depth, off, hdr, gt = self._dev
        if self._blocks is None:
            self._blocks = self._make_blocks()
        if not batches:
            return
        bs, P = self.bs, self.prefetch
        # the epoch's permutation goes up once; a launch reads its slice of it
        flat = np.concatenate(batches)
        gidx = self._g[flat]
        d_idx = torch.from_numpy(np.ascontiguousarray(gidx)).to(self.device)
        d_xf = None
        if self.augment:   # the same draws as without prefetch: one generator per batch
            from . import augment as _aug
            xf = np.empty((flat.size, 24), np.float64)
            pos = 0
            for k, b in enumerate(batches):
                xf[pos:pos + b.size] = _aug.random_affines(self._mid[gidx[pos:pos + b.size]],
                                                           rng=(self.seed, epoch, self.rank, k))[0]
                pos += b.size
            d_xf = torch.from_numpy(xf).to(self.device)
        pos = 0
        for blk, k0 in enumerate(range(0, len(batches), P)):
            mine = batches[k0:k0 + P]
            nfr = int(sum(b.size for b in mine))
            out, gt_nor, g, views = self._blocks[blk % self.ring]
            full = nfr == bs * P
            if full:
                o, gn, gg = out, gt_nor, g
            else:      # the epoch's last block: the leading part of the ring entry
                o = TsdfBatch(out.tsdf[:nfr], out.max_l[:nfr], out.mid_p[:nfr], out.status[:nfr])
                gn, gg = gt_nor[:nfr], g[:nfr]
            voxelize_indexed(depth, off, hdr, d_idx[pos:pos + nfr], gt, res=self.res, layout=self.layout, clamp=self.clamp,
                             out=o, out_gt_nor=gn, out_gt=gg,
                             xforms=None if d_xf is None else d_xf[pos:pos + nfr])
            pos += nfr
            if full:
                yield from views
            else:
                a = 0
                for b in mine:
                    yield self._view(out, gt_nor, g, a, a + int(b.size))
                    a += int(b.size)


class PreBatched:
    """What :meth:`MSRA_Dataset.__getitems__` hands to torch's DataLoader: the batch, ALREADY batched.  torch's
    ``default_collate`` dispatches on the type of ``batch[0]``; this type is registered with it (below), so the list
    ``[PreBatched(batch)]`` collates to ``batch`` itself — no per-item views, no ``torch.stack``."""

    __slots__ = ("batch",)

    def __init__(self, batch):
        self.batch = batch


def _collate_prebatched(batch, *, collate_fn_map=None):
    return batch[0].batch


try:   # the documented extension point of torch.utils.data.default_collate
    from torch.utils.data._utils.collate import default_collate_fn_map as _collate_map
    _collate_map[PreBatched] = _collate_prebatched
except ImportError:   # pragma: no cover  (a torch without the map: MSRA_Dataset(prebatched=False) still works)
    _collate_map = None


class _SlotGuard:
    """Who else holds a ring slot's batch?  Two counts per tensor, compared with what they were when only the ring held
    the slot:

      * the Python reference count of the batch tuple and of each of its tensors — a consumer that keeps the tuple, one
        tensor object, or anything that keeps those objects alive (a dlpack capsule, a closure, a frame a debugger or
        ``sys.settrace`` hook holds on to);
      * the use count of each tensor's STORAGE (``torch._C._storage_Use_Count``) — every other tensor object over the same
        memory: a view (``t[:, 1]``), ``t.detach()``, ``t.data``, ``t.numpy()``, a tensor autograd saved for backward.
        (Round 4 relied on a live view bumping its base tensor's Python reference count — a torch-internal side effect —
        and did not see ``detach()`` / ``.data`` aliases at all: ADVICE round 4.)

    Equality with the baseline means "only the ring"; ANY difference means "held" and the slot gets fresh tensors, so an
    unexpected extra reference (a profiler, a different Python) costs an allocation, never a wrong batch.  What no count can
    see is a consumer that dropped every reference but still has work queued on ANOTHER stream: order that stream after
    the loader's (the usual rule for GPU tensors).  :func:`_slot_guard_works` checks the counting rules themselves once
    per process; a torch on which they do not hold makes the ring hand out fresh tensors for every batch."""

    __slots__ = ("batch", "_stor", "_cdata", "_base")

    def __init__(self, batch):
        self.batch = batch
        self._stor = tuple(t.untyped_storage() for t in batch)   # (kept: a wrapper's _cdata is only valid while it lives)
        self._cdata = tuple(s._cdata for s in self._stor)
        self._base = None

    def counts(self):
        rc, uc, b, c = sys.getrefcount, _storage_use_count, self.batch, self._cdata
        return (rc(b), rc(b[0]), rc(b[1]), rc(b[2]), rc(b[3]), uc(c[0]), uc(c[1]), uc(c[2]), uc(c[3]))

    def arm(self):
        """Call when every reference the ring itself keeps is in place and no local of the caller refers to the batch."""
        self._base = self.counts()

    def held(self) -> bool:
        return self.counts() != self._base


_storage_use_count = getattr(torch._C, "_storage_Use_Count", None)
_guard_ok: Optional[bool] = None


def _slot_guard_works() -> bool:
    """The counting rules _SlotGuard relies on, tried once on small CPU tensors: a kept tuple, a kept tensor, a view, a
    ``detach()`` alias, a ``.data`` alias and a numpy alias must each read as "held", and dropping them as "free"."""
    global _guard_ok
    if _guard_ok is None:
        ok = _storage_use_count is not None
        if ok:
            try:
                holder = [PreBatched(tuple(torch.zeros(2, 3) for _ in range(4)))]
                g = _SlotGuard(holder[0].batch)
                g.arm()
                ok = not g.held()
                for make in (lambda b: b, lambda b: b[1], lambda b: b[0][:, 1], lambda b: b[2].detach(), lambda b: b[3].data,
                             lambda b: b[0].numpy()):
                    ref = make(holder[0].batch)
                    ok = ok and g.held()
                    del ref
                    ok = ok and not g.held()
            except Exception:   # pragma: no cover  (an API that moved)
                ok = False
        _guard_ok = bool(ok)
    return _guard_ok


class MSRA_Dataset(data.Dataset):
    """The reference's dataset class with its constructor and item tuple (3D_CNN/dataset.py:16-79):

        MSRA_Dataset(root_path, opt, train=True, aug=False)[i] -> (tsdf[3,32,32,32], gt[63], max_l, mid_p)

    but ``root_path`` is the RAW MSRA tree (or a directory of packs, see ``packed_dir``), nothing is preprocessed
    and nothing but the depth crops is held in host memory: items are voxelized on the GPU in blocks of
    ``block`` consecutive frames the first time one of them is asked for, and returned as GPU tensors (the
    reference returns numpy rows that its training loop then moves with ``.cuda()``, train.py:200,232).
    ``opt.size`` / ``opt.test_index`` are honoured when present (the reference ignores ``opt`` and hard-codes
    ``'small'`` / 2, :20-22).

    ``aug=True`` (3D_CNN/dataset.py:57-62: "add augmentation dataset") appends an AUGMENTED rendition of every frame the
    dataset holds: items ``[n, 2n)`` are frames ``[0, n)`` under a fixed per-frame 3-D augmentation, drawn once with the
    reference's distributions (``augment.draw_params(n, aug_seed)``) about the frame's own un-augmented grid centre and
    FUSED into the voxelizer (the re-specified contract of ``tsdf_voxelize_aug_hip``; the reference's own ``_aug`` files
    cannot be produced — ``data_aug`` raises AxisError — and its loader reads the plain files again for them, SURVEY.md
    App. B#8,#11).  The item's ``gt`` are the joints under the same map, ``max_l`` / ``mid_p`` those of the augmented
    grid.  With ``aug=True`` every batch goes through the augmented entry — plain items with the identity map, whose
    volumes equal the plain entry's to the float32 rounding (grid, zero mask, sign and z component bit for bit).

    Under the reference's own ``DataLoader(dataset, batch_size=B, shuffle=True)`` (train.py:36,86-91; ``num_workers=0``:
    the items are GPU tensors) the loader hands the batch's indices to :meth:`__getitems__`, which voxelizes exactly
    those B frames in ONE launch and returns them ALREADY BATCHED (``prebatched=True``, the default on a resident
    dataset): the indices go into a page-locked ring the kernel reads over the link, the outputs into a ring of
    preallocated batches, and the result is a :class:`PreBatched` that torch's ``default_collate`` unwraps — the host
    side of a step is one index conversion and one C call.  What the ``DataLoader`` yields is the reference's collated
    batch ``(tsdf[B,3,32,32,32], gt[B,63], max_l[B], mid_p[B,3])``, on the GPU.  The ring (``ring`` slots; default: as
    many batches as fit 2 GiB of volumes, between 2 and 256) recycles a slot's tensors only when the consumer holds no
    reference to that batch any more — a batch that is kept (``list(dl)``, collected outputs, a view of one tensor) keeps
    its tensors and the slot gets new ones, so the loader behaves like the reference's, which returns independent
    tensors.  With a custom ``collate_fn`` pass ``prebatched=False``: a list of
    item tuples, views into the batch's tensors, as torch documents for ``__getitems__``.
    A lone ``dataset[i]`` voxelizes frame i alone unless the access pattern is a sequential walk, which is served
    from a block.  For the highest throughput at small batch sizes use :class:`ResidentLoader` with ``prefetch``: it
    knows the epoch's permutation in advance, which nothing behind ``DataLoader`` can.
    """

    def __init__(self, root_path, opt=None, train=True, aug=False, device="cuda", block: int = 1024,
                 packed_dir: Optional[str] = None, resident: Optional[bool] = None, prebatched: Optional[bool] = None,
                 ring: Optional[int] = None, aug_seed: int = 0, _raw: Optional[MSRADepthDataset] = None):
        self.AUG = bool(aug)
        self.aug_seed = aug_seed
        self.size = getattr(opt, "size", "small")
        self.test_idx = int(getattr(opt, "test_index", 2))
        self.PCA_SZ = int(getattr(opt, "PCA_SZ", 63))
        self.train = train
        self.raw = _raw if _raw is not None else MSRADepthDataset(root_path, train=train, test_idx=self.test_idx,
                                                                  size=self.size, packed_dir=packed_dir)
        self.device = torch.device(device)
        self.block = int(block)
        self._cache_block = -1
        self._cache: Optional[TsdfBatch] = None
        self._cache_gt: Optional[torch.Tensor] = None
        self._last = -1          # the last index served (a sequential walk is answered from blocks)
        # resident (default: whenever the dataset is pack-backed): the packs go to the GPU once and a batch is a list of
        # indices resolved on the device (tsdf_voxelize_indexed_hip) — no crop crosses the link after start-up
        self.resident = self.raw.packed if resident is None else bool(resident)
        self.prebatched = (self.resident and _collate_map is not None) if prebatched is None else bool(prebatched)
        if self.prebatched and not self.resident:
            raise ValueError("prebatched=True needs a resident (pack-backed) dataset")
        self._ring_req = ring
        self._rp: Optional[ResidentPacks] = None
        self._fast = None        # the ring of the pre-batched path (built on the first batch)
        self._n = len(self.raw)
        self._aug_params = None  # aug=True: (stretch, rot_xy, rot_z) of every frame, drawn once
        self._xf_table = None    # ... and, resident, the maps of all 2n items (identity for the first n), float64[2n,24]
        if self.AUG:
            from . import augment as _aug
            self._aug_params = _aug.draw_params(self._n, aug_seed)

    @classmethod
    def from_raw(cls, raw: MSRADepthDataset, device="cuda", **kw) -> "MSRA_Dataset":
        """The on-the-fly dataset over raw frames that are already open (``MSRADepthDataset`` / ``from_packs``)."""
        return cls(None, device=device, _raw=raw, **kw)

    def __len__(self):
        return 2 * self._n if self.AUG else self._n

    def _resident_packs(self) -> "ResidentPacks":
        if self._rp is None:
            self._rp = ResidentPacks(self.raw, self.device)
            if self.AUG:   # every frame's own grid centre (one AABB launch over the pack), then all maps at once
                from . import augment as _aug
                from .voxelize import aabb
                rp = self._rp
                mid = aabb(rp.depth, rp.offsets, rp.headers).grid[:, :3].cpu().numpy().astype(np.float64)[rp.frame]
                self._xf_table = np.ascontiguousarray(np.concatenate(
                    [_aug.identity_affines(self._n), _aug.affines_from_params(mid, *self._aug_params)]))
                self._frame2 = np.ascontiguousarray(np.concatenate([rp.frame, rp.frame]))
        return self._rp

    def _aug_batch_host_fed(self, idx: np.ndarray):
        """aug=True on a dataset that is not resident: upload the frames, one AABB launch for their centres, then the
        augmented entry (identity maps for the plain items)."""
        from . import augment as _aug
        from .voxelize import aabb, voxelize_aug
        src = idx % self._n
        pk = self.raw.take(src)
        depth, offsets, headers = pk.to_torch(self.device, pin=False, non_blocking=False)
        mid = aabb(depth, offsets, headers).grid[:, :3].cpu().numpy().astype(np.float64)
        st, rxy, rz = (p[src] for p in self._aug_params)
        xf = _aug.affines_from_params(mid, st, rxy, rz)
        plain = idx < self._n
        xf[plain] = _aug.identity_affines(int(plain.sum()))
        gt = torch.from_numpy(np.ascontiguousarray(pk.gt)).to(self.device)
        out, _, gt_aug = voxelize_aug(depth, offsets, headers, torch.from_numpy(xf).to(self.device), res=32, gt=gt)
        return out, gt_aug

    def _load_block(self, blk: int):
        a, b = blk * self.block, min(len(self.raw), (blk + 1) * self.block)
        if self.resident:
            rp = self._resident_packs()
            self._cache, _, self._cache_gt = voxelize_indexed(
                rp.depth, rp.offsets, rp.headers, torch.from_numpy(rp.frame[a:b]).to(self.device), rp.gt, gt_copy=True)
        else:
            pk = self.raw.take(np.arange(a, b))
            depth, offsets, headers = pk.to_torch(self.device, pin=False, non_blocking=False)
            self._cache = voxelize(depth, offsets, headers, res=32)
            self._cache_gt = torch.from_numpy(np.ascontiguousarray(pk.gt)).to(self.device)
        self._cache_block = blk

    # ---- the pre-batched path: one C call per batch, nothing allocated, nothing sliced ----
    class _Fast:
        """Ring state for batches of (at most) ``bs`` frames: output slots, one prebuilt result per slot and the C call's
        constant arguments.  Batches of up to 32 plain frames hand their index to the GPU by value (``by_value``:
        ``tsdf_voxelize_indexed_host_hip``); larger ones, and ``aug=True`` datasets, write it into page-locked index slots
        the kernel reads over the link, with one event per ``kGroup`` index slots that tells when a group's words have been
        read (an index slot's words are rewritten ``iring`` batches later).

        A slot is RECYCLED only when nothing outside this object refers to its batch any more (:class:`_SlotGuard`: Python
        reference counts of the tuple and its tensors, storage use counts for views and ``detach()`` / ``.data`` / numpy
        aliases).  Every slot owns its tensors (separate allocations); a slot whose batch is still held is given fresh
        tensors instead (``replaced`` counts them) and the kept batch stays what it was — like the independent tensors the
        reference's loader returns (3D_CNN/train.py:86-91 over numpy rows).  ``list(DataLoader(...))``, an evaluation
        loop that collects outputs, a loss history: all safe.  What cannot be seen is a consumer that dropped every
        reference but still has work queued on ANOTHER stream: order that stream after the loader's (the usual rule for
        GPU tensors).  On a torch where the counting rules do not hold (``_slot_guard_works``) every batch gets fresh
        tensors: slower, never wrong."""

        kGroup = 16

        def __init__(self, rp: "ResidentPacks", bs: int, ring: int, device, frame=None, xf_table=None):
            import ctypes
            from . import _lib
            self._ctypes, self._lib = ctypes, _lib
            self.always_fresh = not _slot_guard_works()
            ring = max(2, int(ring))
            self.iring = -(-max(ring, self.kGroup) // self.kGroup) * self.kGroup   # index slots: whole event groups
            self.bs, self.ring, self.count, self.replaced = bs, ring, 0, 0
            self.L = _lib.load()
            self.rp_gt = rp.gt
            self.nc = rp.gt.shape[1]
            d = torch.device(device)
            self.device = d if d.index is not None else torch.device("cuda", torch.cuda.current_device())
            self.dev_index = self.device.index
            self.h_idx = torch.empty((self.iring, bs), dtype=torch.int64).pin_memory()
            self.h_idx_np = self.h_idx.numpy()
            self.rows = [self.h_idx_np[k] for k in range(self.iring)]
            self.idx_ptrs = [self.h_idx[k].data_ptr() for k in range(self.iring)]
            self.head = (rp.depth.data_ptr(), rp.depth.numel(), rp.offsets.data_ptr(), rp.headers.data_ptr(),
                         int(rp.headers.shape[0]))
            self.slots = [None] * ring      # (tsdf, gt, max_l, mid_p, status, gt_nor) — the slot's own tensors
            self.labels = [None] * ring
            self.args = [None] * ring       # (tsdf, max_l, mid_p, status, byref(labels)) pointers of the C call
            self.results = [None] * ring    # [PreBatched((tsdf, gt, max_l, mid_p))]
            self.guard = [None] * ring      # _SlotGuard of the slot's batch
            for k in range(ring):
                self._fresh(k)
            self.read = [torch.cuda.Event() for _ in range(self.iring // self.kGroup)]
            self.read_used = [False] * (self.iring // self.kGroup)
            from .voxelize import _get_raw_stream
            self.raw_stream = _get_raw_stream if _get_raw_stream is not None else \
                (lambda i: torch.cuda.current_stream(i).cuda_stream)
            self.cur_dev = getattr(torch._C, "_cuda_getDevice", torch.cuda.current_device)
            self.fn = self.L.tsdf_voxelize_indexed_hip
            self.take = (rp.frame if frame is None else frame).take   # dataset item -> frame of the resident packs
            # batches of at most INLINE_INDEX_MAX frames: the index goes to the GPU inside the kernel arguments
            # (tsdf_voxelize_indexed_host_hip reads it during the call) — no page-locked slot, no event, and the launch
            # does not start with a read over the link
            self.by_value = xf_table is None and bs <= _lib.INLINE_INDEX_MAX
            self.fn_host = self.L.tsdf_voxelize_indexed_host_hip
            self.idx_buf = np.empty(bs, np.int64)
            self.idx_ptr = self.idx_buf.ctypes.data
            self.xf_take = None
            if xf_table is not None:   # aug=True: one map per batch position, in page-locked memory the kernel reads
                self.fn_aug = self.L.tsdf_voxelize_indexed_aug_hip
                self.h_xf = torch.empty((self.iring, bs, 24), dtype=torch.float64).pin_memory()
                self.h_xf_np = self.h_xf.numpy()
                self.xf_rows = [self.h_xf_np[k] for k in range(self.iring)]
                self.xf_ptr = [self.h_xf[k].data_ptr() for k in range(self.iring)]
                self.xf_take = xf_table.take

        def _fresh(self, k: int) -> None:
            """New tensors for slot k (at start-up, and whenever its previous batch is still held by the consumer)."""
            self._alloc(k)
            self.guard[k].arm()      # (taken when _alloc's locals are gone: the ring's own references)

        def _alloc(self, k: int) -> None:
            R, bs, nc, dev = 32, self.bs, self.nc, self.device
            tsdf = torch.empty((bs, 3, R, R, R), dtype=torch.float32, device=dev)
            gt = torch.empty((bs, nc), dtype=torch.float32, device=dev)
            max_l = torch.empty(bs, dtype=torch.float32, device=dev)
            mid_p = torch.empty((bs, 3), dtype=torch.float32, device=dev)
            status = torch.empty(bs, dtype=torch.int32, device=dev)
            gt_nor = torch.empty((bs, nc), dtype=torch.float32, device=dev)
            self.slots[k] = (tsdf, gt, max_l, mid_p, status, gt_nor)
            self.labels[k] = self._lib.TsdfLabels(self.rp_gt.data_ptr(), nc // 3, 1, gt_nor.data_ptr(), gt.data_ptr())
            self.args[k] = (tsdf.data_ptr(), max_l.data_ptr(), mid_p.data_ptr(), status.data_ptr(),
                            self._ctypes.byref(self.labels[k]))
            self.results[k] = [PreBatched((tsdf, gt, max_l, mid_p))]
            self.guard[k] = _SlotGuard(self.results[k][0].batch)

        def held(self, k: int) -> bool:
            """Does anything outside the ring still refer to slot k's batch (the tuple, a tensor, any alias of one)?"""
            return self.always_fresh or self.guard[k].held()

        def next_slot(self) -> int:
            """The output slot of the next batch, free to be overwritten."""
            k = self.count % self.ring
            if self.held(k):
                self._fresh(k)
                self.replaced += 1
            return k

        def sync(self) -> None:
            """Everything queued through this ring has run (called before the ring is dropped: queued launches read its
            page-locked index words by raw pointer, which torch's pinned-memory allocator knows nothing about)."""
            torch.cuda.current_stream(self.device).synchronize()

    def _ring_size(self, n: int) -> int:
        """Output slots of the pre-batched ring: an explicit ``ring`` is honoured (at least 2); otherwise as many batches
        as fit 2 GiB of volumes, between 2 and 256 (batch 16: 256 slots = 1.6 GB; batch 1024: 5 slots = 2.0 GB)."""
        if self._ring_req:
            return max(2, int(self._ring_req))
        vol = n * (3 * 32 ** 3 * 4)
        return max(2, min(256, (2 << 30) // max(vol, 1)))

    def _fast_batch(self, indices):
        """One batch through the ring: ~4 us of Python around the C call (the HIP launch itself is the larger part)."""
        f = self._fast
        n = len(indices)
        if f is None or n > f.bs:
            if f is not None:
                f.sync()      # launches that read the old ring's page-locked words must be done before it goes away
            f = self._fast = MSRA_Dataset._Fast(self._rp, n, self._ring_size(n), self.device,
                                                frame=self._frame2 if self.AUG else None, xf_table=self._xf_table)
        k = f.next_slot()
        a = f.args[k]
        if f.by_value:                        # (the epoch's short last batch, or another current device)
            f.idx_buf[:n] = f.take(indices)
            with torch.cuda.device(f.device):
                rc = f.fn_host(*f.head, f.idx_ptr, n, 32, None, 0, f.raw_stream(f.dev_index), a[0], a[1], a[2], a[3], a[4])
            if rc != 0:
                from . import _lib
                _lib.check(rc, "tsdf_voxelize_indexed_host_hip")
            f.count += 1
            if n == f.bs:
                return f.results[k]
            t = f.slots[k]
            return [PreBatched((t[0][:n], t[1][:n], t[2][:n], t[3][:n]))]
        ki = f.count % f.iring                # the index words' slot (page-locked ring of whole event groups)
        within = ki & (f.kGroup - 1)
        if within == 0 and f.read_used[ki >> 4]:
            f.read[ki >> 4].synchronize()     # the launches that read this group's index words a ring ago are done
        # dataset item -> pack frame, written where the kernel will read it; numpy checks the range (IndexError) and,
        # like a Python list, counts negative indices from the end
        if n == f.bs:
            f.take(indices, out=f.rows[ki])
        else:
            f.h_idx_np[ki, :n] = f.take(indices)
        if f.xf_take is not None:     # aug=True: the batch's maps next to its indices
            if n == f.bs:
                f.xf_take(indices, axis=0, out=f.xf_rows[ki])
            else:
                f.h_xf_np[ki, :n] = f.xf_take(indices, axis=0)
            with torch.cuda.device(f.device):
                rc = f.fn_aug(*f.head, f.idx_ptrs[ki], n, 32, None, 0, f.raw_stream(f.dev_index), f.xf_ptr[ki], a[0], a[1],
                              a[2], a[3], a[4])
        elif f.cur_dev() == f.dev_index:
            rc = f.fn(*f.head, f.idx_ptrs[ki], n, 32, None, 0, f.raw_stream(f.dev_index), a[0], a[1], a[2], a[3], a[4])
        else:
            with torch.cuda.device(f.device):
                rc = f.fn(*f.head, f.idx_ptrs[ki], n, 32, None, 0, f.raw_stream(f.dev_index), a[0], a[1], a[2], a[3], a[4])
        if within == f.kGroup - 1:
            f.read[ki >> 4].record(torch.cuda.current_stream(f.device))
            f.read_used[ki >> 4] = True
        if rc != 0:
            from . import _lib
            _lib.check(rc, "tsdf_voxelize_indexed_hip")
        f.count += 1
        if n == f.bs:
            return f.results[k]
        t = f.slots[k]
        return [PreBatched((t[0][:n], t[1][:n], t[2][:n], t[3][:n]))]   # the epoch's short last batch

    def __getitems__(self, indices):
        """The frames of one batch, voxelized by one launch (torch's DataLoader calls this with the batch's indices
        when it exists).  ``prebatched``: ``[PreBatched(batch)]``, which ``default_collate`` turns into the batch; else a
        list of item tuples, views into the batch's tensors."""
        if data.get_worker_info() is not None:
            raise RuntimeError("MSRA_Dataset produces its items on the GPU: use it with num_workers=0 (the reference's "
                               "default, train.py:38), there is nothing for loader processes to do")
        f = self._fast
        if f is not None and f.by_value and len(indices) == f.bs and f.cur_dev() == f.dev_index:
            # the hot path of a training epoch, inlined (every microsecond here is 6 % of a batch of 16): a full batch of
            # plain items on the current device — item -> pack frame, one C call that takes the index by value, the
            # ring slot's prebuilt result
            k = f.next_slot()      # (a slot whose batch the consumer still holds gets fresh tensors)
            f.take(indices, out=f.idx_buf)
            a = f.args[k]
            rc = f.fn_host(*f.head, f.idx_ptr, f.bs, 32, None, 0, f.raw_stream(f.dev_index), a[0], a[1], a[2], a[3], a[4])
            if rc != 0:
                from . import _lib
                _lib.check(rc, "tsdf_voxelize_indexed_host_hip")
            f.count += 1
            self._last = indices[-1]
            return f.results[k]
        if self.resident:
            self._resident_packs()
        if self.prebatched:
            if not indices:
                raise IndexError("empty batch")
            self._last = indices[-1]
            return self._fast_batch(indices)
        idx = np.asarray([int(i) for i in indices], np.int64)
        if idx.size and (idx.min() < 0 or idx.max() >= len(self)):
            raise IndexError(int(idx.max() if idx.max() >= len(self) else idx.min()))
        self._last = int(idx[-1]) if idx.size else self._last
        if self.AUG:
            if self.resident:
                rp = self._rp
                xf = torch.from_numpy(self._xf_table[idx]).to(self.device)
                out, _, gt = voxelize_indexed(rp.depth, rp.offsets, rp.headers,
                                              torch.from_numpy(self._frame2[idx]).to(self.device), rp.gt, gt_copy=True,
                                              xforms=xf)
            else:
                out, gt = self._aug_batch_host_fed(idx)
            return [(out.tsdf[k], gt[k], out.max_l[k], out.mid_p[k]) for k in range(idx.size)]
        if self.resident:
            rp = self._rp
            out, _, gt = voxelize_indexed(rp.depth, rp.offsets, rp.headers,
                                          torch.from_numpy(rp.frame[idx]).to(self.device), rp.gt, gt_copy=True)
            return [(out.tsdf[k], gt[k], out.max_l[k], out.mid_p[k]) for k in range(idx.size)]
        pk = self.raw.take(idx)
        depth, offsets, headers = pk.to_torch(self.device, pin=False, non_blocking=False)
        out = voxelize(depth, offsets, headers, res=32)
        gt = torch.from_numpy(np.ascontiguousarray(pk.gt)).to(self.device)
        return [(out.tsdf[k], gt[k], out.max_l[k], out.mid_p[k]) for k in range(idx.size)]

    def _items_of(self, indices):
        """``indices`` as a list of item tuples whatever ``prebatched`` says (single-item access)."""
        r = self.__getitems__(indices)
        if r and isinstance(r[0], PreBatched):
            b = r[0].batch
            return [tuple(t[k].clone() for t in b) for k in range(len(indices))]   # (out of the ring: the item outlives it)
        return r

    def __getitem__(self, index):
        index = int(index)
        if not 0 <= index < len(self):
            raise IndexError(index)
        if self.AUG:                                              # (no block cache: plain and augmented items mix)
            self._last = index
            return self._items_of([index])[0]
        blk = index // self.block
        if blk != self._cache_block:
            if index != self._last + 1 and index % self.block:   # random access: this frame alone
                return self._items_of([index])[0]
            self._load_block(blk)                                 # a sequential walk: the whole block at once
        self._last = index
        k = index - blk * self.block
        c = self._cache
        return c.tsdf[k], self._cache_gt[k], c.max_l[k], c.mid_p[k]
