"""On-the-fly MSRA dataset: depth crops in, voxel grids out of the GPU — no preprocessed npz files.

The reference voxelizes every frame offline (pre/read_MSRA.py:37-140, ~0.2 s per frame), writes
``result/<subject>/TSDF/<gesture>.npz`` and then loads *all* of it into host RAM
(3D_CNN/dataset.py:35-38,99-117: ~76 k frames x 393 KB).  Here the dataset yields the raw frames
(``header``, ``depth`` crop, ``gt``) straight from the ``.bin`` files and a collate function packs a
batch, uploads it once and calls the HIP voxelizer on the training stream; what comes out is the
tuple the reference's ``__getitem__`` returns (3D_CNN/dataset.py:73-79), batched and already on
the GPU: ``(tsdf[n,3,R,R,R], gt[n,63], max_l[n], mid_p[n,3])``.

Kept from the reference: directory layout ``<root>/<subject>/<gesture>/{joint.txt, 000000_depth.bin..}``
(pre/read_MSRA.py:46-50,79,99), leave-one-subject-out split (3D_CNN/dataset.py:44-53) and the
``small`` subset of 4 subjects x 5 gestures (:26-31).  Fixed: ``opt`` is honoured instead of being
ignored (:20-22).
"""
from __future__ import annotations

import os
import queue
import threading
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.utils.data as data

from . import packing
from .voxelize import TsdfBatch, denormalize_joints, normalize_joints, voxelize  # noqa: F401


class MSRADepthDataset(data.Dataset):
    """Raw MSRA frames: ``__getitem__ -> (header int32[6], depth float32[N], gt float32[63])``."""

    def __init__(self, root_path: str, train: bool = True, test_idx: int = 2, size: str = "full",
                 subjects: Optional[Sequence[str]] = None):
        if size == "full":
            n_sub, n_ges = 9, 17
        elif size == "small":
            n_sub, n_ges = 4, 5
        else:
            raise ValueError("size must be 'full' or 'small'")
        self.root_path = root_path
        self.train = train
        self.test_idx = test_idx
        all_sub = sorted(d for d in os.listdir(root_path) if os.path.isdir(os.path.join(root_path, d)))
        all_sub = list(subjects) if subjects is not None else all_sub[:n_sub]
        if not 0 <= test_idx < len(all_sub):
            raise ValueError("test_idx out of range")
        chosen = [s for i, s in enumerate(all_sub) if (i != test_idx) == train]
        self.paths: List[str] = []
        gts: List[np.ndarray] = []
        for sub in chosen:
            sub_dir = os.path.join(root_path, sub)
            gestures = sorted(g for g in os.listdir(sub_dir) if os.path.isdir(os.path.join(sub_dir, g)))
            for ges in gestures[:n_ges]:
                g_dir = os.path.join(sub_dir, ges)
                bin_num, gt = packing.read_joint(g_dir)
                self.paths += packing.gesture_bin_paths(g_dir, bin_num)
                gts.append(gt)
        self.ground_truth = np.concatenate(gts) if gts else np.zeros((0, 63), np.float32)

    def __len__(self) -> int:
        return len(self.paths)

    def __getitem__(self, index: int):
        header, depth = packing.read_bin(self.paths[index])
        return header, depth, self.ground_truth[index]


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


class VoxelLoader:
    """Batches of voxel grids produced on the fly.

    A worker thread reads and packs the next batches (file I/O + numpy) while the GPU voxelizes and
    trains on the current one; uploads go through pinned buffers on a side stream, and the compute
    stream waits on an event, so H2D copies overlap the previous batch's kernels
    (BASELINE.json configs[2]).
    """

    def __init__(self, dataset: MSRADepthDataset, batch_size: int, device, res: int = 32,
                 shuffle: bool = False, seed: int = 0, prefetch: int = 2, drop_last: bool = False,
                 rank: int = 0, world: int = 1):
        self.ds, self.bs, self.device, self.res = dataset, int(batch_size), torch.device(device), res
        self.shuffle, self.seed, self.prefetch, self.drop_last = shuffle, seed, max(1, prefetch), drop_last
        self.rank, self.world = rank, world
        self.epoch = 0

    def _indices(self) -> np.ndarray:
        n = len(self.ds)
        idx = np.arange(n)
        if self.shuffle:
            np.random.default_rng(self.seed + self.epoch).shuffle(idx)
        return idx[self.rank::self.world]  # frames shard across ranks; no collective involved

    def __len__(self) -> int:
        n = len(self._indices())
        return n // self.bs if self.drop_last else (n + self.bs - 1) // self.bs

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, ...]]:
        idx = self._indices()
        self.epoch += 1
        batches = [idx[i:i + self.bs] for i in range(0, len(idx), self.bs)]
        if self.drop_last and batches and len(batches[-1]) < self.bs:
            batches.pop()
        q: "queue.Queue" = queue.Queue(maxsize=self.prefetch)

        def work():
            try:
                for b in batches:
                    q.put(collate_frames([self.ds[int(i)] for i in b]))
                q.put(None)
            except BaseException as e:  # surface I/O errors in the consumer
                q.put(e)

        t = threading.Thread(target=work, daemon=True)
        t.start()
        copy_stream = torch.cuda.Stream(device=self.device)
        while True:
            item = q.get()
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            pk, gt = item
            with torch.cuda.stream(copy_stream):
                depth, offsets, headers = pk.to_torch(self.device, pin=True, non_blocking=True)
                tgt = torch.from_numpy(np.ascontiguousarray(gt)).pin_memory().to(self.device, non_blocking=True)
                ready = torch.cuda.Event()
                ready.record(copy_stream)
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ready)
            for ten in (depth, offsets, headers, tgt):
                ten.record_stream(cur)
            out = voxelize(depth, offsets, headers, res=self.res)
            yield out.tsdf, tgt, out.max_l, out.mid_p
        t.join()
