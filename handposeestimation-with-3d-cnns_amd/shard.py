"""Frame sharding across GPUs (one process per GPU, no data-path collective).

Frames are independent end to end (the AABB is per frame, pre/tsdf_numba.py:140-147), so
multi-GPU voxelization is a contiguous split of the frame range; each rank voxelizes its own
shard on its own GPU and keeps the result there for its CNN replica.  The only communication
a job needs is host-side (barrier, gathering a handful of timing numbers).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np


def shard_bounds(n: int, world: int, weights: Optional[Sequence[float]] = None) -> List[Tuple[int, int]]:
    """Contiguous [begin, end) per rank covering [0, n).

    Without ``weights`` the split is by frame count (sizes differ by at most 1).  With
    ``weights`` (e.g. pixels per frame) the cut points balance the cumulative weight, which
    matters for MSRA crops whose bounding boxes vary ~3x in area.
    """
    if world < 1:
        raise ValueError("world must be >= 1")
    if n < 0:
        raise ValueError("n must be >= 0")
    if weights is None:
        cuts = [(n * r) // world for r in range(world + 1)]
    else:
        w = np.asarray(weights, dtype=np.float64)
        if w.shape != (n,):
            raise ValueError("weights must have one entry per frame")
        if n == 0 or w.sum() <= 0:
            cuts = [(n * r) // world for r in range(world + 1)]
        else:
            c = np.concatenate([[0.0], np.cumsum(w)])
            targets = c[-1] * np.arange(world + 1) / world
            cuts = [int(np.searchsorted(c, t, side="left")) for t in targets]
            cuts[0], cuts[-1] = 0, n
            for r in range(1, world + 1):  # monotone
                cuts[r] = max(cuts[r], cuts[r - 1])
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def shard_for_rank(n: int, rank: int, world: int, weights: Optional[Sequence[float]] = None) -> Tuple[int, int]:
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    return shard_bounds(n, world, weights)[rank]
