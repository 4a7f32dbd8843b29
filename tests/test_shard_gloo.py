"""CPU tier, world_size 2 over gloo: the multi-GPU path is a plain per-rank frame split with no
data-path collective.  Each rank packs and "voxelizes" its own shard (here: through the oracle, the
checker — on the GPU box the same split feeds the HIP path, tests/test_parity_gpu.py checks that a
shard voxelized alone equals the same slice of the whole batch bit for bit); the only communication is
a barrier and a gather of per-rank digests, exactly what bench.py does with timings."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
    import oracle

    depth, off, hdr = pkg.synth.synth_batch(n, "crop", seed0=77)
    whole = pkg.packing.PackedFrames(depth, off, hdr)
    a, b = pkg.shard.shard_for_rank(n, rank, world, weights=whole.pixels)
    mine = whole.slice(a, b)
    res = oracle.voxelize(mine.depth, mine.offsets, mine.headers, R=32)
    dist.barrier()
    # gather only small per-rank facts (bounds + a digest), never the volumes
    digest = torch.tensor([a, b, float(np.abs(res["tsdf"]).sum()), float(res["max_l"].sum())],
                          dtype=torch.float64)
    gathered = [torch.zeros(4, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, digest)
    if rank == 0:
        ref = oracle.voxelize(depth, off, hdr, R=32)
        ok = gathered[0][0] == 0 and gathered[-1][1] == n
        for r in range(world):
            s, e = int(gathered[r][0]), int(gathered[r][1])
            ok = ok and (r == 0 or s == int(gathered[r - 1][1]))
            ok = ok and abs(float(np.abs(ref["tsdf"][s:e]).sum()) - float(gathered[r][2])) < 1e-9
            ok = ok and abs(float(ref["max_l"][s:e].sum()) - float(gathered[r][3])) < 1e-9
        # and rank 0's own shard is bitwise the same slice of the whole
        ok = ok and np.array_equal(res["tsdf"], ref["tsdf"][a:b])
        ret.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_frame_split_matches_single_process():
    world, n = 2, 12
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, ret)) for r in range(world)]
    for p in procs:
        p.start()
    ok = ret.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def _loader_worker(rank, world, port, pack_path, ret):
    """Each rank plans its epoch from the same pack (what VoxelLoader does before any GPU work) and assembles its
    batches on the host; the ranks then compare notes over gloo: shards are contiguous, disjoint and cover the set."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
    pk = pkg.packing.PackedFrames.load(pack_path, mmap=True)
    ds = pkg.MSRADepthDataset.from_packs([pk])
    batches = pkg.dataset.plan_batches(len(ds), 5, rank=rank, world=world, weights=ds.pixels())
    px = 0
    first = last = -1
    checksum = 0.0
    for b in batches:
        sub = ds.take(b)
        assert len(sub) == len(b) and sub.gt.shape == (len(b), 63)
        px += int(sub.depth.size)
        checksum += float(np.abs(np.asarray(sub.depth, np.float64)).sum())
        first = int(b[0]) if first < 0 else first
        last = int(b[-1])
    mine = torch.tensor([first, last, px, checksum], dtype=torch.float64)
    got = [torch.zeros(4, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(got, mine)
    if rank == 0:
        ok = int(got[0][0]) == 0 and int(got[-1][1]) == len(ds) - 1
        for r in range(1, world):
            ok = ok and int(got[r][0]) == int(got[r - 1][1]) + 1          # contiguous, disjoint, in rank order
        ok = ok and sum(int(g[2]) for g in got) == int(pk.depth.size)    # every pixel exactly once
        tot = float(np.abs(np.asarray(pk.depth, np.float64)).sum())
        ok = ok and abs(sum(float(g[3]) for g in got) - tot) <= 1e-6 * tot
        loads = [float(g[2]) for g in got]
        ok = ok and max(loads) / (sum(loads) / world) < 1.25             # balanced by pixels, not by frames
        ret.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_loader_plan_over_one_pack(tmp_path):
    """The N>1 input side: two ranks, one memory-mapped pack, contiguous pixel-balanced shards, no collective on
    the data path (the gather here is the test's own)."""
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
    frames = [pkg.synth.synth_frame(500 + i, "crop") for i in range(23)]
    pk = pkg.packing.pack_frames(frames)
    pk.gt = np.arange(23 * 63, dtype=np.float32).reshape(23, 63)
    path = str(tmp_path / "P0.tsdfpk")
    pk.save(path)
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loader_worker, args=(r, world, port, path, ret)) for r in range(world)]
    for p in procs:
        p.start()
    ok = ret.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
