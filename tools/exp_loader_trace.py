#!/usr/bin/env python3
"""VoxelLoader under rocprofv3 --memory-copy-trace --kernel-trace: 3 epochs over 17,000 crops."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
dev = torch.device("cuda:0")
crops = [synth.synth_frame(100000 + i, "crop") for i in range(1024)]
base = pkg.packing.pack_frames(crops)
n = 17000
reps = (n + 1023) // 1024
lens = np.tile(np.diff(base.offsets), reps)[:n]
off = np.zeros(n + 1, np.int64); np.cumsum(lens, out=off[1:])
pk = pkg.packing.PackedFrames(np.ascontiguousarray(np.tile(base.depth, reps)[: off[-1]]), off,
                              np.ascontiguousarray(np.tile(base.headers, (reps, 1))[:n]), np.zeros((n, 63), np.float32))
if os.environ.get("BIG_FIRST") == "1":   # the context bench.py's extras run in: 30 GB went through the allocator before
    big = torch.empty(30 * 1024**3 // 4, device=dev); del big; torch.cuda.empty_cache()
loader = pkg.VoxelLoader(pkg.MSRADepthDataset.from_packs([pk]), batch_size=1024, device=dev, max_pixels=1024 * 160 * 160)
for ep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); seen = 0
    for b in loader:
        seen += b.tsdf.shape[0]
    torch.cuda.synchronize(); print(ep, round(seen / (time.perf_counter() - t0)))
