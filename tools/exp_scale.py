#!/usr/bin/env python3
"""Back-to-back time per launch vs batch size for the library selected by TSDF_HIP_LIB."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
dev = torch.device("cuda:0")
kind = os.environ.get("PROF_KIND", "full")
R = int(os.environ.get("PROF_R", "32"))
NMAX = 4096
depth, off, hdr = synth.synth_batch(1024, kind, seed0=0)
# tile the 1024 seeded frames to NMAX
reps = NMAX // 1024
depth = np.tile(depth, reps); hdr = np.tile(hdr, (reps, 1))
off = np.concatenate([[0], np.cumsum(np.tile(np.diff(off), reps))]).astype(np.int64)
td, to, th = (torch.from_numpy(a).to(dev) for a in (depth, off, hdr))
print(os.environ.get("TSDF_HIP_LIB", "default lib"), kind)
NS = [int(v) for v in os.environ.get('PROF_NS', '256,512,1024,2048,4096').split(',')]
for n in NS:
    o = to[: n + 1].contiguous(); h = th[:n].contiguous()
    out = pkg.voxelize(td, o, h, res=R)
    for _ in range(3): pkg.voxelize(td, o, h, res=R, out=out)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    K = 20
    for _ in range(K): pkg.voxelize(td, o, h, res=R, out=out)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / K * 1e3
    nbytes = 4 * int(off[n]) + n * (48 + 12 * R ** 3)
    print(f"  n={n:5d}: {us:8.1f} us/launch  {us/n*256:6.2f} us per 256 frames  {nbytes/us/1e3:7.1f} GB/s algorithmic")
    del out
