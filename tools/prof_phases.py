#!/usr/bin/env python3
"""Phase split of the fused kernel on the bench workload (run on the GPU box).

Times (HIP events, current stream) the full kernel and the phase-1-only entry
(tsdf_aabb_hip) on 1024 full frames and on 1024 MSRA-like crops.  Used under rocprofv3 too:
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python3 tools/prof_phases.py
"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")

N = int(os.environ.get("PROF_FRAMES", "1024"))
ITERS = int(os.environ.get("PROF_ITERS", "10"))
KINDS = os.environ.get("PROF_KINDS", "full,crop").split(",")
dev = torch.device("cuda:0")


def timeit(fn, iters=ITERS):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return np.median(ts), np.min(ts)


for kind in KINDS:
    depth, off, hdr = synth.synth_batch(N, kind, seed0=0)
    td, to, th = (torch.from_numpy(a).to(dev) for a in (depth, off, hdr))
    out = pkg.voxelize(td, to, th)
    for layout in ("czyx", "cxyz"):
        med, mn = timeit(lambda: pkg.voxelize(td, to, th, layout=layout, out=out))
        nbytes = 4 * depth.size + N * (48 + 12 * 32 ** 3)
        print(f"{kind:5s} full kernel {layout}: median {med:8.1f} us  min {mn:8.1f} us  "
              f"{nbytes / med / 1e3:7.1f} GB/s algorithmic  ({N / med:.3f} Mframes/s)")
    med, mn = timeit(lambda: pkg.aabb(td, to, th))
    print(f"{kind:5s} phase 1 only      : median {med:8.1f} us  min {mn:8.1f} us  "
          f"{4 * depth.size / med / 1e3:7.1f} GB/s read")
