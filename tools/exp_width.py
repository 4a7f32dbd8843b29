#!/usr/bin/env python3
"""Does the row length matter for the phase-1 stream?  Same 76,800-pixel payloads declared as
320x240, 256x300 and 512x150 crops (GPU box)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
dev = torch.device("cuda:0")
depth, off, hdr = synth.synth_batch(1024, "full", seed0=0)
td, to = torch.from_numpy(depth).to(dev), torch.from_numpy(off).to(dev)
def timeit(fn, K=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(K): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / K * 1e3
for (w, h) in ((320, 240), (256, 300), (512, 150), (128, 600), (64, 1200)):
    hh = hdr.copy(); hh[:, 4] = w; hh[:, 5] = h
    th = torch.from_numpy(hh).to(dev)
    t_a = timeit(lambda: pkg.aabb(td, to, th))
    out = pkg.voxelize(td, to, th)
    t_f = timeit(lambda: pkg.voxelize(td, to, th, out=out))
    print(f"{w:4d}x{h:4d}: phase-1-only {t_a:7.1f} us ({depth.size*4/t_a/1e3:6.1f} GB/s)   full {t_f:7.1f} us")
