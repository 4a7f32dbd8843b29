#!/usr/bin/env python3
"""K1 (extents kernel) alone vs K1+K2, back to back, on the diagnostic library (run under rocprofv3)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TSDF_HIP_LIB"] = os.path.join(ROOT, "handposeestimation-with-3d-cnns_amd", "libtsdf_hip_stamps.so")
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
dev = torch.device("cuda:0")
depth, off, hdr = synth.synth_batch(1024, "full", seed0=0)
td, to, th = (torch.from_numpy(a).to(dev) for a in (depth, off, hdr))
out = pkg.voxelize(td, to, th)
torch.cuda.synchronize()
def run(k, label):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): pkg.voxelize(td, to, th, out=out)
    torch.cuda.synchronize(); a.record()
    for _ in range(k): pkg.voxelize(td, to, th, out=out)
    b.record(); torch.cuda.synchronize()
    print(f"{label}: {a.elapsed_time(b)/k*1e3:.1f} us per call (back to back x{k})")
run(20, "K1+K2")
os.environ["TSDF_DEBUG_SKIP_K2"] = "1"
run(20, "K1 only (depth resident in Infinity Cache? 315 MB > 256 MiB)")
del os.environ["TSDF_DEBUG_SKIP_K2"]
os.environ["TSDF_DEBUG_FUSED"] = "1"
run(20, "fused single launch (512 threads, phase 1 inside)")
