#!/usr/bin/env python3
"""Does a kernel run slower on buffers allocated after 30 GB went through torch's caching allocator and
empty_cache()? (GPU box)  1024 crops resident, fresh output volume each time."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
dev = torch.device("cuda:0")
depth, off, hdr = synth.synth_batch(1024, "crop", seed0=0)
def measure(tag):
    td, to, th = (torch.from_numpy(a).to(dev) for a in (depth, off, hdr))
    out = pkg.voxelize(td, to, th)
    for _ in range(3): pkg.voxelize(td, to, th, out=out)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(20): pkg.voxelize(td, to, th, out=out)
    b.record(); torch.cuda.synchronize()
    k_us = a.elapsed_time(b) / 20 * 1e3
    # H2D of 64 MB from pinned memory into a fresh device buffer
    h = torch.empty(16 * 1024 * 1024, dtype=torch.float32).pin_memory(); d = torch.empty_like(h, device=dev)
    d.copy_(h, non_blocking=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); gbps = 10 * h.numel() * 4 / (time.perf_counter() - t0) / 1e9
    # the same copy while the kernel runs on another stream
    cs = torch.cuda.Stream(dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(cs):
        for _ in range(10): d.copy_(h, non_blocking=True)
    for _ in range(100): pkg.voxelize(td, to, th, out=out)
    cs.synchronize(); gb2 = 10 * h.numel() * 4 / (time.perf_counter() - t0) / 1e9
    torch.cuda.synchronize()
    print(f"{tag}: kernel {k_us:.1f} us, H2D {gbps:.1f} GB/s alone, {gb2:.1f} GB/s beside the kernel", flush=True)
    del td, to, th, out, h, d
measure("fresh process")
big = torch.empty(30 * 1024**3 // 4, device=dev); del big; torch.cuda.empty_cache()
measure("after 30 GB alloc + empty_cache")
big = torch.empty(30 * 1024**3 // 4, device=dev); big.zero_(); torch.cuda.synchronize(); del big; torch.cuda.empty_cache()
measure("after 30 GB touched + empty_cache")
