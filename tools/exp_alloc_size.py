#!/usr/bin/env python3
"""Does the SIZE of the allocation that backs the output volume decide whether it is a "fast" or a "slow" buffer?
(GPU box; follow-up of exp_out_buffers.py)  128 frames -> 128^3 (3 GiB of volume, the most placement-sensitive case)
into the first 3 GiB of fresh allocations of several sizes, each size tried several times."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
dev = torch.device("cuda:0")
R = int(os.environ.get("PROF_R", "128")); n = int(os.environ.get("PROF_N", "128"))
depth, off, hdr = synth.synth_batch(n, "full", seed0=0)
td, to, th = (torch.from_numpy(a).to(dev) for a in (depth, off, hdr))
small = pkg.voxelize(td, to, th, res=32)
vol = n * 3 * R ** 3


def run(buf, K=8):
    o = pkg.TsdfBatch(buf[:vol].view(n, 3, R, R, R), small.max_l, small.mid_p, small.status)
    for _ in range(2):
        pkg.voxelize(td, to, th, res=R, out=o)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(K):
        pkg.voxelize(td, to, th, res=R, out=o)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / K * 1e3


GiB = 1 << 30
sizes = [("3 GiB (exact)", 3 * GiB), ("3 GiB + 2 MiB", 3 * GiB + (2 << 20)), ("4 GiB", 4 * GiB), ("6 GiB", 6 * GiB),
         ("8 GiB", 8 * GiB), ("3.5 GiB", 3 * GiB + GiB // 2)]
for rep in range(3):
    for name, nbytes in sizes:
        torch.cuda.empty_cache()
        buf = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
        t = run(buf)
        print(f"rep {rep}  {name:16s} at {buf.data_ptr():#x}: {t:8.1f} us")
        del buf
print("several live at once:")
live = [torch.empty(3 * GiB // 4, dtype=torch.float32, device=dev) for _ in range(6)]
for i, b in enumerate(live):
    print(f"  buffer {i} at {b.data_ptr():#x}: {run(b):8.1f} us")
