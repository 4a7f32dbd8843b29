#!/usr/bin/env python3
"""Per-phase timeline of the fused kernel from in-kernel stamps (diagnostic library only).

Build:  make -C handposeestimation-with-3d-cnns_amd/csrc stamps     (libtsdf_hip_stamps.so)
Run on the GPU box:  python tools/stamps.py
Slots: 0 frame start | 1 rows streamed | 2 row/column extremes back-projected | 3 wave partials in LDS
       4 AABB known (after barrier + final reduce + glue) | 5 z table | 6 projection tables
       7 stage copied | 8 after stage barrier | 9 phase 2 done.   Times are 10 ns ticks (s_memrealtime).
"""
import ctypes, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TSDF_HIP_LIB", os.path.join(ROOT, "handposeestimation-with-3d-cnns_amd", "libtsdf_hip_stamps.so"))
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
L = pkg._lib.load()
dev = torch.device("cuda:0")
N = int(os.environ.get("PROF_FRAMES", "1024"))
kind = os.environ.get("PROF_KIND", "full")
depth, off, hdr = synth.synth_batch(N, kind, seed0=0)
td, to, th = (torch.from_numpy(a).to(dev) for a in (depth, off, hdr))
out = pkg.voxelize(td, to, th)
for _ in range(3):
    pkg.voxelize(td, to, th, out=out)
torch.cuda.synchronize()
SL, FR, BL = 16, 8, 512
buf = np.zeros(BL * FR * SL, np.uint64)
L.tsdf_debug_read_stamps.restype = ctypes.c_int
L.tsdf_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
got = L.tsdf_debug_read_stamps(buf.ctypes.data, buf.size)
assert got == buf.size
s = buf.reshape(BL, FR, SL).astype(np.int64)
nb = min(256, N)
iters = min(FR, (N + nb - 1) // nb)
t0 = s[:nb, 0, 0].min()
names = ["start", "rows", "backproj", "partials", "aabb", "ztab", "tabs", "staged", "barrier", "phase2"]
print(f"{N} {kind} frames, {nb} workgroups x {iters} frames; ticks are 10 ns")
for it in range(iters):
    rel = s[:nb, it, :10] - t0
    d = np.diff(rel, axis=1)
    print(f"frame #{it}: start median {np.median(rel[:,0])/100:7.2f} us  end median {np.median(rel[:,9])/100:7.2f} us  (max end {rel[:,9].max()/100:7.2f})")
    print("   segment medians (us): " + "  ".join(f"{names[i+1]} {np.median(d[:,i])/100:5.2f}" for i in range(9)))
print(f"kernel span (first start -> last end): {(s[:nb,:iters,9].max()-t0)/100:.2f} us")
