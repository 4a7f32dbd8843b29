#!/usr/bin/env python3
"""Per-phase timeline of the fused kernel from in-kernel stamps (diagnostic library only).

Build:  make -C handposeestimation-with-3d-cnns_amd/csrc stamps     (build/libtsdf_hip_stamps.so)
Run on the GPU box:  python tools/stamps.py
Slots: 0 frame start | 1 rows streamed | 2 row/column extremes back-projected | 3 wave partials in LDS
       4 AABB known (after barrier + final reduce + glue) | 5 z table | 6 projection tables
       7 stage copied | 8 after stage barrier | 9 phase 2 done.   Times are 10 ns ticks (s_memrealtime).
"""
import ctypes, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
# the diagnostic library is loaded directly (any ABI version: only tsdf_voxelize_hip is called)
L = ctypes.CDLL(os.environ.get("STAMPS_LIB") or os.path.join(ROOT, "build", "libtsdf_hip_stamps.so"))
vp = ctypes.c_void_p
L.tsdf_voxelize_hip.restype = ctypes.c_int
L.tsdf_voxelize_hip.argtypes = [vp, ctypes.c_int64, vp, vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp, vp, vp, vp, vp]
dev = torch.device("cuda:0")
N = int(os.environ.get("PROF_FRAMES", "1024"))
kind = os.environ.get("PROF_KIND", "full")
depth, off, hdr = synth.synth_batch(N, kind, seed0=0)
td, to, th = (torch.from_numpy(a).to(dev) for a in (depth, off, hdr))
t = torch.empty((N, 3, 32, 32, 32), dtype=torch.float32, device=dev)
ml = torch.empty(N, dtype=torch.float32, device=dev); mp = torch.empty((N, 3), dtype=torch.float32, device=dev)
st = torch.empty(N, dtype=torch.int32, device=dev)
for _ in range(4):
    rc = L.tsdf_voxelize_hip(td.data_ptr(), td.numel(), to.data_ptr(), th.data_ptr(), N, 32, None, 0,
                             torch.cuda.current_stream().cuda_stream, t.data_ptr(), ml.data_ptr(), mp.data_ptr(), st.data_ptr())
    assert rc == 0
torch.cuda.synchronize()
SL, FR, BL = 16, 8, 512
buf = np.zeros(BL * FR * SL, np.uint64)
L.tsdf_debug_read_stamps.restype = ctypes.c_int
L.tsdf_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
got = L.tsdf_debug_read_stamps(buf.ctypes.data, buf.size)
assert got == buf.size
s = buf.reshape(BL, FR, SL).astype(np.int64)
s[:, :, 7] = np.where(s[:, :, 7] == 0, s[:, :, 6], s[:, :, 7])  # builds without a staging copy do not stamp slot 7
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
