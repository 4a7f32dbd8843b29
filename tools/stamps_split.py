#!/usr/bin/env python3
"""Timeline of the split kernel (small batches) from the diagnostic library's stamps: n frames, S workgroups each."""
import ctypes, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
L = ctypes.CDLL(os.path.join(ROOT, "build", "libtsdf_hip_stamps.so"))
vp = ctypes.c_void_p
L.tsdf_voxelize_hip.restype = ctypes.c_int
L.tsdf_voxelize_hip.argtypes = [vp, ctypes.c_int64, vp, vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp, vp, vp, vp, vp]
L.tsdf_debug_read_stamps.restype = ctypes.c_int
L.tsdf_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
for kind in ("full", "crop"):
    for n in (1, 16):
        depth, off, hdr = synth.synth_batch(n, kind, seed0=0)
        td, to, th = (torch.from_numpy(a).to(dev) for a in (depth, off, hdr))
        t = torch.empty((n, 3, 32, 32, 32), device=dev); ml = torch.empty(n, device=dev); mp = torch.empty((n, 3), device=dev)
        st = torch.empty(n, dtype=torch.int32, device=dev)
        for _ in range(5):
            assert L.tsdf_voxelize_hip(td.data_ptr(), td.numel(), to.data_ptr(), th.data_ptr(), n, 32, None, 0,
                                       torch.cuda.current_stream().cuda_stream, t.data_ptr(), ml.data_ptr(), mp.data_ptr(), st.data_ptr()) == 0
        torch.cuda.synchronize()
        SL, FR, BL = 16, 8, 512
        buf = np.zeros(BL * FR * SL, np.uint64)
        assert L.tsdf_debug_read_stamps(buf.ctypes.data, buf.size) == buf.size
        s = buf.reshape(BL, FR, SL).astype(np.int64)[: min(512, n * 8), 0, :]
        t0 = s[:, 0].min()
        r = (s[:, [0, 1, 3, 4, 6, 9, 11]] - t0) / 100.0
        m = np.median(r, axis=0)
        print(f"{kind} n={n}: {s.shape[0]} workgroups; medians (us) start {m[0]:.2f}  rows streamed {m[1]:.2f}  partials in LDS {m[2]:.2f}  "
              f"extents known {m[3]:.2f}  staged+tables {m[4]:.2f}  voxels issued {m[5]:.2f}  stores done {m[6]:.2f}  (max end {r[:,6].max():.2f})")
        if os.environ.get("STAMPS_ROWS"):   # every workgroup of the first frame: who waits for whom in the exchange
            np.set_printoptions(precision=2, suppress=True, linewidth=160)
            print(r[: s.shape[0] // n])
            print('published / exchange done:', ((s[: s.shape[0] // n][:, [13, 12]] - t0) / 100.0).tolist())
