#!/usr/bin/env python3
"""GPU-side small-batch latency without the host's launch cost (GPU box): 20 launches of tsdf_voxelize_hip captured
into one hipGraph and replayed; per-launch time for n = 1..256 full frames, for several builds of the library.
    python tools/exp_latency_graph.py libtsdf_hip_r01.so libtsdf_hip.so"""
import ctypes, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
PKG = os.path.join(ROOT, "handposeestimation-with-3d-cnns_amd")
dev = torch.device("cuda:0")
vp = ctypes.c_void_p
kind = os.environ.get("LAT_KIND", "full")
depth, off, hdr = synth.synth_batch(256, kind, seed0=0)
res = {}
for name in sys.argv[1:]:
    L = ctypes.CDLL(os.path.join(PKG, name))
    L.tsdf_voxelize_hip.restype = ctypes.c_int
    L.tsdf_voxelize_hip.argtypes = [vp, ctypes.c_int64, vp, vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp, vp, vp, vp, vp]
    for n in (1, 16, 64, 256):
        td = torch.from_numpy(depth[: off[n]]).to(dev); to = torch.from_numpy(off[: n + 1]).to(dev); th = torch.from_numpy(hdr[:n]).to(dev)
        t = torch.empty((n, 3, 32, 32, 32), device=dev); ml = torch.empty(n, device=dev); mp = torch.empty((n, 3), device=dev)
        st = torch.empty(n, dtype=torch.int32, device=dev)
        def launch(stream):
            assert L.tsdf_voxelize_hip(td.data_ptr(), td.numel(), to.data_ptr(), th.data_ptr(), n, 32, None, 0, stream,
                                       t.data_ptr(), ml.data_ptr(), mp.data_ptr(), st.data_ptr()) == 0
        launch(torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(); cs = torch.cuda.Stream(dev)
        with torch.cuda.stream(cs):
            with torch.cuda.graph(g, stream=cs):
                for _ in range(20):
                    launch(cs.cuda_stream)
        for _ in range(3): g.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(20): g.replay()
        b.record(); torch.cuda.synchronize()
        res[(name, n)] = a.elapsed_time(b) / 400 * 1e3
        del g
for n in (1, 16, 64, 256):
    print(f"{kind} n={n:3d}: " + "   ".join(f"{name} {res[(name, n)]:.2f} us" for name in sys.argv[1:]))
