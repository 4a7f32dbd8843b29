#!/usr/bin/env python3
"""K launches of tsdf_voxelize_hip from an arbitrary build of the library (run under rocprofv3 --pmc ...):
    python3 tools/exp_pmc_lib.py libtsdf_hip_r01.so          env: PROF_KIND=full|crop PROF_N=1024"""
import ctypes, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
name = sys.argv[1]
L = ctypes.CDLL(name if os.path.isabs(name) else os.path.join(ROOT, "handposeestimation-with-3d-cnns_amd", name))
vp = ctypes.c_void_p
L.tsdf_voxelize_hip.restype = ctypes.c_int
L.tsdf_voxelize_hip.argtypes = [vp, ctypes.c_int64, vp, vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp, vp, vp, vp, vp]
kind = os.environ.get("PROF_KIND", "full"); n = int(os.environ.get("PROF_N", "1024"))
dev = torch.device("cuda:0")
depth, off, hdr = synth.synth_batch(n, kind, seed0=0)
td, to, th = (torch.from_numpy(a).to(dev) for a in (depth, off, hdr))
t = torch.empty((n, 3, 32, 32, 32), dtype=torch.float32, device=dev)
ml = torch.empty(n, dtype=torch.float32, device=dev); mp = torch.empty((n, 3), dtype=torch.float32, device=dev)
st = torch.empty(n, dtype=torch.int32, device=dev)
for _ in range(8):
    assert L.tsdf_voxelize_hip(td.data_ptr(), td.numel(), to.data_ptr(), th.data_ptr(), n, 32, None, 0,
                               torch.cuda.current_stream().cuda_stream, t.data_ptr(), ml.data_ptr(), mp.data_ptr(), st.data_ptr()) == 0
torch.cuda.synchronize()
