#!/usr/bin/env python3
"""K launches of the voxelizer from an arbitrary build of the library (run under rocprofv3 --pmc ...):
    python3 tools/exp_pmc_lib.py libtsdf_hip_r01.so          (bare names: build/, then the package)
    env: PROF_KIND=full|crop PROF_N=1024 PROF_R=32 PROF_K=8; PROF_AUG=1: the augmented entry, reference-distribution maps"""
import ctypes, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
augment = importlib.import_module("handposeestimation-with-3d-cnns_amd.augment")
name = sys.argv[1]
if not os.path.isabs(name):
    cand = [os.path.join(ROOT, "build", name), os.path.join(ROOT, "handposeestimation-with-3d-cnns_amd", name)]
    name = next((c for c in cand if os.path.exists(c)), cand[-1])
L = ctypes.CDLL(name)
vp = ctypes.c_void_p
L.tsdf_voxelize_hip.restype = ctypes.c_int
L.tsdf_voxelize_hip.argtypes = [vp, ctypes.c_int64, vp, vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp, vp, vp, vp, vp]
L.tsdf_voxelize_aug_hip.restype = ctypes.c_int
L.tsdf_voxelize_aug_hip.argtypes = [vp, ctypes.c_int64, vp, vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp, vp, vp, vp, vp, vp]
kind = os.environ.get("PROF_KIND", "full"); n = int(os.environ.get("PROF_N", "1024"))
R = int(os.environ.get("PROF_R", "32")); K = int(os.environ.get("PROF_K", "8")); AUG = os.environ.get("PROF_AUG") == "1"
dev = torch.device("cuda:0")
depth, off, hdr = synth.synth_batch(n, kind, seed0=0, threads=8)
td, to, th = (torch.from_numpy(a).to(dev) for a in (depth, off, hdr))
t = torch.empty((n, 3, R, R, R), dtype=torch.float32, device=dev)
ml = torch.empty(n, dtype=torch.float32, device=dev); mp = torch.empty((n, 3), dtype=torch.float32, device=dev)
st = torch.empty(n, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
if AUG:
    import oracle  # (only for the grid centres the maps are drawn around; nothing is checked here)
    mid = oracle.voxelize(depth, off, hdr, R=R, n_threads=16, want_tsdf=False)["mid_p"]
    txf = torch.from_numpy(augment.random_affines(mid, rng=np.random.RandomState(2026))[0]).to(dev)
for _ in range(K):
    if AUG:
        rc = L.tsdf_voxelize_aug_hip(td.data_ptr(), td.numel(), to.data_ptr(), th.data_ptr(), n, R, None, 0, s, txf.data_ptr(),
                                     t.data_ptr(), ml.data_ptr(), mp.data_ptr(), st.data_ptr())
    else:
        rc = L.tsdf_voxelize_hip(td.data_ptr(), td.numel(), to.data_ptr(), th.data_ptr(), n, R, None, 0, s,
                                 t.data_ptr(), ml.data_ptr(), mp.data_ptr(), st.data_ptr())
    assert rc == 0, rc
torch.cuda.synchronize()
