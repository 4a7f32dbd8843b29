#!/usr/bin/env python3
"""Back-to-back 16-frame launches (the reference's batch size) for several builds of the library (GPU box):
    python tools/exp_small_batch.py libtsdf_hip.so libtsdf_hip_x30.so ...     env: PROF_KIND=crop|full  PROF_N=16
Each library runs K launches of the indexed entry with labels (what MSRA_Dataset issues per batch), index by value."""
import ctypes, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
dev = torch.device("cuda:0")
kind = os.environ.get("PROF_KIND", "crop"); n = int(os.environ.get("PROF_N", "16")); K = 2000
depth, off, hdr = synth.synth_batch(512, kind, seed0=0)
td, to, th = (torch.from_numpy(a).to(dev) for a in (depth, off, hdr))
gt = torch.zeros((512, 63), device=dev)
out = pkg.voxelize(td[: off[n]], to[: n + 1].contiguous(), th[:n].contiguous())
gn, gg = torch.empty((n, 63), device=dev), torch.empty((n, 63), device=dev)
lab = pkg._lib.TsdfLabels(gt.data_ptr(), 21, 1, gn.data_ptr(), gg.data_ptr())
rng = np.random.default_rng(0)
idx = [np.ascontiguousarray(rng.integers(0, 512, n).astype(np.int64)) for _ in range(64)]
stream = torch.cuda.current_stream().cuda_stream


def load(name):
    cand = [os.path.join(ROOT, "build", name), os.path.join(ROOT, "handposeestimation-with-3d-cnns_amd", name)]
    L = ctypes.CDLL(next(c for c in cand if os.path.exists(c)))
    L.tsdf_voxelize_indexed_host_hip.restype = ctypes.c_int
    L.tsdf_voxelize_indexed_host_hip.argtypes = pkg._lib.load().tsdf_voxelize_indexed_hip.argtypes
    return L


def run(L, k):
    rc = L.tsdf_voxelize_indexed_host_hip(td.data_ptr(), td.numel(), to.data_ptr(), th.data_ptr(), 512, idx[k & 63].ctypes.data, n, 32,
                                          None, 0, stream, out.tsdf.data_ptr(), out.max_l.data_ptr(), out.mid_p.data_ptr(),
                                          out.status.data_ptr(), ctypes.byref(lab))
    assert rc == 0


libs = [(a, load(a)) for a in sys.argv[1:]]
ref = None
for name, L in libs:
    run(L, 0); torch.cuda.synchronize()
    if ref is None:
        ref = out.tsdf.clone()
    assert torch.equal(out.tsdf, ref), name
for rep in range(3):
    for name, L in libs:
        for k in range(50):
            run(L, k)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for k in range(K):
            run(L, k)
        b.record(); torch.cuda.synchronize()
        print(f"{kind} n={n} {name:24s} {a.elapsed_time(b) / K * 1e3:7.2f} us per launch")
