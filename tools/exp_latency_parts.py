#!/usr/bin/env python3
"""Where a small-batch launch spends its time (GPU box): back-to-back time of a trivial kernel, of phase 1 alone
(tsdf_aabb_hip) and of the whole voxelizer, for n = 1 and 16 full frames."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
dev = torch.device("cuda:0")
def b2b(fn, K=300):
    for _ in range(30): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(K): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / K * 1e3
for kind in ("full", "crop"):
    depth, off, hdr = synth.synth_batch(16, kind, seed0=0)
    for n in (1, 16):
        td = torch.from_numpy(depth[: off[n]]).to(dev); to = torch.from_numpy(off[: n + 1]).to(dev); th = torch.from_numpy(hdr[:n]).to(dev)
        out = pkg.voxelize(td, to, th)
        gt = torch.zeros((n, 63), device=dev); nor = torch.empty_like(gt)
        t_triv = b2b(lambda: pkg.normalize_joints(gt, out.max_l, out.mid_p, out=nor))
        t_aabb = b2b(lambda: pkg.aabb(td, to, th))
        t_vox = b2b(lambda: pkg.voxelize(td, to, th, out=out))
        t_v64 = b2b(lambda: pkg.voxelize(td, to, th, res=64), 100)
        print(f"{kind} n={n}: trivial kernel {t_triv:.2f} us | phase 1 only (1 workgroup/frame, allocs outputs) {t_aabb:.2f} us | "
              f"voxelize 32^3 {t_vox:.2f} us | 64^3 (allocs) {t_v64:.2f} us")
