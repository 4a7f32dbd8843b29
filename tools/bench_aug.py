#!/usr/bin/env python3
"""BASELINE.json configs[4]: 64^3 grid with the 3-D augmentation fused into the TSDF kernel (GPU box)."""
import importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
dev = torch.device("cuda:0")
n = int(os.environ.get("AUG_FRAMES", "1024"))
res = {}
for kind in ("full", "crop"):
    depth, off, hdr = synth.synth_batch(n, kind, seed0=0)
    td, to, th = (torch.from_numpy(a).to(dev) for a in (depth, off, hdr))
    mid = pkg.voxelize(td, to, th).mid_p.cpu().numpy()
    xf = torch.from_numpy(pkg.augment.random_affines(mid, rng=1)[0]).to(dev)
    for R in (32, 64):
        out = pkg.voxelize_aug(td, to, th, xf, res=R)
        outp = pkg.voxelize(td, to, th, res=R)
        def timeit(fn, K=10):
            for _ in range(2): fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); a.record()
            for _ in range(K): fn()
            b.record(); torch.cuda.synchronize()
            return a.elapsed_time(b) / K * 1e3
        ta = timeit(lambda: pkg.voxelize_aug(td, to, th, xf, res=R, out=out))
        tp = timeit(lambda: pkg.voxelize(td, to, th, res=R, out=outp))
        nbytes = 4 * depth.size + n * (48 + 12 * R ** 3)
        res[f"{kind}_R{R}"] = {"aug_us": round(ta, 1), "plain_us": round(tp, 1), "aug_frames_per_s": round(n / ta * 1e6),
                               "aug_GBps_algorithmic": round(nbytes / ta / 1e3, 1)}
        del out, outp
print(json.dumps(res, indent=1))
