#!/usr/bin/env python3
"""BASELINE.json configs[3] on ONE GPU: all 9 MSRA subjects' worth of frames (~76.5 k MSRA-like crops),
device resident, one launch.  (Across 8 GPUs the frames shard by rank with no collective: bench.py.)"""
import importlib, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
dev = torch.device("cuda:0")
N = int(os.environ.get("ALL_FRAMES", "76500"))
base = [synth.synth_frame(i, "crop") for i in range(1500)]
pk = pkg.packing.pack_frames([base[i % 1500] for i in range(N)])
td, to, th = pk.to_torch(dev)
out = pkg.voxelize(td, to, th)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 3
a.record()
for _ in range(K):
    pkg.voxelize(td, to, th, out=out)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / K
nbytes = 4 * pk.depth.size + N * (48 + 12 * 32 ** 3)
print(json.dumps({"frames": N, "ms_per_pass": round(ms, 3), "frames_per_s": round(N / ms * 1e3),
                  "GBps_algorithmic": round(nbytes / ms / 1e6, 1), "out_GB": round(N * 12 * 32 ** 3 / 1e9, 1),
                  "all_ok": bool((out.status == 0).all())}))
