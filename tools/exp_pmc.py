#!/usr/bin/env python3
"""aabb-only and full launches for PMC collection (run under rocprofv3 --pmc ...)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
dev = torch.device("cuda:0")
depth, off, hdr = synth.synth_batch(1024, "full", seed0=0, threads=8)
td, to, th = (torch.from_numpy(a).to(dev) for a in (depth, off, hdr))
out = pkg.voxelize(td, to, th)
mode = os.environ.get("PMC_MODE", "aabb")
# PMC_ROTATE=k (modes full / aabb): k buffer sets launched in rotation, as bench.py's timed region does — the same 1024
# frames in every set, rolled by 170 frames from set to set (all 76,800 px: the offsets stay valid), so that the
# counters see the same work from 4.3 GB of distinct addresses instead of one batch a cache may still hold
ROT = max(1, int(os.environ.get("PMC_ROTATE", "1")))
sets = [(td, th, out)]
for k in range(1, ROT):
    dk = td.view(1024, -1).roll(170 * k, 0).reshape(-1).contiguous()
    sets.append((dk, th, pkg.voxelize(dk, to, th)))
# Launches per run.  Trace runs of the 64^3 kernels take 100: the augmented kernel runs 5-15 % slower for its first ~30
# launches after the GPU did something else (profiles/r04/warmup.log), so a 30-launch average (rounds 2-3: 739.5 us) is
# the transient, not the kernel; the summaries report the average over dispatches 41.. as well.  Counter runs keep 30.
K = int(os.environ.get("PMC_LAUNCHES", "100" if (mode.startswith("aug") or mode == "r64") and os.environ.get("PMC_TRACE") else "30"))
if mode.startswith("aug"):       # aug32 / aug64: the fused-augmentation kernel (BASELINE configs[4] at aug64)
    R = int(mode[3:] or 64)
    xf = torch.from_numpy(pkg.augment.random_affines(out.mid_p.cpu().numpy(), rng=1)[0]).to(dev)
    oa = pkg.voxelize_aug(td, to, th, xf, res=R)
    for _ in range(K):
        pkg.voxelize_aug(td, to, th, xf, res=R, out=oa)
elif mode == "r64":              # 1024 full frames -> 64^3, plain
    o64 = pkg.voxelize(td, to, th, res=64)
    for _ in range(K):
        pkg.voxelize(td, to, th, res=64, out=o64)
elif mode == "crop":             # 1024 MSRA-like crops
    depth, off, hdr = synth.synth_batch(1024, "crop", seed0=0)
    td, to, th = (torch.from_numpy(a).to(dev) for a in (depth, off, hdr))
    for _ in range(K):
        pkg.voxelize(td, to, th, out=out)
else:
    for i in range(K):
        dk, hk, ok = sets[i % ROT]
        if mode == "aabb":
            pkg.aabb(dk, to, hk)
        else:
            pkg.voxelize(dk, to, hk, out=ok)
torch.cuda.synchronize()
