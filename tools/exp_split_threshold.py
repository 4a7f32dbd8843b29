#!/usr/bin/env python3
"""(Needs a build with tools/patches/r05_removed_knobs.diff applied: the product no longer reads TSDF_SPLIT_MAXN.)
Back-to-back launch time for mid-size batches with the split kernel forced on (TSDF_SPLIT_MAXN) or off: child
processes, one per setting (the knob is read once per process)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import importlib, json, os, sys, numpy as np, torch
sys.path.insert(0, %r)
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd"); synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
dev = torch.device("cuda:0"); res = {}
for kind in ("full", "crop"):
    depth, off, hdr = synth.synth_batch(512, kind, seed0=0)
    for n in (96, 128, 192, 256, 384, 512):
        td = torch.from_numpy(depth[: off[n]]).to(dev); to = torch.from_numpy(off[: n + 1]).to(dev); th = torch.from_numpy(hdr[:n]).to(dev)
        out = pkg.voxelize(td, to, th)
        for _ in range(20): pkg.voxelize(td, to, th, out=out)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(200): pkg.voxelize(td, to, th, out=out)
        b.record(); torch.cuda.synchronize()
        res[f"{kind}_{n}"] = round(a.elapsed_time(b) / 200 * 1e3, 2)
print(json.dumps(res))
''' % ROOT
out = {}
for maxn in ("128", "512"):
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TSDF_SPLIT_MAXN=maxn), capture_output=True, text=True)
    out[maxn] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else r.stderr[-500:]
for k in out["128"]:
    print(k, "fused-above-128:", out["128"][k], "us   split-up-to-512:", out["512"][k], "us")
