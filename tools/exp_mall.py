#!/usr/bin/env python3
"""Does re-launching the SAME 1024 frames flatter the headline (Infinity Cache: 256 MB; one batch is 315 MB in, 403 MB out)?
Times back-to-back launches over 1 batch (what bench.py does) and over K different batches and output buffers in rotation
(K x 718 MB of distinct memory).  GPU box:  python tools/exp_mall.py"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
dev = torch.device("cuda:0")
K = int(os.environ.get("MALL_BATCHES", "6"))
sets = []
for k in range(K):
    d, o, h = synth.synth_batch(1024, "full", seed0=1024 * k)
    td, to, th = (torch.from_numpy(a).to(dev) for a in (d, o, h))
    sets.append((td, to, th, pkg.voxelize(td, to, th)))
torch.cuda.synchronize()


def run(ks_in, ks_out=None, launches=240):
    ks_out = ks_in if ks_out is None else ks_out
    def go(i):
        td, to, th, _ = sets[ks_in[i % len(ks_in)]]
        pkg.voxelize(td, to, th, out=sets[ks_out[i % len(ks_out)]][3])
    for i in range(40):
        go(i)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for i in range(launches):
        go(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / launches * 1e3


allk = list(range(K))
for rep in range(3):
    same = [run([k]) for k in range(min(K, 3))]
    print(f"rep {rep}: the same batch over and over: " + " / ".join(f"{x:.1f}" for x in same) + f" us per launch;  "
          f"{K} batches and output buffers in rotation: {run(allk):.1f};  inputs in rotation, one output buffer: {run(allk, [0]):.1f};  "
          f"one input batch, outputs in rotation: {run([0], allk):.1f}", flush=True)
