#!/usr/bin/env python3
"""One bench step (16 launches over six batches in rotation) issued eagerly vs replayed from a captured hipGraph (GPU box).
Captured launches use CU-local work queues (no per-stream queue word: include/tsdf.h)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
dev = torch.device("cuda:0")
ROT, PER = 6, 16
sets = []
for k in range(ROT):
    d, o, h = synth.synth_batch(1024, "full", seed0=k * 1024, threads=8)
    t = tuple(torch.from_numpy(a).to(dev) for a in (d, o, h))
    sets.append(t + (pkg.voxelize(*t),))
turn = [0]
def step():
    for _ in range(PER):
        a = sets[turn[0] % ROT]; turn[0] += 1
        pkg.voxelize(a[0], a[1], a[2], out=a[3])
for _ in range(3): step()
torch.cuda.synchronize()
def timed(fn, k=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); a.record()
    for _ in range(k): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k / PER * 1e3
# a graph of 3 steps = 48 launches = 8 full turns of the rotation (so that a replay continues the rotation)
s = torch.cuda.Stream(dev)
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    turn[0] = 0
    g.capture_begin()
    for _ in range(3): step()
    g.capture_end()
torch.cuda.synchronize()
for rep in range(3):
    e = timed(step, 30)
    with torch.cuda.stream(s):
        gr = timed(g.replay, 10) / 3
    print(f"rep {rep}: eager {e:.2f} us per launch   graph replay {gr:.2f} us per launch")
