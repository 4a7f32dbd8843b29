#!/usr/bin/env python3
"""How many warm-up launches does a 64^3 measurement need?  The augmented kernel is co-bound by instruction issue, so it
sees the engine clock; the plain one is bound by its stores.  Times k launches after w warm-up launches, each case after
a host-side pause (the GPU idle, as in bench.py's extras between two workloads).  GPU box:  python tools/exp_warmup.py"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
dev = torch.device("cuda:0")
depth, off, hdr = synth.synth_batch(1024, "full", seed0=0)
td, to, th = (torch.from_numpy(a).to(dev) for a in (depth, off, hdr))
mid = pkg.voxelize(td, to, th).mid_p.cpu().numpy()
xf = torch.from_numpy(pkg.augment.random_affines(mid, rng=np.random.RandomState(2026))[0]).to(dev)
out_a = pkg.voxelize_aug(td, to, th, xf, res=64)
out_p = pkg.voxelize(td, to, th, res=64)


def run(fn, w, k, pause):
    torch.cuda.synchronize()
    time.sleep(pause)
    for _ in range(w):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(k):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / k * 1e3


fa = lambda: pkg.voxelize_aug(td, to, th, xf, res=64, out=out_a)
fp = lambda: pkg.voxelize(td, to, th, res=64, out=out_p)
for name, fn in (("augmented 64^3", fa), ("plain 64^3", fp)):
    for pause in (0.0, 1.0):
        for w, k in ((3, 10), (30, 10), (100, 10), (30, 40)):
            print(f"{name:15s} pause {pause:3.1f} s  warm {w:3d}  timed {k:3d}: {run(fn, w, k, pause):7.1f} us", flush=True)
# per-launch profile of a cold start
for name, fn in (("augmented 64^3", fa), ("plain 64^3", fp)):
    torch.cuda.synchronize(); time.sleep(1.0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
    ev[0].record()
    for i in range(60):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    t = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(60)]
    print(name, "launch by launch after 1 s idle:", " ".join(f"{x:.0f}" for x in t))

# ---- is the transient the engine clock?  A one-wave probe kernel after every launch reads shader-clock cycles per 100 MHz
# tick (tools/probes/clk_probe.hip); s_memtime counts at a constant rate on some parts — then the ratio is flat and says so.
import ctypes
so = os.path.join(ROOT, "tools", "probes", "libclk_probe.so")
if os.path.exists(so):
    C = ctypes.CDLL(so)
    C.clk_probe_launch.restype = ctypes.c_int
    C.clk_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    N = 60
    buf = torch.zeros((N, 4), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for name, fn in (("augmented 64^3", fa), ("plain 64^3", fp)):
        torch.cuda.synchronize(); time.sleep(1.0)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * N)]
        for i in range(N):
            ev[2 * i].record(); fn(); ev[2 * i + 1].record()
            C.clk_probe_launch(stream, buf[i].data_ptr(), 20000)
        torch.cuda.synchronize()
        t = [ev[2 * i].elapsed_time(ev[2 * i + 1]) * 1e3 for i in range(N)]
        b = buf.cpu().numpy()
        mhz = b[:, 0] / np.maximum(b[:, 1], 1) * 100.0
        print(name, "launch us :", " ".join(f"{x:.0f}" for x in t))
        print(name, "probe MHz :", " ".join(f"{x:.0f}" for x in mhz))
