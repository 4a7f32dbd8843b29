#!/usr/bin/env python3
"""BASELINE.json configs[2]: one MSRA subject's worth of frames (~8.5 k MSRA-like crops) streamed from
pinned host memory, H2D copies (hipMemcpyAsync on a copy stream) overlapped with the voxelizer.

Reports the PCIe-inclusive rate (what a host-fed pipeline gets) next to the device-resident rate
(what bench.py reports as `value`).  The boundary itself takes device pointers; this is a note for
DESIGN.md, never the headline number.
"""
import importlib, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
dev = torch.device("cuda:0")
N = int(os.environ.get("STREAM_FRAMES", "8500")); B = int(os.environ.get("STREAM_BATCH", "1024"))
kind = os.environ.get("STREAM_KIND", "crop")
base = [synth.synth_frame(i, kind) for i in range(1024)]
frames = [base[i % 1024] for i in range(N)]
batches = []
for i in range(0, N, B):
    pk = pkg.packing.pack_frames(frames[i:i + B])
    batches.append(tuple(torch.from_numpy(a).pin_memory() for a in (pk.depth, pk.offsets, pk.headers)))
nbytes_in = sum(b[0].numel() * 4 for b in batches)
copy = torch.cuda.Stream(device=dev); comp = torch.cuda.current_stream(dev)
# device buffers: two sets (double buffer), outputs per set
maxpx = max(b[0].numel() for b in batches)
dbuf = [(torch.empty(maxpx, dtype=torch.float32, device=dev), torch.empty(B + 1, dtype=torch.int64, device=dev),
         torch.empty((B, 6), dtype=torch.int32, device=dev)) for _ in range(2)]
def make_out(n):
    return pkg.TsdfBatch(torch.empty((n, 3, 32, 32, 32), dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev),
                         torch.empty((n, 3), dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.int32, device=dev))
out_sets = [make_out(B), make_out(B)]
def run_stream():
    ev_copied = [torch.cuda.Event(), torch.cuda.Event()]; ev_done = [torch.cuda.Event(), torch.cuda.Event()]
    for k, (hd, ho, hh) in enumerate(batches):
        s = k & 1; n = hh.shape[0]
        with torch.cuda.stream(copy):
            if k >= 2: copy.wait_event(ev_done[s])           # buffer s free again
            dbuf[s][0][:hd.numel()].copy_(hd, non_blocking=True)
            dbuf[s][1][:n + 1].copy_(ho, non_blocking=True)
            dbuf[s][2][:n].copy_(hh, non_blocking=True)
            ev_copied[s].record(copy)
        comp.wait_event(ev_copied[s])
        o = out_sets[s] if n == B else make_out(n)
        pkg.voxelize(dbuf[s][0][:hd.numel()], dbuf[s][1][:n + 1], dbuf[s][2][:n].contiguous(), out=o)
        ev_done[s].record(comp)
    torch.cuda.synchronize()
run_stream()
t0 = time.perf_counter(); reps = 3
for _ in range(reps): run_stream()
t_stream = (time.perf_counter() - t0) / reps
# device-resident: same batches already on the GPU
res = [tuple(t.to(dev) for t in b) for b in batches]
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    for k, (d, o, h) in enumerate(res):
        n = h.shape[0]
        pkg.voxelize(d, o, h, out=out_sets[k & 1] if n == B else make_out(n))
torch.cuda.synchronize()
t_res = (time.perf_counter() - t0) / reps
print(json.dumps({"frames": N, "kind": kind, "batch": B, "input_MB": round(nbytes_in / 1e6, 1),
                  "streamed_frames_per_s": round(N / t_stream, 1), "streamed_h2d_GBps": round(nbytes_in / t_stream / 1e9, 2),
                  "resident_frames_per_s": round(N / t_res, 1)}))
