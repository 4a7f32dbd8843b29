#!/usr/bin/env python3
"""Same library, same batch, several separately allocated output volumes (and depth copies): how much does the
buffer alone change the launch time?  GPU box:  PROF_KIND=crop python tools/exp_buffers.py"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
L = pkg._lib.load()
dev = torch.device("cuda:0")
kind = os.environ.get("PROF_KIND", "crop")
n, R = int(os.environ.get("PROF_N", "1024")), 32
NB = int(os.environ.get("NBUF", "6"))
rounds, K = 10, 30
depth, off, hdr = synth.synth_batch(n, kind, seed0=0)
to, th = torch.from_numpy(off).to(dev), torch.from_numpy(hdr).to(dev)
ins = [torch.from_numpy(depth).to(dev) for _ in range(NB)]
outs = [torch.empty((n, 3, R, R, R), dtype=torch.float32, device=dev) for _ in range(NB)]
ml = torch.empty(n, dtype=torch.float32, device=dev)
mp = torch.empty((n, 3), dtype=torch.float32, device=dev)
st = torch.empty(n, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
print(kind, "n =", n)
for i in range(NB):
    print(f"  buffer {i}: out @ 0x{outs[i].data_ptr():x} (% 2MiB = {outs[i].data_ptr() % (2 << 20)}), in @ 0x{ins[i].data_ptr():x}")


def run(i_in, i_out):
    def launch():
        rc = L.tsdf_voxelize_hip(ins[i_in].data_ptr(), ins[i_in].numel(), to.data_ptr(), th.data_ptr(), n, R, None, 0,
                                 stream, outs[i_out].data_ptr(), ml.data_ptr(), mp.data_ptr(), st.data_ptr())
        assert rc == 0
    for _ in range(2):
        launch()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(K):
        launch()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / K * 1e3


for label, pairs in (("output buffer (input 0)", [(0, j) for j in range(NB)]), ("input buffer (output 0)", [(j, 0) for j in range(NB)])):
    t = {p: [] for p in pairs}
    for r in range(rounds):
        for p in (pairs if r % 2 == 0 else pairs[::-1]):
            t[p].append(run(*p))
    print("--", label)
    for p in pairs:
        v = np.array(t[p])
        print(f"   in {p[0]} out {p[1]}: median {np.median(v):7.2f} us  min {v.min():7.2f}")
