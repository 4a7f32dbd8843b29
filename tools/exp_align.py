#!/usr/bin/env python3
"""Does the placement of the caller's buffers matter?  Same library, same batch; the output volume (and, in a
second sweep, the depth buffer) is moved by a byte offset inside one big allocation.  Offsets are interleaved
round by round so that drift cancels.  GPU box:  PROF_KIND=full|crop python tools/exp_align.py
"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
L = pkg._lib.load()
dev = torch.device("cuda:0")
kind = os.environ.get("PROF_KIND", "full")
n, R = int(os.environ.get("PROF_N", "1024")), 32
rounds, K = int(os.environ.get("AB_BLOCKS", "12")), int(os.environ.get("AB_LAUNCHES", "30"))
depth, off, hdr = synth.synth_batch(n, kind, seed0=0)
to, th = torch.from_numpy(off).to(dev), torch.from_numpy(hdr).to(dev)
PAD = 8 << 20
vol_bytes = n * 3 * R ** 3 * 4
raw_out = torch.empty(vol_bytes + PAD, dtype=torch.uint8, device=dev)
raw_in = torch.empty(depth.nbytes + PAD, dtype=torch.uint8, device=dev)
ml = torch.empty(n, dtype=torch.float32, device=dev)
mp = torch.empty((n, 3), dtype=torch.float32, device=dev)
st = torch.empty(n, dtype=torch.int32, device=dev)
base_out = raw_out.data_ptr()
base_in = raw_in.data_ptr()
# start from a 2 MiB boundary inside the allocation so that the offsets below mean the same thing in every process
a_out = (-base_out) % (2 << 20)
a_in = (-base_in) % (2 << 20)
print(f"{kind} n={n}: raw out base % 2MiB = {base_out % (2 << 20)}, raw in base % 2MiB = {base_in % (2 << 20)}")
stream = torch.cuda.current_stream().cuda_stream
host_depth = torch.from_numpy(depth)


def run(off_out, off_in):
    d_in = raw_in[a_in + off_in: a_in + off_in + depth.nbytes].view(torch.float32)
    d_in.copy_(host_depth)
    p_out = base_out + a_out + off_out
    torch.cuda.synchronize()

    def launch():
        rc = L.tsdf_voxelize_hip(d_in.data_ptr(), d_in.numel(), to.data_ptr(), th.data_ptr(), n, R, None, 0, stream,
                                 p_out, ml.data_ptr(), mp.data_ptr(), st.data_ptr())
        assert rc == 0
    for _ in range(3):
        launch()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(K):
        launch()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / K * 1e3


offs = [0, 16, 256, 512, 1024, 2048, 4096, 8192, 65536, 131072, 131072 + 1024, 393216, 1 << 20]
for label, mk in (("output volume", lambda o: (o, 0)), ("depth buffer", lambda o: (0, o))):
    t = {o: [] for o in offs}
    for r in range(rounds):
        for o in (offs if r % 2 == 0 else offs[::-1]):
            t[o].append(run(*mk(o)))
    base = np.median(t[0])
    print(f"-- offset of the {label} from a 2 MiB boundary --")
    for o in offs:
        v = np.array(t[o])
        print(f"   +{o:8d} B: median {np.median(v):7.2f} us  ({(np.median(v) / base - 1) * 100:+.2f} % vs +0)  min {v.min():7.2f}")
