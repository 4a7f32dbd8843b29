import importlib, sys, time, numpy as np, torch
import torch.utils.data as tdata
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
dev = torch.device("cuda:0")
crops = [synth.synth_frame(100000 + i, "crop") for i in range(2048)]
base = pkg.packing.pack_frames(crops)
n = 8500
reps = (n + 2047) // 2048
lens = np.tile(np.diff(base.offsets), reps)[:n]
off = np.zeros(n + 1, np.int64); np.cumsum(lens, out=off[1:])
pk = pkg.packing.PackedFrames(np.ascontiguousarray(np.tile(base.depth, reps)[: off[-1]]), off, np.ascontiguousarray(np.tile(base.headers, (reps, 1))[:n]))
pk.gt = np.zeros((n, 63), np.float32)
ds = pkg.MSRADepthDataset.from_packs([pk])
mds = pkg.MSRA_Dataset.from_raw(ds, device=dev)
dl = tdata.DataLoader(mds, batch_size=16, shuffle=True)
for ep in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter(); seen = 0
    for b in dl: seen += b[0].shape[0]
    torch.cuda.synchronize(); print(f"epoch {ep}: {seen / (time.perf_counter() - t0) / 1e6:.3f} M crops/s")
# the dataset's own part, without torch's loader around it: the same shuffled batches straight into __getitems__
perm = torch.randperm(n).tolist()
batches = [perm[i:i + 16] for i in range(0, n - 15, 16)]
for ep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b in batches: mds.__getitems__(b)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"__getitems__ alone: host {len(batches) * 16 / (t1 - t0) / 1e6:.3f} M crops/s issued, {len(batches) * 16 / (t2 - t0) / 1e6:.3f} M crops/s completed "
          f"({(t1 - t0) / len(batches) * 1e6:.2f} us per batch on the host, {(t2 - t0) / len(batches) * 1e6:.2f} us per batch end to end)")
