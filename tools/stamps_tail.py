#!/usr/bin/env python3
"""Who makes the tail?  Frames per group and group end times of ONE launch, from the in-kernel stamps
(diagnostic library: make -C handposeestimation-with-3d-cnns_amd/csrc stamps).  Run on the GPU box:

    PROF_FRAMES=1024 PROF_KIND=full python tools/stamps_tail.py
    PROF_ROTATE=6: every launch takes the next of 6 copies of the input and of 6 output sets (bench.py's cache-cold regime)
    STAMPS_LIB=build/libtsdf_hip_x.so: another stamps build (tools/devbuild.sh x WORK -DTSDF_STAMPS -DTSDF_DEV_ONLY32)

Stamps are never cleared, so entries older than this launch's first start stamp are ignored.
"""
import ctypes, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TSDF_HIP_LIB", os.environ.get("STAMPS_LIB") or os.path.join(ROOT, "build", "libtsdf_hip_stamps.so"))
os.environ.setdefault("TSDF_ALLOW_LIB_OVERRIDE", "1")
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
L = pkg._lib.load()
dev = torch.device("cuda:0")
N = int(os.environ.get("PROF_FRAMES", "1024"))
kind = os.environ.get("PROF_KIND", "full")
G = int(os.environ.get("TSDF_GROUPS", "2"))
depth, off, hdr = synth.synth_batch(N, kind, seed0=0)
td, to, th = (torch.from_numpy(a).to(dev) for a in (depth, off, hdr))
out = pkg.voxelize(td, to, th)
ROT = max(1, int(os.environ.get("PROF_ROTATE", "1")))
sets = [(td, out)] + [(td.clone(), pkg.voxelize(td, to, th)) for _ in range(ROT - 1)]
turn = 0
SL, FR, BL = 16, 8, 512
L.tsdf_debug_read_stamps.restype = ctypes.c_int
L.tsdf_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
for rep in range(3):
    for _ in range(3 if ROT == 1 else 2 * ROT + 1):
        d_, o_ = sets[turn % ROT]
        turn += 1
        pkg.voxelize(d_, to, th, out=o_)
    torch.cuda.synchronize()
    buf = np.zeros(BL * FR * SL, np.uint64)
    assert L.tsdf_debug_read_stamps(buf.ctypes.data, buf.size) == buf.size
    s = buf.reshape(BL, FR, SL).astype(np.int64)[:256]
    t0 = s[:, :G, 0].max(axis=None)  # latest first-frame start: everything of this launch is >= the earliest
    t0 = s[:, :G, 0][s[:, :G, 0] > t0 - 100000].min()
    live = s[:, :, 0] >= t0                       # [block][iter*G+group]
    cnt = np.zeros((256, G), int)
    end = np.zeros((256, G), np.int64)
    for g in range(G):
        lv = live[:, g::G]
        cnt[:, g] = lv.sum(axis=1)
        e = np.where(lv, s[:, g::G, 9], 0)
        end[:, g] = e.max(axis=1) - t0
    endus = end / 100.0
    print(f"rep {rep}: {N} {kind} frames; frames/group histogram: "
          + ", ".join(f"{k}:{int((cnt == k).sum())}" for k in range(cnt.max() + 1))
          + f"  (stamps keep {FR // G} frames per group)")
    cu_end = endus.max(axis=1)
    print("   mean CU end per XCD (blockIdx % 8): " + " ".join(f"{cu_end[x::8].mean():7.1f}" for x in range(8))
          + f"   (all CUs: mean {cu_end.mean():.1f}, max {cu_end.max():.1f})")
    q = np.percentile(endus, [0, 10, 50, 90, 99, 100])
    print("   group end (us) min/p10/p50/p90/p99/max: " + " ".join(f"{x:7.2f}" for x in q))
    hv = s[:, :, 10]
    print(f"   frames voxelized alone: {int(((hv == 1) & live).sum())}, with the other group's help: {int(((hv == 2) & live).sum())}")
    for it in range(FR // G):
        for g in range(G):
            m = live[:, it * G + g] & (hv[:, it * G + g] == 2)
            if m.any():
                d = (s[:, it * G + g, 9] - s[:, it * G + g, 8])[m] / 100.0
                print(f"      helped phase 2, frame #{it} group {g}: n={int(m.sum())} median {np.median(d):.2f} us")
            m = live[:, it * G + g] & (hv[:, it * G + g] == 1)
            if m.any():
                d = (s[:, it * G + g, 9] - s[:, it * G + g, 8])[m] / 100.0
                print(f"      alone  phase 2, frame #{it} group {g}: n={int(m.sum())} median {np.median(d):.2f} us")
    for k in range(1, cnt.max() + 1):
        m = cnt == k
        if m.any():
            print(f"   groups with {k} frames: end median {np.median(endus[m]):7.2f}  max {endus[m].max():7.2f}")
    # per-frame durations by ordinal
    for it in range(FR // G):
        d = []
        for g in range(G):
            lv = live[:, it * G + g]
            d.append((s[:, it * G + g, 9] - s[:, it * G + g, 0])[lv])
        d = np.concatenate(d) / 100.0
        if d.size:
            p1 = []
            lockw = []
            p2 = []
            for g in range(G):
                lv = live[:, it * G + g]
                p1.append((s[:, it * G + g, 3] - s[:, it * G + g, 0])[lv])
                lockw.append((s[:, it * G + g, 4] - s[:, it * G + g, 3])[lv])
                p2.append((s[:, it * G + g, 9] - s[:, it * G + g, 4])[lv])
            p1, lockw, p2 = (np.concatenate(x) / 100.0 for x in (p1, lockw, p2))
            print(f"   frame #{it}: n={d.size:4d} total median {np.median(d):6.2f} (p90 {np.percentile(d,90):6.2f})  "
                  f"stream {np.median(p1):6.2f}  barrier+lock {np.median(lockw):6.2f}  tables+stage+phase2 {np.median(p2):6.2f}")
