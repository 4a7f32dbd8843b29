#!/usr/bin/env python3
"""Paired A/B of two builds of libtsdf_hip.so in ONE process (GPU box): the builds alternate block by block on
the same buffers, so box-to-box and minute-to-minute drift cancels and a 1 % difference is visible.

    python tools/ab_precise.py libtsdf_hip_r02.so libtsdf_hip.so   (bare names: looked up in build/, then the package)
    env: PROF_KIND=full|crop  PROF_N=1024  PROF_R=32  AB_BLOCKS=30  AB_LAUNCHES=40
         AB_AUG=1: the augmented entry (tsdf_voxelize_aug_hip) with reference-distribution maps instead of the plain one;
         AB_TOL=x: the two builds may differ by x (default 0: they must agree bit for bit)
         AB_ROTATE=k: while timed, every launch takes the next of k copies of the input (at distinct addresses) and of k
                      output buffer sets — no launch finds its input in the Infinity Cache (bench.py's headline regime)
"""
import ctypes, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
PKG = os.path.join(ROOT, "handposeestimation-with-3d-cnns_amd")
kind = os.environ.get("PROF_KIND", "full")
n = int(os.environ.get("PROF_N", "1024"))
R = int(os.environ.get("PROF_R", "32"))
blocks = int(os.environ.get("AB_BLOCKS", "30"))
K = int(os.environ.get("AB_LAUNCHES", "40"))
dev = torch.device("cuda:0")


def load(name):
    if not os.path.isabs(name):
        cand = [os.path.join(ROOT, "build", name), os.path.join(PKG, name)]
        name = next((c for c in cand if os.path.exists(c)), cand[-1])
    L = ctypes.CDLL(name)
    vp = ctypes.c_void_p
    L.tsdf_voxelize_hip.restype = ctypes.c_int
    L.tsdf_voxelize_hip.argtypes = [vp, ctypes.c_int64, vp, vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp,
                                    vp, vp, vp, vp]
    L.tsdf_voxelize_aug_hip.restype = ctypes.c_int
    L.tsdf_voxelize_aug_hip.argtypes = [vp, ctypes.c_int64, vp, vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp,
                                        vp, vp, vp, vp, vp]
    return L


libs = [load(a) for a in sys.argv[1:3]]
depth, off, hdr = synth.synth_batch(min(n, 1024), kind, seed0=0)
if n > 1024:
    reps = (n + 1023) // 1024
    depth = np.tile(depth, reps); hdr = np.tile(hdr, (reps, 1))[:n]
    off = np.concatenate([[0], np.cumsum(np.tile(np.diff(off), reps))]).astype(np.int64)[: n + 1]
td, to, th = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (depth, off, hdr))
outs = []
for L in libs:
    t = torch.empty((n, 3, R, R, R), dtype=torch.float32, device=dev)
    ml = torch.empty(n, dtype=torch.float32, device=dev)
    mp = torch.empty((n, 3), dtype=torch.float32, device=dev)
    st = torch.empty(n, dtype=torch.int32, device=dev)
    outs.append((t, ml, mp, st))
stream = torch.cuda.current_stream().cuda_stream


SWAP = os.environ.get("AB_SWAP_OUTS") == "1"   # library i writes into the other library's buffers
SAME = os.environ.get("AB_SAME_OUT") == "1"    # after the equality check both libraries write into ONE buffer: where a
                                               # buffer lives is worth up to 4 % at 64^3 and 20 % at 128^3 (exp_out_buffers.py)
AUG = os.environ.get("AB_AUG") == "1"
TOL = float(os.environ.get("AB_TOL", "0"))
txf = None
if AUG:
    aug = importlib.import_module("handposeestimation-with-3d-cnns_amd.augment")
    t, ml, mp, st = outs[0]
    assert libs[0].tsdf_voxelize_hip(td.data_ptr(), td.numel(), to.data_ptr(), th.data_ptr(), n, R, None, 0, stream,
                                     t.data_ptr(), ml.data_ptr(), mp.data_ptr(), st.data_ptr()) == 0
    torch.cuda.synchronize()
    txf = torch.from_numpy(aug.random_affines(mp.cpu().numpy(), rng=np.random.RandomState(2026))[0]).to(dev)


timing = False
ROT = max(1, int(os.environ.get("AB_ROTATE", "1")))
rot_in = [td] + [td.clone() for _ in range(ROT - 1)]
rot_out = [outs[0]] + [tuple(torch.empty_like(x) for x in outs[0]) for _ in range(ROT - 1)]   # shared by both libraries
turn = [0]


def launch(i):
    global td
    t, ml, mp, st = outs[0] if (SAME and timing) else outs[1 - i if SWAP else i]
    if timing and ROT > 1:
        k = turn[0] % ROT
        turn[0] += 1
        td = rot_in[k]
        t, ml, mp, st = rot_out[k]
    if AUG:
        rc = libs[i].tsdf_voxelize_aug_hip(td.data_ptr(), td.numel(), to.data_ptr(), th.data_ptr(), n, R, None, 0, stream,
                                           txf.data_ptr(), t.data_ptr(), ml.data_ptr(), mp.data_ptr(), st.data_ptr())
    else:
        rc = libs[i].tsdf_voxelize_hip(td.data_ptr(), td.numel(), to.data_ptr(), th.data_ptr(), n, R, None, 0, stream,
                                       t.data_ptr(), ml.data_ptr(), mp.data_ptr(), st.data_ptr())
    assert rc == 0, rc


for i in (0, 1):
    for _ in range(5):
        launch(i)
torch.cuda.synchronize()
if not os.environ.get("AB_NOCHECK"):
    if TOL > 0:
        worst = 0.0
        for a in range(0, n, 64):
            worst = max(worst, float((outs[0][0][a:a + 64] - outs[1][0][a:a + 64]).abs().max()))
        print(f"max |A - B| = {worst:.3g} (allowed {TOL})")
        assert worst <= TOL, "the two builds disagree"
    else:
        assert torch.equal(outs[0][0], outs[1][0]), "the two builds disagree"
if SWAP:
    print("(output buffers swapped between the two libraries)")
if SAME:
    print("(both libraries write into the same output buffer while timed)")
timing = True
times = [[], []]
for b in range(blocks):
    for i in ((0, 1) if b % 2 == 0 else (1, 0)):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        launch(i)  # one untimed launch after the switch
        a.record()
        for _ in range(K):
            launch(i)
        e.record()
        torch.cuda.synchronize()
        times[i].append(a.elapsed_time(e) / K * 1e3)
ta, tb = np.array(times[0]), np.array(times[1])
d = (tb - ta) / ta * 100.0
print(f"{kind} n={n} R={R}: A={sys.argv[1]} {np.median(ta):.2f} us (min {ta.min():.2f})   "
      f"B={sys.argv[2]} {np.median(tb):.2f} us (min {tb.min():.2f})")
print(f"   paired difference B vs A: median {np.median(d):+.2f} %  mean {d.mean():+.2f} % +- {d.std(ddof=1) / np.sqrt(len(d)):.2f} (s.e.), "
      f"{int((d < 0).sum())}/{len(d)} blocks faster")
