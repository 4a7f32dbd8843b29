#!/usr/bin/env python3
"""Randomised parity run, far beyond what the test-suite holds (GPU box; the oracle runs on the host cores):
random geometry (widths 1..645, heights 1..240, bboxes anywhere, sparse..dense, blobs and scatter, negative / mixed-sign /
sub-threshold depths, NaN sprinkles), random camera constants, both layouts, R in {16,32,40,64}, fused (n > 128) and split (n <= 128)
kernels, the augmented entry with reference-distribution maps, labels.  Checks: status / max_l / mid_p / labels bit exact,
volume <= 1e-5, and — on half of the plain contiguous rounds, through the debug build — the exact pixel map of up to 24 frames.      python tools/fuzz_parity.py [rounds=40] [seed=1]"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
import oracle
dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
TOL = 1e-5
tot = dict(frames=0, voxels=0, ok_frames=0, max_err=0.0, bad=0)
t_start = time.time()

def make_frames(n):
    frames = []
    for k in range(n):
        bw = int(rng.choice([rng.integers(1, 9), rng.integers(9, 200), rng.integers(200, 330), rng.integers(330, 646)], p=[.1, .6, .25, .05]))
        bh = int(rng.integers(1, 241)) if bw * 240 < 70000 else int(rng.integers(1, max(2, 70000 // bw)))
        left, top = int(rng.integers(-60, 420)), int(rng.integers(-60, 320))
        base = float(rng.uniform(120, 2000))
        d = (base + rng.normal(0, base * rng.uniform(0.0, 0.08), (bh, bw))).astype(np.float32)
        keep = rng.random((bh, bw)) < rng.choice([0.01, 0.2, 0.8, 1.0])
        if rng.random() < 0.7:
            yy, xx = np.mgrid[0:bh, 0:bw]
            cy, cx = rng.uniform(0, bh), rng.uniform(0, bw)
            ry, rx = rng.uniform(1, bh + 1), rng.uniform(1, bw + 1)
            keep &= ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        d[~keep] = 0.0
        r = rng.random()
        if r < 0.08: d *= -1.0
        elif r < 0.13:   # mixed signs: the grid straddles the camera plane z = 0 (q = -F / v_z changes sign inside it)
            flip = (np.arange(bw)[None, :] < rng.uniform(0, bw)) if rng.random() < 0.5 else (rng.random((bh, bw)) < 0.5)
            d = np.where(flip, -d * np.float32(rng.choice([1.0, 0.3, 2.5])), d).astype(np.float32)
        elif r < 0.18: d[rng.random((bh, bw)) < 0.1] = 0.75
        elif r < 0.22: d[rng.random((bh, bw)) < 0.05] = np.nan
        frames.append((np.array([640, 480, left, top, left + bw, top + bh], np.int32), d.reshape(-1)))
    headers = np.stack([f[0] for f in frames])
    offsets = np.zeros(n + 1, np.int64); offsets[1:] = np.cumsum([f[1].size for f in frames])
    return np.concatenate([f[1] for f in frames]), offsets, headers

for rnd in range(rounds):
    n = int(rng.choice([1, 3, 17, 100, 128, 129, 300, 700]))
    R = int(rng.choice([16, 32, 32, 32, 40, 64] if rng.random() < 0.85 else [4, 8, 12, 20, 24, 28, 48, 96, 128]))
    if R >= 64: n = min(n, 300 if R == 64 else 17)
    layout = str(rng.choice(["czyx", "cxyz"]))
    cam = ocam = None
    if rng.random() < 0.3:   # other camera constants: focal, principal point, invalid-depth threshold, truncation (voxels)
        ocam = (float(rng.uniform(100, 700)), float(rng.uniform(60, 400)), float(rng.uniform(40, 300)),
                float(rng.choice([0.5, 1.0, 2.0, 150.0])), float(rng.choice([1.0, 2.5, 3.0, 6.0])))
        cam = pkg.TsdfCam(*ocam)
    depth, off, hdr = make_frames(n)
    gt = rng.normal(0, 300, (n, 63)).astype(np.float32)
    td, to, th, tg = (torch.from_numpy(a).to(dev) for a in (depth, off, hdr, gt))
    aug = rng.random() < 0.35
    indexed = rng.random() < 0.3      # the batch is drawn by index (shuffled, with repeats) from the frames as a resident pack
    if indexed:
        pick = rng.integers(0, n, n).astype(np.int64)
        pack = (td, to, th, tg)
        lens = np.diff(off)[pick]
        depth = np.concatenate([depth[off[i]:off[i + 1]] for i in pick]) if n else depth
        off = np.zeros(n + 1, np.int64); off[1:] = np.cumsum(lens)
        hdr, gt = hdr[pick], gt[pick]
        # small plain batches: sometimes hand the index over in ordinary host memory (by value, in the kernel arguments)
        tpick = torch.from_numpy(pick.copy()) if (n <= 32 and not aug and rng.random() < 0.6) else torch.from_numpy(pick).to(dev)
    with np.errstate(all="ignore"):
        if aug:
            mid = oracle.voxelize(depth, off, hdr, R=R, n_threads=16, cam=ocam)["mid_p"]
            xf = pkg.augment.random_affines(mid, rng=int(rng.integers(1 << 30)))[0]
            if indexed:
                got, g_nor, g_aug = pkg.voxelize_indexed(*pack[:3], tpick, pack[3], res=R, layout=layout, cam=cam,
                                                         xforms=torch.from_numpy(xf).to(dev), gt_copy=True)
            else:
                got, g_nor, g_aug = pkg.voxelize_aug(td, to, th, torch.from_numpy(xf).to(dev), res=R, layout=layout, gt=tg, cam=cam)
            ref = oracle.voxelize_aug(depth, off, hdr, xf, R=R, layout=0 if layout == "czyx" else 1, n_threads=16, cam=ocam)
            r_aug = oracle.transform_joints(gt, xf) if hasattr(oracle, "transform_joints") else None
            r_nor = oracle.normalize_joints(r_aug if r_aug is not None else gt, ref["max_l"], ref["mid_p"])
        else:
            if indexed:
                got, g_nor = pkg.voxelize_indexed(*pack[:3], tpick, pack[3], res=R, layout=layout, cam=cam)
            else:
                got, g_nor = pkg.voxelize_labels(td, to, th, tg, res=R, layout=layout, cam=cam)
            ref = oracle.voxelize(depth, off, hdr, R=R, layout=0 if layout == "czyx" else 1, n_threads=16, cam=ocam)
            r_nor = oracle.normalize_joints(gt, ref["max_l"], ref["mid_p"])
    torch.cuda.synchronize()
    st = got.status.cpu().numpy()
    bad = 0
    bad += int((st != ref["status"]).sum())
    bad += int((got.max_l.cpu().numpy().view(np.uint32) != ref["max_l"].view(np.uint32)).sum())
    bad += int((got.mid_p.cpu().numpy().view(np.uint32) != ref["mid_p"].view(np.uint32)).any(axis=1).sum())
    okf = st == 0
    gn = g_nor.cpu().numpy()
    if not aug or r_aug is not None:
        bad += int((gn[okf].view(np.uint32) != r_nor[okf].view(np.uint32)).any(axis=1).sum())
    err = np.abs(got.tsdf.cpu().numpy() - ref["tsdf"]).reshape(n, -1).max(axis=1)
    bad += int((err > TOL).sum())
    # Pixel maps, EXACT (plain, contiguous batches, any camera; through the debug build's tsdf_debug_pixmap_hip): the
    # pixel every voxel gathers — or why it gathers none — against the oracle's, for up to 24 frames of the batch
    pm_checked = 0
    if not aug and not indexed and R <= 64 and n <= 300 and rng.random() < 0.5:
        ex = oracle.voxelize(depth, off, hdr, R=R, n_threads=16, extras=True, want_tsdf=False, cam=ocam)
        _, pm, pst = pkg.voxel_pixels(td, to, th, res=R, layout=layout, cam=cam)
        torch.cuda.synchronize()
        pm = pm.cpu().numpy()
        for i in (range(n) if n <= 24 else rng.choice(n, 24, replace=False)):
            if ref["status"][i] != 0:
                continue
            with np.errstate(all="ignore"):
                _, want = oracle.voxels(depth[off[i]:off[i + 1]], hdr[i], ex["ori"][i], ex["grid"][i, 4], ex["grid"][i, 5], R=R,
                                        want_pixmap=True, cam=ocam)
            bad += int(not np.array_equal(pm[i].reshape(-1), np.asarray(want).reshape(-1)))
            pm_checked += 1
        tot["pixmaps"] = tot.get("pixmaps", 0) + pm_checked
    tot["frames"] += n; tot["voxels"] += n * 3 * R ** 3; tot["ok_frames"] += int(okf.sum()); tot["bad"] += bad
    tot["max_err"] = max(tot["max_err"], float(err.max()))
    print(f"round {rnd:3d}: n={n:4d} R={R:3d} {layout} {'aug' if aug else 'plain'}{' cam' if cam is not None else ''}{(' indexed(by value)' if not tpick.is_cuda else ' indexed') if indexed else ''}  ok frames {int(okf.sum()):4d}  max err {err.max():.2e}  mismatches {bad}{f'  pixmaps {pm_checked}' if pm_checked else ''}", flush=True)
    if bad:
        print("   first bad frames:", np.flatnonzero(err > TOL)[:5], "seed state differs: rerun with the same arguments to reproduce")
print(tot, f"{time.time() - t_start:.0f} s")
sys.exit(1 if tot["bad"] else 0)
