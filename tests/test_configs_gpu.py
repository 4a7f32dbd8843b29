"""GPU tier: BASELINE.json configs[2], [3] (its one-GPU form) and [4] at their stated sizes — oracle parity on a
sample, size-independent properties on everything, shard identity at the 8 rank boundaries.  The rates the runs
achieve are printed (pytest -s) and measured properly by bench.py's `extras`.

Inputs are the seeded synthetic MSRA-like crops / full frames of synth.py (the reference ships no data; MSRA is not
redistributable and there is no network): 2,048 distinct crops repeated to the stated counts.
"""
import time

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
TOL = 1e-5
N_P0 = 8500      # BASELINE configs[2]: one MSRA subject
N_ALL = 76500    # BASELINE configs[3]: all nine subjects


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def crops(synth):
    return [synth.synth_frame(100000 + i, "crop") for i in range(2048)]


def tiled_pack(pkg, crops, n):
    P = pkg.packing
    base = P.pack_frames(crops)
    reps = (n + len(crops) - 1) // len(crops)
    lens = np.tile(np.diff(base.offsets), reps)[:n]
    off = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    depth = np.tile(base.depth, reps)[: off[-1]]
    return P.PackedFrames(np.ascontiguousarray(depth), off, np.ascontiguousarray(np.tile(base.headers, (reps, 1))[:n]))


def check_properties(t, status, max_l):
    """Value range, shared zero mask and sign across the three channels, |.| <= 1 off the snapped voxels."""
    assert bool((status == 0).all()) and bool((max_l > 0).all())
    assert float(t.abs().max()) <= 1.0
    zero = t == 0
    assert bool((zero[:, 0] == zero[:, 1]).all()) and bool((zero[:, 0] == zero[:, 2]).all())
    neg = t < 0
    nz = ~zero[:, 0]
    assert bool((neg[:, 0] == neg[:, 1])[nz].all()) and bool((neg[:, 0] == neg[:, 2])[nz].all())
    far = (t.abs() == 1).all(dim=1)
    norm2 = (t.double() ** 2).sum(dim=1)
    assert bool((norm2[~far & nz] <= 1.0 + 1e-6).all())
    assert bool(nz.flatten(1).any(dim=1).all())         # every frame has occupied voxels


def test_config2_subject_streamed_through_the_loader(pkg, crops, tmp_path):
    """configs[2]: ~8.5 k frames of one subject, streamed from a memory-mapped pack through VoxelLoader (worker
    thread -> reusable pinned staging sets -> H2D on a copy stream -> fused voxelizer + labels)."""
    pk = tiled_pack(pkg, crops, N_P0)
    rng = np.random.default_rng(0)
    pk.gt = rng.normal(0, 60, (N_P0, 63)).astype(np.float32)
    pk.gt[:, 2::3] -= 450.0
    pk.group_start, pk.group_names = np.array([0, N_P0]), ["all"]
    pk.save(str(tmp_path / "P0.tsdfpk"))
    ds = pkg.MSRADepthDataset(None, train=False, test_idx=0, subjects=["P0"], packed_dir=str(tmp_path), build_packs=False)
    assert len(ds) == N_P0 and ds.packed
    B = 1024
    loader = pkg.VoxelLoader(ds, batch_size=B, device=dev(), max_pixels=B * 160 * 160)
    sample = rng.choice(N_P0, 40, replace=False)
    rates = []
    for epoch in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        seen = 0
        kept = []
        for batch in loader:
            n = batch.tsdf.shape[0]
            if epoch == 0:
                check_properties(batch.tsdf, batch.status, batch.max_l)
                assert bool(((batch.gt_nor >= 0) & (batch.gt_nor <= 1)).all())
                for i in sample[(sample >= seen) & (sample < seen + n)]:
                    kept.append((int(i), batch.tsdf[i - seen].cpu().numpy(), float(batch.max_l[i - seen]),
                                 batch.gt_nor[i - seen].cpu().numpy(), batch.mid_p[i - seen].cpu().numpy()))
            seen += n
        torch.cuda.synchronize()
        rates.append(seen / (time.perf_counter() - t0))
        assert seen == N_P0
        if epoch == 0:
            assert len(kept) == len(sample)
            for i, t, ml, nor, mp in kept:
                h, d = pk.frame(i)
                ref = oracle.voxelize(d, np.array([0, d.size], np.int64), h[None])
                assert np.abs(t - ref["tsdf"][0]).max() <= TOL and ml == ref["max_l"][0]
                np.testing.assert_array_equal(nor, oracle.normalize_joints(pk.gt[i:i + 1], ref["max_l"], ref["mid_p"])[0])
    print(f"configs[2]: {N_P0} crops through VoxelLoader: {rates[0]:.0f} (first epoch, with checks), "
          f"{max(rates[1:]):.0f} crops/s (steady)")
    assert max(rates[1:]) > 2.0e5    # PCIe-bound; the floor only catches a broken pipeline (bench.py has the number)


def test_config2_subject_resident_and_shuffled(pkg, crops):
    """configs[2] the MI355X way: the subject's pack uploaded to HBM once (0.53 GB), SHUFFLED batches of 1024 drawn by
    index on the device (ResidentLoader / tsdf_voxelize_indexed_hip), labels from the same launch.  Every frame is seen
    exactly once per epoch, properties hold on all of them, a sample equals the oracle, and batches of the reference's
    size (16) run too."""
    pk = tiled_pack(pkg, crops, N_P0)
    rng = np.random.default_rng(1)
    pk.gt = rng.normal(0, 60, (N_P0, 63)).astype(np.float32)
    pk.gt[:, 2::3] -= 450.0
    ds = pkg.MSRADepthDataset.from_packs([pk])
    loader = pkg.ResidentLoader(ds, batch_size=1024, device=dev(), shuffle=True, seed=4)
    assert loader.resident_bytes() == 4 * pk.depth.size
    plan = pkg.dataset.plan_batches(N_P0, 1024, shuffle=True, seed=4, epoch=0, weights=ds.pixels())
    sample = set(rng.choice(N_P0, 40, replace=False).tolist())
    rates = []
    for epoch in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        seen = 0
        for k, batch in enumerate(loader):
            n = batch.tsdf.shape[0]
            if epoch == 0:
                check_properties(batch.tsdf, batch.status, batch.max_l)
                assert bool(((batch.gt_nor >= 0) & (batch.gt_nor <= 1)).all())
                ids = plan[k]
                np.testing.assert_array_equal(batch.gt.cpu().numpy(), pk.gt[ids])
                for j in [j for j, i in enumerate(ids) if int(i) in sample]:
                    h, d = pk.frame(int(ids[j]))
                    ref = oracle.voxelize(d, np.array([0, d.size], np.int64), h[None])
                    assert np.abs(batch.tsdf[j].cpu().numpy() - ref["tsdf"][0]).max() <= TOL
                    assert float(batch.max_l[j]) == ref["max_l"][0]
                    np.testing.assert_array_equal(batch.gt_nor[j].cpu().numpy(),
                                                  oracle.normalize_joints(pk.gt[ids[j]:ids[j] + 1], ref["max_l"], ref["mid_p"])[0])
                    sample.discard(int(ids[j]))
            seen += n
        torch.cuda.synchronize()
        rates.append(seen / (time.perf_counter() - t0))
        assert seen == N_P0
    assert not sample                                   # every sampled frame came by
    assert sorted(np.concatenate(plan).tolist()) == list(range(N_P0))
    small = pkg.ResidentLoader(ds, batch_size=16, device=dev(), shuffle=True, seed=4)
    n16 = sum(b.tsdf.shape[0] for b in small)
    torch.cuda.synchronize()
    assert n16 == N_P0 and len(small) == (N_P0 + 15) // 16
    print(f"configs[2] resident: {N_P0} crops, shuffled batches of 1024: {max(rates[1:]):.0f} crops/s")
    assert max(rates[1:]) > 1.0e6    # an order of magnitude above the link-bound loader; bench.py has the number


def test_config3_all_subjects_on_one_gpu(pkg, crops):
    """configs[3], one-GPU form: ~76.5 k crops resident, one launch (19 frames per group through the work queue);
    properties on every frame, oracle on a sample, and each of the 8 pixel-balanced rank shards voxelized alone is
    bit-identical to its slice of the whole (frame-sharding across 8 GPUs needs no exchange)."""
    d = dev()
    pk = tiled_pack(pkg, crops, N_ALL)
    td, to, th = pk.to_torch(d)
    out = pkg.voxelize(td, to, th)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pkg.voxelize(td, to, th, out=out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"configs[3] on one GPU: {N_ALL} crops in {dt * 1e3:.2f} ms = {N_ALL / dt / 1e6:.2f} M frames/s")
    for a in range(0, N_ALL, 4096):
        b = min(N_ALL, a + 4096)
        check_properties(out.tsdf[a:b], out.status[a:b], out.max_l[a:b])
    # periodic input -> periodic output
    assert torch.equal(out.tsdf[:2048], out.tsdf[2048 * 30:2048 * 31])
    idx = np.random.default_rng(1).choice(N_ALL, 48, replace=False)
    for i in idx:
        h, dd = pk.frame(int(i))
        ref = oracle.voxelize(dd, np.array([0, dd.size], np.int64), h[None])
        assert np.abs(out.tsdf[int(i)].cpu().numpy() - ref["tsdf"][0]).max() <= TOL
        assert float(out.max_l[int(i)]) == ref["max_l"][0]
    bounds = pkg.shard.shard_bounds(N_ALL, 8, weights=pk.pixels)
    assert bounds[0][0] == 0 and bounds[-1][1] == N_ALL
    px = [int(pk.offsets[b] - pk.offsets[a]) for a, b in bounds]
    assert max(px) / (sum(px) / 8) < 1.01
    for a, b in bounds:
        sub_off = (to[a:b + 1] - to[a]).contiguous()
        sub = pkg.voxelize(td[int(pk.offsets[a]):int(pk.offsets[b])], sub_off, th[a:b].contiguous())
        assert torch.equal(sub.tsdf, out.tsdf[a:b]) and torch.equal(sub.mid_p, out.mid_p[a:b])
        assert torch.equal(sub.max_l, out.max_l[a:b])
        del sub


def test_config4_64cubed_with_fused_augmentation(pkg, synth):
    """configs[4]: 1024 full 320x240 frames -> 64^3 with the 3-D augmentation fused into the kernel (re-specified,
    parity unpinned: the oracle restates the same contract); identity map == plain 64^3 path."""
    d = dev()
    n = 1024
    depth, off, hdr = synth.synth_batch(n, "full", seed0=0)
    td, to, th = (torch.from_numpy(a).to(d) for a in (depth, off, hdr))
    plain = pkg.voxelize(td, to, th, res=64)
    xf, prm = pkg.augment.random_affines(plain.mid_p.cpu().numpy(), rng=np.random.RandomState(2026))
    assert prm["stretch"].min() >= 2 / 3 and prm["stretch"].max() < 1.5 and set(prm["rot_xy"]) <= set(range(-30, 30))
    txf = torch.from_numpy(xf).to(d)
    out = pkg.voxelize_aug(td, to, th, txf, res=64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pkg.voxelize_aug(td, to, th, txf, res=64, out=out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"configs[4]: {n} full frames -> 64^3 augmented in {dt * 1e3:.3f} ms = {n / dt / 1e6:.3f} M frames/s")
    for a in range(0, n, 128):
        check_properties(out.tsdf[a:a + 128], out.status[a:a + 128], out.max_l[a:a + 128])
    idx = np.random.default_rng(4).choice(n, 12, replace=False)
    for i in idx:
        sl = slice(off[i], off[i + 1])
        ref = oracle.voxelize_aug(depth[sl], np.array([0, off[i + 1] - off[i]], np.int64), hdr[i][None], xf[i][None], R=64)
        assert float(out.max_l[i]) == ref["max_l"][0]
        np.testing.assert_array_equal(out.mid_p[i].cpu().numpy(), ref["mid_p"][0])
        assert np.abs(out.tsdf[i].cpu().numpy() - ref["tsdf"][0]).max() <= TOL
    ident = torch.from_numpy(pkg.augment.identity_affines(64)).to(d)
    a0 = pkg.voxelize_aug(td[: off[64]], to[:65].contiguous(), th[:64].contiguous(), ident, res=64)
    assert torch.equal(a0.max_l, plain.max_l[:64]) and float((a0.tsdf - plain.tsdf[:64]).abs().max()) <= TOL


def test_bench_one_rank_through_torchrun_prints_the_plain_line():
    """The N = 1 point of a scaling run is launched through torch.distributed.run (world 1, RCCL process group of one
    rank), the round's BENCH run as plain `python bench.py`: both on the real GPU, short, and the two lines must carry the
    same keys — `roofline` and `cpu_baseline` included — and agree on the rate (tests/test_bench_ranks.py does the same
    without a GPU)."""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    tail = [os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--no-extras", "--no-live-traffic"]
    env = {k: v for k, v in os.environ.items() if k not in ("TORCHELASTIC_RUN_ID", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    lines = []
    for cmd in ([sys.executable] + tail,
                [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                 "--master-port", str(port)] + tail):
        r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        js = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(js) == 1
        lines.append(json.loads(js[0]))
    plain, tr = lines
    assert plain["dist_backend"] is None and tr["dist_backend"] == "nccl" and tr["dist_world_size"] == 1
    for j in lines:
        assert j["n_gpus"] == 1 and "dry_run" not in j and "rehearsal" not in j
        assert j["roofline"]["rotation"] >= 6 and j["roofline"]["working_set_bytes"] >= 4e9
        assert 0.3 < j["roofline"]["frac"] <= j["roofline"]["frac_same_batch"] * 1.02 < 1.0
        assert j["cpu_baseline"]["value"] == max(leg["frames_per_s"] for leg in j["cpu_baseline"]["legs"])
        assert j["extras"]["configs[3]_sharded"]["per_rank_frames"] == [N_ALL]
    assert set(plain) == set(tr) and set(plain["roofline"]) == set(tr["roofline"]) and set(plain["cpu_baseline"]) == set(tr["cpu_baseline"])
    assert abs(plain["value"] - tr["value"]) / plain["value"] < 0.05
