"""CPU tier: host-side logic — MSRA .bin I/O and batch packing, sharding, the reference-signature
shims' host halves, and "fails loudly without the HIP path"."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_names


def test_read_write_bin_round_trip(pkg, synth, tmp_path):
    h, d = synth.synth_frame(3, "crop")
    p = tmp_path / "000000_depth.bin"
    pkg.packing.write_bin(str(p), h, d)
    assert os.path.getsize(p) == 24 + 4 * d.size  # 6 x int32 + float32 payload (read_MSRA.py:158-162)
    h2, d2 = pkg.packing.read_bin(str(p))
    assert h2.dtype == np.int32 and d2.dtype == np.float32
    np.testing.assert_array_equal(h, h2)
    np.testing.assert_array_equal(d, d2)


def test_read_bin_rejects_truncated_or_mismatched_files(pkg, synth, tmp_path):
    h, d = synth.synth_frame(4, "crop")
    p = tmp_path / "bad.bin"
    pkg.packing.write_bin(str(p), h, d[:-5])
    with pytest.raises(ValueError):
        pkg.packing.read_bin(str(p))
    with open(p, "wb") as f:
        f.write(b"\x00" * 10)
    with pytest.raises(ValueError):
        pkg.packing.read_bin(str(p))


def test_pack_frames_layout_and_slicing(pkg, synth):
    frames = [synth.synth_frame(i, "crop") for i in range(7)]
    pk = pkg.packing.pack_frames(frames)
    assert len(pk) == 7 and pk.offsets[0] == 0 and pk.offsets[-1] == pk.depth.size
    for i, (h, d) in enumerate(frames):
        hh, dd = pk.frame(i)
        np.testing.assert_array_equal(hh, h)
        np.testing.assert_array_equal(dd, d)
    sub = pk.slice(2, 5)
    assert len(sub) == 3 and sub.offsets[0] == 0
    np.testing.assert_array_equal(sub.frame(1)[1], frames[3][1])
    # same three arrays as the synthetic generator's batch form
    d2, o2, h2 = synth.synth_batch(7, "crop", 0)
    np.testing.assert_array_equal(pk.depth, d2)
    np.testing.assert_array_equal(pk.offsets, o2)
    np.testing.assert_array_equal(pk.headers, h2)
    empty = pkg.packing.pack_frames([])
    assert len(empty) == 0 and empty.offsets.tolist() == [0]


def test_pack_frames_rejects_inconsistent_frames(pkg, synth):
    h, d = synth.synth_frame(1, "crop")
    with pytest.raises(ValueError):
        pkg.packing.pack_frames([(h, d[:-1])])
    hb = h.copy()
    hb[4] = hb[2]
    with pytest.raises(ValueError):
        pkg.packing.pack_frames([(hb, d)])


def test_gesture_directory_reader(pkg, synth, tmp_path):
    gdir = tmp_path / "P0" / "1"
    gdir.mkdir(parents=True)
    n = 5
    gt = np.random.default_rng(0).normal(0, 50, (n, 63)).astype(np.float32)
    with open(gdir / "joint.txt", "w") as f:
        f.write(f"{n}\n")
        for row in gt:
            f.write(" ".join(f"{v:.6f}" for v in row) + "\n")
    frames = [synth.synth_frame(20 + i, "crop") for i in range(n)]
    for i, (h, d) in enumerate(frames):
        pkg.packing.write_bin(str(gdir / ("%06d_depth.bin" % i)), h, d)
    bin_num, gt2 = pkg.packing.read_joint(str(gdir))
    assert bin_num == n
    np.testing.assert_allclose(gt2, gt, atol=1e-5)
    pk = pkg.packing.pack_bin_files(pkg.packing.gesture_bin_paths(str(gdir)))
    assert len(pk) == n
    np.testing.assert_array_equal(pk.frame(4)[1], frames[4][1])


def test_shard_bounds_cover_and_balance(pkg):
    sb = pkg.shard.shard_bounds
    for n in (0, 1, 7, 8, 1024, 76531):
        for w in (1, 2, 3, 8):
            b = sb(n, w)
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [e - s for s, e in b]
            assert max(sizes) - min(sizes) <= 1
    # weighted: balance by pixels
    rng = np.random.default_rng(1)
    wts = rng.integers(8000, 26000, 1000).astype(np.float64)
    b = sb(1000, 8, wts)
    loads = [wts[s:e].sum() for s, e in b]
    assert b[0][0] == 0 and b[-1][1] == 1000
    assert max(loads) / (wts.sum() / 8) < 1.03
    with pytest.raises(ValueError):
        sb(4, 0)
    with pytest.raises(ValueError):
        pkg.shard.shard_for_rank(4, 2, 2)


@pytest.mark.parametrize("name", golden_names()[:3])
def test_point_cloud_port_matches_reference_run(pkg, golden_dir, name):
    """DataProcess.point_cloud / max_min_point (vectorised here) against the values the reference's own
    loops produced (tools/make_goldens.py -> pc_min, pc_max, pc_n), bit for bit."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    dp = pkg.DataProcess({"header": g["header"], "depth": g["depth"]}, np.zeros(63, np.float32))
    pts = dp.point_cloud()
    assert pts.shape == (int(g["pc_n"]), 3) and pts.dtype == np.float64
    mx, mn = dp.max_min_point(pts)
    np.testing.assert_array_equal(mx, g["pc_max"])
    np.testing.assert_array_equal(mn, g["pc_min"])
    rs = dp.set_length(pts)
    assert rs.shape == (6000, 3)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_entry_points_fail_loudly_without_a_gpu(pkg, synth):
    """No silent CPU fallback anywhere: every entry point raises when there is no HIP device."""
    h, d = synth.synth_frame(0, "crop")
    with pytest.raises(RuntimeError, match="no CPU path"):
        pkg.cal_tsdf_cuda({"header": h, "data": d})
    with pytest.raises(RuntimeError, match="no CPU path"):
        pkg.tsdf_f({"header": h, "depth": d}, np.array([[0, 0, -400.0], [50, 60, -300.0]]))
    with pytest.raises(ValueError, match="no CPU path"):
        pkg.voxelize(torch.from_numpy(d), torch.tensor([0, d.size]), torch.from_numpy(h[None]))
    with pytest.raises(RuntimeError, match="no CPU path"):
        pkg.DataProcess({"header": h, "depth": d}, np.zeros(63), aug=True).process()


def test_missing_library_is_an_import_error(pkg, monkeypatch):
    monkeypatch.setattr(pkg._lib, "_lib", None)
    monkeypatch.setattr(pkg._lib, "LIB_PATH", "/nonexistent/libtsdf_hip.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        pkg._lib.load()


def _make_msra_tree(pkg, synth, root, n_sub=3, n_ges=2, n_frames=3):
    rng = np.random.default_rng(5)
    k = 0
    for s in range(n_sub):
        for g in range(n_ges):
            gdir = root / f"P{s}" / f"{g + 1}"
            gdir.mkdir(parents=True)
            gt = rng.normal(0, 60, (n_frames, 63)).astype(np.float32)
            with open(gdir / "joint.txt", "w") as f:
                f.write(f"{n_frames}\n")
                for row in gt:
                    f.write(" ".join(f"{v:.6f}" for v in row) + "\n")
            for i in range(n_frames):
                h, d = synth.synth_frame(1000 + k, "crop")
                pkg.packing.write_bin(str(gdir / ("%06d_depth.bin" % i)), h, d)
                k += 1
    return n_sub * n_ges * n_frames


def test_on_the_fly_dataset_split_and_items(pkg, synth, tmp_path):
    """Leave-one-subject-out split (3D_CNN/dataset.py:44-53) over raw .bin files; items are the raw
    frame + label; collate gives the voxelizer's three input arrays."""
    total = _make_msra_tree(pkg, synth, tmp_path)
    subs = ["P0", "P1", "P2"]
    tr = pkg.MSRADepthDataset(str(tmp_path), train=True, test_idx=1, subjects=subs)
    te = pkg.MSRADepthDataset(str(tmp_path), train=False, test_idx=1, subjects=subs)
    assert len(tr) == total * 2 // 3 and len(te) == total // 3
    assert all("/P1/" not in p for p in tr.paths) and all("/P1/" in p for p in te.paths)
    h, d, gt = tr[4]
    assert h.shape == (6,) and d.dtype == np.float32 and gt.shape == (63,)
    pk, g = pkg.dataset.collate_frames([tr[i] for i in range(5)])
    assert len(pk) == 5 and g.shape == (5, 63)
    np.testing.assert_array_equal(pk.frame(4)[1], d)
    with pytest.raises(ValueError):
        pkg.MSRADepthDataset(str(tmp_path), size="medium")


def test_offline_export_writes_the_reference_schema(pkg, synth, tmp_path):
    """export.preprocess_tree: the files pre/read_MSRA.py:117-139 writes and 3D_CNN/dataset.py:93-131 reads
    (names, keys, shapes, scalar counts).  The voxelizer is injected (the oracle) so that the file handling
    is checked without a GPU; without the injection the call needs the HIP device."""
    import oracle

    db, out = tmp_path / "db", tmp_path / "result"
    db.mkdir()
    total = _make_msra_tree(pkg, synth, db, n_sub=2, n_ges=2, n_frames=3)

    def fake(pk, res, layout, device):
        r = oracle.voxelize(pk.depth, pk.offsets, pk.headers, R=res, layout=0 if layout == "czyx" else 1)
        return r["tsdf"], r["max_l"], r["mid_p"], r["status"]

    totals = pkg.export.preprocess_tree(str(db), str(out), points_num=500, voxelize_fn=fake,
                                        rng=np.random.default_rng(1))
    assert totals == {"P0": 6, "P1": 6} and sum(totals.values()) == total
    for sub in ("P0", "P1"):
        assert int(np.load(out / f"data_num-{sub}.npy")) == 6
        for d in ("Point_Cloud", "TSDF", "ground_truth", "num"):
            assert sorted(os.listdir(out / sub / d)) == ["1.np" + ("z" if d == "TSDF" else "y"),
                                                         "2.np" + ("z" if d == "TSDF" else "y")]
        z = np.load(out / sub / "TSDF" / "1.npz")
        assert z["tsdf"].shape == (3, 3, 32, 32, 32) and z["tsdf"].dtype == np.float32
        assert z["max_l"].shape == (3,) and z["mid_p"].shape == (3, 3) and not z["status"].any()
        assert np.load(out / sub / "ground_truth" / "1.npy").shape == (3, 63)
        assert int(np.load(out / sub / "num" / "2.npy")) == 3
        pc = np.load(out / sub / "Point_Cloud" / "1.npy")
        assert pc.shape == (3, 500, 3) and np.all(pc[:, :, 2] < 0)       # z = -depth, every point a valid pixel
    # the stored volume is the loop layout [c,x,y,z] (what the reference writer produced) ...
    pk = pkg.packing.pack_bin_files(pkg.packing.gesture_bin_paths(str(db / "P0" / "1")))
    ref = oracle.voxelize(pk.depth, pk.offsets, pk.headers, R=32, layout=0)
    z = np.load(out / "P0" / "TSDF" / "1.npz")
    np.testing.assert_array_equal(z["tsdf"], ref["tsdf"].transpose(0, 1, 4, 3, 2))
    # ... and the switches: numba layout, float64, [n,21,3] labels, no point clouds
    out2 = tmp_path / "result2"
    pkg.export.preprocess_tree(str(db), str(out2), layout="czyx", dtype=np.float64, gt_3d=True,
                               point_clouds=False, subjects=["P1"], gestures=["2"], voxelize_fn=fake)
    z = np.load(out2 / "P1" / "TSDF" / "2.npz")
    assert z["tsdf"].dtype == np.float64 and os.listdir(out2 / "P1" / "Point_Cloud") == []
    assert np.load(out2 / "P1" / "ground_truth" / "2.npy").shape == (3, 21, 3)
    # no injection, no GPU here: must fail loudly, not fall back
    if not torch.cuda.is_available():
        with pytest.raises((ValueError, RuntimeError, AssertionError, pkg.TsdfError)):
            pkg.export.preprocess_tree(str(db), str(tmp_path / "r3"), device="cpu", point_clouds=False)


def test_joint_normalisation_oracle_matches_reference_formula(pkg):
    """(gt - mid_p) / max_l + 0.5 per joint (pre/joint_nor.py:8-18) with the clamp of 3D_CNN/train.py:241-242:
    the C oracle and the numpy restatement against the reference's own loop expression, bit for bit; the product
    function has no CPU path."""
    import pytest
    import oracle
    from oracle import tsdf_oracle_np as onp

    rng = np.random.default_rng(2)
    gt = rng.normal(0, 120, (6, 63)).astype(np.float32)
    max_l = rng.uniform(150, 300, 6).astype(np.float32)
    mid_p = rng.normal(0, 50, (6, 3)).astype(np.float32)
    max_l[4] = 0.0  # a degenerate frame: defined as 0.5 everywhere (the reference divides by zero)
    want = np.empty((6, 21, 3), np.float32)
    with np.errstate(all="ignore"):
        for i in range(6):  # the reference's loop (joint_nor.py:15-16 == train.py:239-240)
            want[i] = (gt[i].reshape(21, 3) - mid_p[i]) / max_l[i] + np.float32(0.5)
    raw = want.copy()
    want[want < 0] = 0  # train.py:241-242
    want[want > 1] = 1
    want[4] = 0.5
    raw[4] = 0.5
    assert (want == 0).any() and (want == 1).any()  # the clamp is exercised
    for fn in (oracle.normalize_joints, onp.normalize_joints):
        np.testing.assert_array_equal(fn(gt, max_l, mid_p).reshape(6, 21, 3), want)
        np.testing.assert_array_equal(fn(gt, max_l, mid_p, clamp=False).reshape(6, 21, 3), raw)
    with pytest.raises(ValueError):
        pkg.normalize_joints(torch.from_numpy(gt), torch.from_numpy(max_l), torch.from_numpy(mid_p))


def test_packed_subject_blob_round_trip_and_fast_reader(pkg, synth, tmp_path):
    """a subject directory -> ONE pack file (headers, offsets, depth, gt, gesture boundaries) that
    is memory-mapped afterwards; the threaded raw-byte reader equals the frame-by-frame read_bin path."""
    total = _make_msra_tree(pkg, synth, tmp_path / "db", n_sub=2, n_ges=3, n_frames=4)
    P = pkg.packing
    g_dir = str(tmp_path / "db" / "P1" / "2")
    slow = P.pack_bin_files(P.gesture_bin_paths(g_dir))
    fast = P.pack_bin_files_fast(P.gesture_bin_paths(g_dir), threads=3)
    for a, b in ((slow.depth, fast.depth), (slow.offsets, fast.offsets), (slow.headers, fast.headers)):
        np.testing.assert_array_equal(a, b)
    paths = P.pack_tree(str(tmp_path / "db"), str(tmp_path / "packs"))
    assert sorted(paths) == ["P0", "P1"] and all(os.path.exists(p) for p in paths.values())
    for mmap in (True, False):
        pk = P.PackedFrames.load(paths["P1"], mmap=mmap)
        assert len(pk) == 12 and pk.group_names == ["1", "2", "3"] and pk.group_start.tolist() == [0, 4, 8, 12]
        np.testing.assert_array_equal(pk.slice(4, 8).depth, slow.depth)
        _, gt = P.read_joint(g_dir)
        np.testing.assert_array_equal(pk.gt[4:8], gt)
        sub = pk.take(np.array([9, 4, 7]))
        np.testing.assert_array_equal(sub.frame(1)[1], slow.frame(0)[1])
        np.testing.assert_array_equal(sub.gt[2], gt[3])
    # a damaged pack is an error, not garbage
    bad = tmp_path / "bad.tsdfpk"
    raw = open(paths["P0"], "rb").read()
    open(bad, "wb").write(raw[: len(raw) // 2])
    with pytest.raises(ValueError):
        P.PackedFrames.load(str(bad))
    open(bad, "wb").write(b"NOTAPACK" + raw[8:])
    with pytest.raises(ValueError):
        P.PackedFrames.load(str(bad))
    # interior damage: first and last offset intact, one in the middle not; and a header that contradicts its payload
    good = P.PackedFrames.load(paths["P0"], mmap=False)
    for what in ("offset", "header"):
        offs, hdrs = good.offsets.copy(), good.headers.copy()
        if what == "offset":
            offs[3], offs[4] = offs[4], offs[3]
        else:
            hdrs[2, 4] += 1
        dmg = tmp_path / f"dmg_{what}.tsdfpk"
        P.PackedFrames(good.depth, good.offsets, good.headers, good.gt, good.group_start, good.group_names).save(str(dmg))
        blob = bytearray(open(dmg, "rb").read())
        pos = 64                                                   # headers, then offsets (64-byte aligned sections)
        blob[pos:pos + hdrs.nbytes] = hdrs.tobytes()
        pos = (pos + hdrs.nbytes + 63) // 64 * 64
        blob[pos:pos + offs.nbytes] = offs.tobytes()
        open(dmg, "wb").write(bytes(blob))
        with pytest.raises(ValueError):
            P.PackedFrames.load(str(dmg))
    with open(tmp_path / "db" / "P0" / "1" / "000001_depth.bin", "ab") as f:
        f.write(b"\0\0\0\0")
    with pytest.raises(ValueError):
        P.pack_subject(str(tmp_path / "db" / "P0"))
    assert total == 24


def test_packed_dataset_equals_file_dataset(pkg, synth, tmp_path):
    """MSRADepthDataset over the packs (built on first use) yields exactly the frames / labels / split of the
    dataset over the .bin tree, and take() assembles the same batches either way."""
    _make_msra_tree(pkg, synth, tmp_path / "db", n_sub=3, n_ges=2, n_frames=3)
    subs = ["P0", "P1", "P2"]
    a = pkg.MSRADepthDataset(str(tmp_path / "db"), train=True, test_idx=0, subjects=subs)
    b = pkg.MSRADepthDataset(str(tmp_path / "db"), train=True, test_idx=0, subjects=subs,
                             packed_dir=str(tmp_path / "packs"))
    assert b.packed and not a.packed and len(a) == len(b) == 12
    assert sorted(os.listdir(tmp_path / "packs")) == ["P1.tsdfpk", "P2.tsdfpk"]   # only the split's subjects
    for i in range(len(a)):
        for x, y in zip(a[i], b[i]):
            np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(b.pixels(), [a[i][1].size for i in range(len(a))])
    for idx in (np.arange(2, 6), np.array([11, 0, 7, 3]), np.arange(4, 9)):   # inside a pack, shuffled, across packs
        pa, pb = a.take(idx), b.take(idx)
        for f in ("depth", "offsets", "headers", "gt"):
            np.testing.assert_array_equal(getattr(pa, f), getattr(pb, f))
    # packs alone are enough (no raw tree on the training box), and n_ges limits by gesture boundaries
    c = pkg.MSRADepthDataset(None, train=False, test_idx=1, subjects=subs, packed_dir=str(tmp_path / "packs"),
                             build_packs=False)
    assert len(c) == 6
    with pytest.raises(FileNotFoundError):
        pkg.MSRADepthDataset(None, train=False, test_idx=0, subjects=subs, packed_dir=str(tmp_path / "packs"),
                             build_packs=False)


def test_plan_batches_contiguous_rank_shards(pkg):
    """Ranks own contiguous, pixel-balanced shards of the frame range (BASELINE configs[3]); batches cover a shard
    exactly once; shuffling stays inside the shard and depends on (seed, epoch)."""
    pb = pkg.dataset.plan_batches
    w = np.random.default_rng(0).integers(8000, 26000, 1000)
    seen = []
    for r in range(8):
        bs = pb(1000, 64, rank=r, world=8, weights=w)
        flat = np.concatenate(bs)
        assert (np.diff(flat) == 1).all()                           # contiguous
        assert all(b.size == 64 for b in bs[:-1]) and 0 < bs[-1].size <= 64
        seen.append(flat)
    allf = np.concatenate(seen)
    np.testing.assert_array_equal(allf, np.arange(1000))            # a partition, in rank order
    loads = [w[s].sum() for s in seen]
    assert max(loads) / (w.sum() / 8) < 1.03
    a = np.concatenate(pb(1000, 64, rank=3, world=8, shuffle=True, seed=5, epoch=0))
    b = np.concatenate(pb(1000, 64, rank=3, world=8, shuffle=True, seed=5, epoch=1))
    base = np.concatenate(pb(1000, 64, rank=3, world=8))
    assert sorted(a) == sorted(b) == sorted(base) and not np.array_equal(a, b)
    assert len(pb(130, 64, drop_last=True)) == 2 and len(pb(130, 64)) == 3


def test_training_loaders_give_every_rank_the_same_number_of_batches(pkg, synth):
    """a training loop that steps a gradient collective once per batch hangs when ranks run out of
    batches at different times.  With MSRA-like crop areas (3x apart between subjects) a pixel-balanced split of 8 ranks
    gives 7 to 15 batches of 1024; the loaders therefore split by FRAME COUNT by default and pad the one rank that
    can come up a batch short; balance="pixels" stays available for export jobs."""
    pb = pkg.dataset.plan_batches
    rng = np.random.default_rng(5)
    # nine "subjects" of 8,500 frames whose crops differ ~3x in area
    w = np.concatenate([rng.integers(int(a * 0.9), int(a * 1.1), 8500) for a in rng.uniform(8000, 26000, 9)])
    n = w.size
    by_px = [len(pb(n, 1024, rank=r, world=8, weights=w)) for r in range(8)]
    assert max(by_px) > min(by_px) + 1                      # the hazard is real
    for world, bs, drop in ((8, 1024, False), (8, 1024, True), (3, 16, False), (7, 64, True), (8, 100, False)):
        plans = [pb(n, bs, rank=r, world=world, shuffle=True, seed=1, epoch=2, drop_last=drop) for r in range(world)]
        assert len({len(p) for p in plans}) == 1, (world, bs, drop)
        assert len({tuple(b.size for b in p) for p in plans}) == 1, (world, bs, drop)   # ... and the same batch SIZES
        flat = np.concatenate([np.concatenate(p) for p in plans])
        if drop:
            assert len(set(flat.tolist())) == flat.size and all(b.size == bs for p in plans for b in p)
        else:
            assert set(flat.tolist()) == set(range(n))      # everything is seen; at most one repeat per rank (the pad)
            assert flat.size - n <= world
    # the one-frame-short case: 2 ranks, 7 frames, batches of 3 -> shards 3 / 4 -> 1 / 2 batches -> padded to 2 / 2
    p0, p1 = pb(7, 3, rank=0, world=2), pb(7, 3, rank=1, world=2)
    assert [b.tolist() for b in p0] == [[0, 1, 2], [0]] and [b.tolist() for b in p1] == [[3, 4, 5], [6]]
    # the loaders: __len__ is planned on the host (no GPU needed) and agrees across ranks by default
    frames = [synth.synth_frame(900 + i, "crop") for i in range(37)]
    pk = pkg.packing.pack_frames(frames)
    pk.gt = np.zeros((37, 63), np.float32)
    ds = pkg.MSRADepthDataset.from_packs([pk])
    for cls in (pkg.VoxelLoader, pkg.ResidentLoader):
        lens = [len(cls(ds, batch_size=4, device="cuda", rank=r, world=4)) for r in range(4)]
        assert len(set(lens)) == 1 and lens[0] == 3, (cls.__name__, lens)
        px = [sum(b.size for b in cls(ds, batch_size=4, device="cuda", rank=r, world=4, balance="pixels")._batches())
              for r in range(4)]
        assert sum(px) == 37
    # the pad is a SAMPLE, not a one-frame batch — 2 ranks, 9 frames, batches of 4: shards 4 / 5,
    # rank 0 repeats a frame so that both cut [4, 1]; fewer frames than ranks is an error, not a silent empty rank
    q0, q1 = pb(9, 4, rank=0, world=2), pb(9, 4, rank=1, world=2)
    assert [b.tolist() for b in q0] == [[0, 1, 2, 3], [0]] and [b.tolist() for b in q1] == [[4, 5, 6, 7], [8]]
    q0, q1 = pb(11, 4, rank=0, world=2), pb(11, 4, rank=1, world=2)
    assert [b.tolist() for b in q0] == [[0, 1, 2, 3], [4, 0]] and [b.tolist() for b in q1] == [[5, 6, 7, 8], [9, 10]]
    with pytest.raises(ValueError):
        pb(3, 2, rank=0, world=4)
    with pytest.raises(ValueError):
        pb(10, 2, balance="pixels")
    with pytest.raises(ValueError):
        pb(10, 2, balance="bytes")


def test_gt_3d_export_round_trips_through_the_reference_reader(pkg, tmp_path):
    """export.write_gesture(gt_3d=True) stores z negated; the reference reader's branch for 3-D label arrays
    (3D_CNN/dataset.py:107-109: negate z, reshape to [n,63]) then gives back the camera-frame labels."""
    gt = np.random.default_rng(1).normal(0, 50, (4, 63)).astype(np.float32)
    pkg.export.write_gesture(str(tmp_path / "P0"), "1", np.zeros((4, 3, 4, 4, 4), np.float32), np.ones(4, np.float32),
                             np.zeros((4, 3), np.float32), gt, gt_3d=True)
    ground_truth = np.load(tmp_path / "P0" / "ground_truth" / "1.npy").astype(np.float32)
    assert ground_truth.shape == (4, 21, 3)
    if len(ground_truth.shape) == 3:                 # the reader's own lines, restated
        ground_truth[:, :, 2] = -ground_truth[:, :, 2]
        g_t = ground_truth.reshape(-1, 63)
    np.testing.assert_array_equal(g_t, gt)


def test_native_gather_equals_numpy_gather(pkg):
    """PackedFrames.take of a shuffled batch goes through tsdf_host_gather_frames (threads, host memory only) when the
    library is there: same bytes and offsets as the frame-by-frame numpy copy, into a caller's buffer too; the C entry
    rejects indices outside the pack and a destination that is too small."""
    import ctypes
    packing = pkg.packing
    rng = np.random.default_rng(12)
    frames = []
    for k in range(60):
        bw, bh = int(rng.integers(1, 90)), int(rng.integers(1, 70))
        frames.append((np.array([320, 240, 3, 4, 3 + bw, 4 + bh], np.int32), rng.normal(400, 30, bw * bh).astype(np.float32)))
    pk = packing.pack_frames(frames)
    idx = rng.permutation(60)[:37].astype(np.int64)
    want = np.concatenate([frames[i][1] for i in idx])
    got = pk.take(idx)
    np.testing.assert_array_equal(got.depth, want)
    np.testing.assert_array_equal(np.diff(got.offsets), [frames[i][1].size for i in idx])
    np.testing.assert_array_equal(got.headers, np.stack([frames[i][0] for i in idx]))
    buf = np.full(want.size + 100, -1.0, np.float32)
    got2 = pk.take(idx, buf)
    np.testing.assert_array_equal(got2.depth, want)
    assert (buf[want.size:] == -1.0).all()
    L = pkg._lib.load()
    off2 = np.zeros(3, np.int64)
    bad = np.array([0, 60], np.int64)
    args = lambda ix, cap: (pk.depth.ctypes.data, pk.offsets.ctypes.data, 60, ix.ctypes.data, ix.size, buf.ctypes.data, cap,
                            off2.ctypes.data, 4)
    assert L.tsdf_host_gather_frames(*args(bad, buf.size)) == -1
    assert L.tsdf_host_gather_frames(*args(np.array([0, 1], np.int64), 1)) == -1
    assert L.tsdf_host_gather_frames(*args(np.array([5, 5], np.int64), buf.size)) == 0 and off2[2] == 2 * frames[5][1].size
    # ABI v5: the entry that knows the source's length refuses offsets that leave it, or that run backwards (a damaged
    # pack), before copying anything — the v4 entry could only trust them
    args_n = lambda offs, ix, src_len: (pk.depth.ctypes.data, src_len, offs.ctypes.data, 60, ix.ctypes.data, ix.size,
                                        buf.ctypes.data, buf.size, off2.ctypes.data, 4)
    two = np.array([5, 59], np.int64)
    assert L.tsdf_host_gather_frames_n(*args_n(pk.offsets, two, pk.depth.size)) == 0
    assert L.tsdf_host_gather_frames_n(*args_n(pk.offsets, two, pk.depth.size - 1)) == -1     # last frame leaves the buffer
    assert L.tsdf_host_gather_frames_n(*args_n(pk.offsets, two, -5)) == -1
    broken = pk.offsets.copy()
    broken[6] = broken[5] - 3                                                                 # frame 5 runs backwards
    assert L.tsdf_host_gather_frames_n(*args_n(broken, two, pk.depth.size)) == -1
    assert L.tsdf_host_gather_frames(pk.depth.ctypes.data, broken.ctypes.data, 60, two.ctypes.data, 2, buf.ctypes.data,
                                     buf.size, off2.ctypes.data, 4) == -1
    broken = pk.offsets.copy()
    broken[5] = -1
    assert L.tsdf_host_gather_frames_n(*args_n(broken, np.array([5, 4], np.int64), pk.depth.size)) == -1
    bad_pk = packing.PackedFrames(pk.depth, broken, pk.headers)
    with pytest.raises(ValueError):
        bad_pk.take(np.array([9, 5, 4, 7, 1], np.int64))


def test_prebatched_ring_reference_count_rule(pkg):
    """The host logic of MSRA_Dataset's pre-batched ring (ADVICE rounds 3-4): a slot counts as held
    while the consumer keeps its batch tuple, one of its tensors, a view of one, or an ALIAS of one that does not refer to
    the tensor object at all (detach(), .data, numpy()) — and then gets fresh tensors instead of being overwritten.  The
    rule must also hold inside a generator frame and under a sys.settrace hook (which keeps frames, hence locals, alive
    longer: that may only make a slot look held, never free).  (The GPU tier runs the real ring under DataLoader; here
    the GPU-facing parts are stubbed out.)"""
    import ctypes
    import sys

    ds = pkg.dataset
    Fast = ds.MSRA_Dataset._Fast
    assert ds._slot_guard_works()

    class HostOnly(Fast):
        def __init__(self):
            class L:
                @staticmethod
                def TsdfLabels(*a):
                    return ctypes.c_int(0)
            self._ctypes, self._lib, self.always_fresh = ctypes, L, False
            self.bs, self.nc, self.device, self.rp_gt = 4, 63, "cpu", torch.zeros(2, 63)
            self.ring, self.count, self.replaced = 2, 0, 0
            self.slots, self.labels, self.args, self.results, self.guard = ([None] * 2 for _ in range(5))
            for k in range(2):
                self._fresh(k)

    def scenario():
        f = HostOnly()
        assert not f.held(0) and not f.held(1)
        b = ds._collate_prebatched(f.results[0])          # what torch's default_collate hands the consumer
        assert f.held(0) and not f.held(1)
        del b
        assert not f.held(0)
        t = f.results[0][0].batch[0]
        assert f.held(0)
        v = t[:, 1]                                        # a view: its _base is the slot's tensor
        del t
        assert f.held(0)
        del v
        assert not f.held(0)
        # aliases that do NOT refer to the slot's tensor object: only the storage's use count sees them
        for alias in (lambda x: x.detach(), lambda x: x.data, lambda x: x.numpy(), lambda x: x.view(-1)[3:5].detach(),
                      lambda x: torch.as_strided(x, (2,), (1,))):
            for which in range(4):
                a = alias(f.results[0][0].batch[which])
                assert f.held(0) and not f.held(1), (which, alias)
                del a
                assert not f.held(0)
        assert f.next_slot() == 0 and f.replaced == 0      # free slot: reused as it is
        keep = f.results[0][0].batch[2].detach()           # an alias, not the tensor
        ptr = keep.data_ptr()
        assert f.next_slot() == 0 and f.replaced == 1      # held slot: new tensors, the kept one untouched
        assert ptr != f.results[0][0].batch[2].data_ptr() and keep.data_ptr() == ptr
        del keep
        # a consumer that is a generator: the batch lives in its suspended frame
        def consumer():
            held = ds._collate_prebatched(f.results[1])
            yield 1
            yield held[0].shape
        g = consumer()
        assert not f.held(1)
        next(g)
        assert f.held(1)                                   # suspended frame holds the batch
        next(g)
        g.close()
        del g
        assert not f.held(1)
        # a torch on which the counting rules do not hold: every batch gets fresh tensors
        f.always_fresh = True
        assert f.held(0) and f.held(1)
        return True

    assert scenario()
    # the same under a trace hook (debuggers, coverage tools): frames and locals are visible to it, nothing may read as free
    # that is held
    seen = []
    def tracer(frame, event, arg):
        if event == "call" and len(seen) < 4:
            seen.append(frame)                              # keeps a few frames alive on purpose
        return None
    sys.settrace(tracer)
    try:
        assert scenario()
    finally:
        sys.settrace(None)
    del seen[:]


def test_design_quotes_the_tracked_rocprof_numbers():
    """"docs and tracked profiles disagree by 3 %".  DESIGN.md's rocprofv3 column must be what
    tools/collect_profiles.py prints from the TRACKED profiles/r05/summary.json: every average / steady-state figure of that
    table appears in DESIGN.md literally (and the minima of the two 32^3 rows).  Round 5: the resource figures DESIGN quotes
    (scratch bytes per lane of the three BASELINE instantiations) are those of the tracked profiles/r05/resources.txt, which
    tools/resources.sh writes from the compiler's own remarks ("docs that disagree with the binary")."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "collect_profiles.py"), "-", "table", "r05"],
                       capture_output=True, text=True, cwd=root)
    assert r.returncode == 0, r.stderr
    design = open(os.path.join(root, "DESIGN.md")).read()
    rows = [ln for ln in r.stdout.splitlines() if ln.startswith("| ") and "workload" not in ln and "---" not in ln]
    assert len(rows) >= 4
    for ln in rows:
        cells = [c.strip() for c in ln.strip("|").split("|")]
        want = [cells[2], cells[4]] + ([cells[3]] if "32^3" in cells[0] else [])    # average, steady state; minimum at 32^3
        for num in want:
            if re.fullmatch(r"\d+\.\d", num):
                assert num in design, f"DESIGN.md does not quote {num} ({cells[0]})"
    res = {}
    for ln in open(os.path.join(root, "profiles", "r05", "resources.txt")):
        if ln.startswith("#") or "|" not in ln:
            continue
        name, vgpr, spilled, scratch, occ, lds = [c.strip() for c in ln.split("|")]
        res[name] = (int(vgpr), int(spilled), int(scratch), int(occ), int(lds))
    assert len(res) >= 30
    fused = {k: v for k, v in res.items() if k.startswith("tsdf_fused_kernel")}
    assert all(v[0] <= 128 and v[3] == 4 and v[4] == 163776 for v in fused.values())
    for inst, phrase in (("tsdf_fused_kernel<32, 0, false, false, 2>", "{} B/lane for `<32,0,false,false,2>`"),
                         ("tsdf_fused_kernel<64, 0, false, false, 1>", "{} for `<64,0,false,false,1>`"),
                         ("tsdf_fused_kernel<64, 0, true, false, 1>", "{} for `<64,0,true,false,1>`")):
        assert phrase.format(res[inst][2]) in design, (inst, res[inst])
    assert f"up to {max(v[2] for v in fused.values())}" in design
    split = [v for k, v in res.items() if k.startswith("tsdf_split_kernel")]
    assert all(v[2] == 0 for v in split) and f"{min(v[0] for v in split)}–{max(v[0] for v in split)} VGPRs" in design
