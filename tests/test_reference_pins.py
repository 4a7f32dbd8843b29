"""Rows a1 and f1 of SURVEY.md section 8 pinned to RUNS of the reference (fixtures written by tools/make_goldens.py in
the build container; nothing here reads /root/reference):

  io_ref.npz       pre/read_MSRA.py::read_bin / read_joint (:143-164) on tiny files — their bytes and what the reference
                   returned.  packing.read_bin / read_joint / pack_bin_files[_fast] must reproduce it bit for bit.
  dataset_ref.npz  3D_CNN/dataset.py::MSRA_Dataset (:16-117), train and test, over an export of the synthetic tree
                   synth.synth_msra_tree(**tree) with the ORACLE as voxelizer: len, item order, dtypes, shapes, gt /
                   max_l / mid_p of every item, a digest of every volume, a few volumes whole (reference writer layout
                   [c,x,y,z]).  pkg.MSRA_Dataset on the RAW tree must yield the same items (GPU tier).

One documented difference: the reference's read_joint returns a 1-D (63,) array for a gesture with ONE frame
(np.loadtxt squeezes); packing.read_joint returns [1,63] so that `ground_truth[i, :]` (pre/read_MSRA.py:106) means the
same for every gesture length.
"""
import hashlib
import os

import numpy as np
import pytest
import torch

import oracle


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint8)


@pytest.fixture(scope="module")
def io_ref(golden_dir):
    return np.load(os.path.join(golden_dir, "io_ref.npz"))


@pytest.fixture(scope="module")
def ds_ref(golden_dir):
    return np.load(os.path.join(golden_dir, "dataset_ref.npz"))


def _write_bins(io_ref, root):
    paths = []
    for name in io_ref["bin_names"]:
        p = os.path.join(str(root), "%s.bin" % name)
        io_ref["bin_%s_bytes" % name].tofile(p)
        paths.append(p)
    return paths


def test_read_bin_reproduces_the_reference_reader(pkg, io_ref, tmp_path):
    """packing.read_bin == pre/read_MSRA.py::read_bin (:155-164) on the same bytes: dtype, shape and every bit
    (the payload of one file holds CR / LF / ^Z / NaN-pattern bytes: the reference opens it in text mode)."""
    for name, p in zip(io_ref["bin_names"], _write_bins(io_ref, tmp_path)):
        h, d = pkg.packing.read_bin(p)
        rh, rd = io_ref["bin_%s_header" % name], io_ref["bin_%s_depth" % name]
        assert h.dtype == rh.dtype == np.int32 and d.dtype == rd.dtype == np.float32
        assert h.shape == rh.shape == (6,) and d.shape == rd.shape
        np.testing.assert_array_equal(h, rh)
        np.testing.assert_array_equal(_bits(d), _bits(rd))


@pytest.mark.parametrize("fast", [False, True])
def test_pack_bin_files_is_the_reference_reads_back_to_back(pkg, io_ref, tmp_path, fast):
    """The batch wire format = what the reference's reader returned for each file, concatenated."""
    paths = _write_bins(io_ref, tmp_path)
    pk = (pkg.packing.pack_bin_files_fast if fast else pkg.packing.pack_bin_files)(paths)
    want_d = np.concatenate([io_ref["bin_%s_depth" % n] for n in io_ref["bin_names"]])
    want_h = np.stack([io_ref["bin_%s_header" % n] for n in io_ref["bin_names"]])
    np.testing.assert_array_equal(_bits(pk.depth), _bits(want_d))
    np.testing.assert_array_equal(pk.headers, want_h)
    np.testing.assert_array_equal(np.diff(pk.offsets), [io_ref["bin_%s_depth" % n].size for n in io_ref["bin_names"]])
    assert pk.offsets.dtype == np.int64 and pk.offsets[0] == 0


def test_read_joint_reproduces_the_reference_reader(pkg, io_ref, tmp_path):
    """packing.read_joint == pre/read_MSRA.py::read_joint (:143-152): count and float32 values bit for bit; the
    one-frame gesture keeps its row axis here ([1,63]) where the reference returns (63,)."""
    for n in (1, 3):
        g = tmp_path / ("ges%d" % n)
        g.mkdir()
        io_ref["joint%d_bytes" % n].tofile(str(g / "joint.txt"))
        cnt, gt = pkg.packing.read_joint(str(g))
        ref = io_ref["joint%d_gt" % n]
        assert cnt == int(io_ref["joint%d_count" % n]) == n
        assert gt.dtype == ref.dtype == np.float32 and gt.shape == (n, 63)
        assert ref.shape == ((63,) if n == 1 else (3, 63))          # what the reference returned
        np.testing.assert_array_equal(_bits(gt), _bits(ref.reshape(n, 63)))


def _tree_kwargs(ds_ref):
    kw = dict(s.split("=") for s in ds_ref["tree"])
    return dict(n_sub=int(kw["n_sub"]), n_ges=int(kw["n_ges"]), n_frames=int(kw["n_frames"]), seed=int(kw["seed"]),
                kind=kw["kind"])


def _oracle_items(pkg, raw):
    """The items of a raw dataset through the oracle, in the reference writer's layout [c,x,y,z]."""
    pk = raw.take(np.arange(len(raw)))
    r = oracle.voxelize(pk.depth, pk.offsets, pk.headers, R=32, layout=1, n_threads=4)
    return r, pk


@pytest.mark.parametrize("split", ["train", "test"])
def test_raw_dataset_walks_the_tree_like_the_reference_class(pkg, synth, ds_ref, tmp_path, split):
    """CPU tier: the raw on-the-fly dataset (leave-one-subject-out, `small` = 4 x 5, test_idx = 2 — the reference's
    hard-coded values, 3D_CNN/dataset.py:20-31,44-53) holds the frames the reference class returned, in its order:
    labels bit for bit (the reader's z flip of the pre-negated export, :107-109, gives the joint.txt values back), and
    the oracle on those frames reproduces every volume's digest, max_l and mid_p — i.e. the regenerated tree and the
    oracle on this machine are the ones the fixture was written from."""
    total = synth.synth_msra_tree(str(tmp_path / "db"), **_tree_kwargs(ds_ref))
    assert total == int(ds_ref["total_frames"])
    raw = pkg.MSRADepthDataset(str(tmp_path / "db"), train=(split == "train"), test_idx=2, size="small")
    n = int(ds_ref[split + "_len"])
    assert len(raw) == n == (30 if split == "train" else 10)
    r, pk = _oracle_items(pkg, raw)
    np.testing.assert_array_equal(_bits(pk.gt), _bits(ds_ref[split + "_gt"]))
    np.testing.assert_array_equal(r["max_l"], ds_ref[split + "_max_l"])
    np.testing.assert_array_equal(r["mid_p"], ds_ref[split + "_mid_p"])
    sha = [hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() for v in r["tsdf"]]
    assert sha == list(ds_ref[split + "_tsdf_sha256"])
    np.testing.assert_array_equal(r["tsdf"][ds_ref[split + "_kept"]], ds_ref[split + "_tsdf_kept"])
    assert list(ds_ref[split + "_item_types"]) == ["ndarray:float32:(3, 32, 32, 32)", "ndarray:float32:(63,)",
                                                    "float32:float32:()", "ndarray:float32:(3,)"]


# ---------------------------------------------------------------------------------------------- GPU tier
@pytest.mark.gpu
def test_reference_read_files_through_the_hip_path(pkg, io_ref, tmp_path):
    """a1 end to end: the fixture's files -> pack_bin_files -> tsdf_voxelize_hip == the oracle on the arrays the
    REFERENCE's reader returned for those files (the NaN-pattern file is a degenerate or tiny frame: status and scalars
    must agree too)."""
    dev = torch.device("cuda:0")
    paths = _write_bins(io_ref, tmp_path)
    pk = pkg.packing.pack_bin_files(paths)
    depth = np.concatenate([io_ref["bin_%s_depth" % n] for n in io_ref["bin_names"]])
    headers = np.stack([io_ref["bin_%s_header" % n] for n in io_ref["bin_names"]])
    off = np.concatenate([[0], np.cumsum([io_ref["bin_%s_depth" % n].size for n in io_ref["bin_names"]])]).astype(np.int64)
    ref = oracle.voxelize(depth, off, headers, R=32, layout=0)
    out = pkg.voxelize(*pk.to_torch(dev, pin=False, non_blocking=False), res=32)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.status.cpu().numpy(), ref["status"])
    np.testing.assert_array_equal(out.max_l.cpu().numpy(), ref["max_l"])
    np.testing.assert_array_equal(out.mid_p.cpu().numpy(), ref["mid_p"])
    assert np.abs(out.tsdf.cpu().numpy() - ref["tsdf"]).max() <= 1e-5
    assert (ref["status"] == 0).sum() >= 3


@pytest.mark.gpu
@pytest.mark.parametrize("split", ["train", "test"])
def test_msra_dataset_yields_the_reference_class_items(pkg, synth, ds_ref, tmp_path, split):
    """f1: pkg.MSRA_Dataset(raw tree, opt=None, train) — the reference's constructor call (3D_CNN/train.py:86,89) —
    against what the reference's own class returned for the exported tree: same length, same order, the tuple of
    :73-79 with the same shapes and dtypes; gt, max_l, mid_p exact; volumes <= 1e-5 after [c,z,y,x] -> [c,x,y,z] (the
    reference writer went through the CPU loop's layout, pre/tsdf_for.py:118-120; this class returns the numba layout,
    the documented default) — the kept volumes against the fixture, every volume against the oracle whose digests the
    CPU tier ties to the fixture.  Then the same through the reference's loader call, DataLoader(batch_size=16)."""
    synth.synth_msra_tree(str(tmp_path / "db"), **_tree_kwargs(ds_ref))
    train = split == "train"
    ds = pkg.MSRA_Dataset(str(tmp_path / "db"), None, train=train)
    n = int(ds_ref[split + "_len"])
    assert len(ds) == n
    raw = pkg.MSRADepthDataset(str(tmp_path / "db"), train=train, test_idx=2, size="small")
    r, _ = _oracle_items(pkg, raw)
    assert [hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() for v in r["tsdf"]] == \
        list(ds_ref[split + "_tsdf_sha256"])
    kept = {int(i): k for k, i in enumerate(ds_ref[split + "_kept"])}
    for i in range(n):
        item = ds[i]
        assert len(item) == 4
        tsdf, gt, max_l, mid_p = (t.cpu().numpy() for t in item)
        assert tsdf.shape == (3, 32, 32, 32) and tsdf.dtype == np.float32
        assert gt.shape == (63,) and gt.dtype == np.float32 and mid_p.shape == (3,) and max_l.shape == ()
        np.testing.assert_array_equal(_bits(gt), _bits(ds_ref[split + "_gt"][i]))
        assert max_l == ds_ref[split + "_max_l"][i]
        np.testing.assert_array_equal(mid_p, ds_ref[split + "_mid_p"][i])
        cxyz = tsdf.transpose(0, 3, 2, 1)
        assert np.abs(cxyz - r["tsdf"][i]).max() <= 1e-5
        if i in kept:
            assert np.abs(cxyz - ds_ref[split + "_tsdf_kept"][kept[i]]).max() <= 1e-5
    # the reference's loader call (train.py:86-91; shuffle off so that the order can be compared)
    dl = torch.utils.data.DataLoader(ds, batch_size=16, shuffle=False, num_workers=0)
    a = 0
    for tsdf, gt, max_l, mid_p in dl:
        b = a + tsdf.shape[0]
        np.testing.assert_array_equal(_bits(gt.cpu().numpy()), _bits(ds_ref[split + "_gt"][a:b]))
        np.testing.assert_array_equal(max_l.cpu().numpy(), ds_ref[split + "_max_l"][a:b])
        np.testing.assert_array_equal(mid_p.cpu().numpy(), ds_ref[split + "_mid_p"][a:b])
        assert np.abs(tsdf.cpu().numpy().transpose(0, 1, 4, 3, 2) - r["tsdf"][a:b]).max() <= 1e-5
        a = b
    assert a == n
