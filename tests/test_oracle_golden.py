"""CPU tier: the C oracle (oracle/tsdf_oracle.c) against goldens produced by RUNNING the
reference (tools/make_goldens.py -> /root/reference/pre/tsdf_for.py, pre/process.py), and
against the independent numpy restatement (oracle/tsdf_oracle_np.py)."""
import os

import numpy as np
import pytest

import oracle
from oracle import tsdf_oracle_np as onp
from conftest import golden_names

NAMES = golden_names()


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


@pytest.mark.parametrize("name", NAMES)
def test_voxels_match_reference_loop_numba_typing(golden_dir, name):
    """a4: tsdf_cal run by the reference on float64-typed parameters == numba typing.
    The oracle must reproduce it bit for bit after the float32 store."""
    g = load(golden_dir, name)
    out, pm = oracle.voxels(g["depth"], g["header"], g["vox_ori"], g["voxel_len"], g["trunc"],
                            R=32, layout=0, want_pixmap=True)
    assert out.shape == (3, 32, 32, 32)
    np.testing.assert_array_equal(out, g["loop64"])
    # rejected voxels (outside the bbox / invalid depth) are zero in all three channels
    assert np.all(out[:, pm < 0] == 0)


@pytest.mark.parametrize("name", NAMES)
def test_voxels_within_tol_of_reference_loop_as_it_runs(golden_dir, name):
    """a4': the loop as it runs today (float32 scalars under numpy 2) differs from the numba typing only by rounding
    (<= 1e-5) except where a pixel index / threshold / sign test flips.  The golden records how many
    voxels flip, the `*_flip` fixtures were searched for so that the count is not zero: the oracle — float64, the numba
    typing — must disagree with loop32 on exactly those voxels and agree with loop64 everywhere."""
    g = load(golden_dir, name)
    out = oracle.voxels(g["depth"], g["header"], g["vox_ori"], g["voxel_len"], g["trunc"])
    bad = (np.abs(out - g["loop32"]) > 1e-5).any(axis=0)
    assert bad.sum() == int(g["n_flip"])
    assert (int(g["n_flip"]) > 0) == name.endswith("_flip")
    np.testing.assert_array_equal(bad, (np.abs(g["loop64"] - g["loop32"]) > 1e-5).any(axis=0))


def test_the_golden_set_covers_what_the_survey_called_hard(golden_dir):
    """>= 20 frames; at least two with float32/float64 flips, and every kind of flip present among them (a pixel index
    one off, and a voxel one typing rejects or truncates and the other does not); hands at 150 mm and at 1,500 mm, a
    bbox far off the principal point, ~1 % valid and fully valid crops, negative and mixed-sign depths (a grid that
    straddles the camera plane)."""
    gs = {n: load(golden_dir, n) for n in NAMES}
    assert len(gs) >= 20
    flips = [n for n, g in gs.items() if int(g["n_flip"]) > 0]
    assert len(flips) >= 2 and sum(int(gs[n]["n_flip"]) for n in flips) >= 20
    mid_z = {n: float(g["mid_p"][2]) for n, g in gs.items()}
    assert min(abs(z) for z in mid_z.values() if abs(z) > 50) < 160 and min(mid_z.values()) < -1400
    assert any(z > 100 for z in mid_z.values())                                     # all-negative depths: z = +d
    straddle = [n for n, g in gs.items() if abs(mid_z[n]) < float(g["max_l"]) / 2]  # z = 0 inside the cube
    assert len(straddle) >= 3
    for n in straddle:
        g = gs[n]
        vz = g["vox_ori"][2] + np.arange(32) * g["voxel_len"]
        assert (vz < 0).any() and (vz > 0).any()
        nz = (g["loop64"] != 0).any(axis=0)           # [z,y,x]: valid voxels on BOTH sides of the camera plane
        assert nz[vz < 0].any() and nz[vz > 0].any(), n
    frac = {n: int(g["n_valid"]) / g["depth"].size for n, g in gs.items()}
    assert min(frac.values()) < 0.012 and max(frac.values()) == 1.0
    off = {n: max(abs(int(g["header"][2]) + int(g["header"][4]) - 320), abs(int(g["header"][3]) + int(g["header"][5]) - 240)) / 2
           for n, g in gs.items()}
    assert max(off.values()) >= 100                                                  # bbox centre >= 100 px off (160, 120)
    # kinds of flip, voxel by voxel: another pixel gathered (both typings write a value, the values differ), a voxel only the
    # float32 loop writes, a voxel only the float64 loop writes
    kinds = {"value": 0, "only32": 0, "only64": 0}
    for n in flips:
        g = gs[n]
        bad = (np.abs(g["loop32"] - g["loop64"]) > 1e-5).any(axis=0)
        a, b = (g["loop32"] != 0).any(axis=0), (g["loop64"] != 0).any(axis=0)
        kinds["value"] += int((bad & a & b).sum())
        kinds["only32"] += int((bad & a & ~b).sum())
        kinds["only64"] += int((bad & ~a & b).sum())
    assert all(v > 0 for v in kinds.values()), kinds


@pytest.mark.parametrize("name", NAMES)
def test_layout_cxyz_is_transpose(golden_dir, name):
    """App. B#10: loop layout [c,x,y,z] == numba layout [c,z,y,x] transposed."""
    g = load(golden_dir, name)
    a = oracle.voxels(g["depth"], g["header"], g["vox_ori"], g["voxel_len"], g["trunc"], layout=0)
    b = oracle.voxels(g["depth"], g["header"], g["vox_ori"], g["voxel_len"], g["trunc"], layout=1)
    np.testing.assert_array_equal(b, a.transpose(0, 3, 2, 1))


@pytest.mark.parametrize("name", NAMES)
def test_glue_matches_reference(golden_dir, name):
    """a3: grid placement computed by the reference's own tsdf_f (tsdf_for.py:11-16) from the
    full-pixel AABB — float32, bit exact."""
    g = load(golden_dir, name)
    grid, ori = oracle.glue(g["aabb_min"], g["aabb_max"], 32)
    np.testing.assert_array_equal(grid[:3], g["mid_p"])
    assert grid[3] == g["max_l"] and grid[4] == g["voxel_len"] and grid[5] == g["trunc"]
    np.testing.assert_array_equal(ori, g["vox_ori"])


@pytest.mark.parametrize("name", NAMES)
def test_aabb_restatement_and_cpu_witness(golden_dir, name):
    """a2: numba-typing AABB (restated, not executed) — C == numpy restatement bit exact; the
    reference's runnable CPU analogue (process.py point_cloud + max_min_point, float32 x/y) agrees
    to 1 ulp-level (App. A.1)."""
    g = load(golden_dir, name)
    nv, mn, mx = oracle.aabb(g["depth"], g["header"])
    assert nv == int(g["n_valid"]) == int(g["pc_n"])
    np.testing.assert_array_equal(mn, g["aabb_min"])
    np.testing.assert_array_equal(mx, g["aabb_max"])
    np.testing.assert_allclose(mn, g["pc_min"], rtol=3e-7, atol=0)
    np.testing.assert_allclose(mx, g["pc_max"], rtol=3e-7, atol=0)


@pytest.mark.parametrize("name", NAMES)
def test_end_to_end_frame(golden_dir, name):
    """a5: cal_tsdf_cuda contract — (tsdf, max_l, mid_p) from one call."""
    g = load(golden_dir, name)
    off = np.array([0, g["depth"].size], np.int64)
    res = oracle.voxelize(g["depth"], off, g["header"][None], R=32, extras=True)
    assert res["status"][0] == 0
    np.testing.assert_array_equal(res["tsdf"][0], g["loop64"])
    assert res["max_l"][0] == g["max_l"]
    np.testing.assert_array_equal(res["mid_p"][0], g["mid_p"])


def test_c_vs_numpy_restatement_many_frames(synth):
    """Two independent restatements agree bit for bit on 40 seeded frames of the benchmark distributions and 20 of the
    families outside them — near / far hands, corner bboxes, sparse, negative and mixed-sign depths, where the grid straddles
    the camera plane and q = -F / v_z takes both signs and large magnitudes — incl. pixel maps."""
    variants = [dict(bbox=(40, 20, 300, 230), base=150.0, rad=100.0, bulge=30.0),
                dict(bbox=(130, 90, 190, 150), base=1500.0, rad=18.0, bulge=25.0),
                dict(bbox=(0, 0, 110, 100), base=380.0, rad=45.0),
                dict(bbox=(60, 40, 260, 200), base=420.0, rad=75.0, keep=0.012),
                dict(bbox=(80, 60, 240, 200), base=450.0, rad=60.0, sign="neg"),
                dict(bbox=(90, 50, 230, 190), base=300.0, rad=60.0, sign="halves"),
                dict(bbox=(90, 50, 230, 190), base=250.0, rad=55.0, sign="checker"),
                dict(bbox=(120, 80, 200, 160), base=40.0, rad=35.0, bulge=10.0, sign="halves"),
                dict(bbox=(0, 0, 320, 240), base=120.0, rad=90.0, bulge=50.0, sign="checker"),
                dict(bbox=(200, 120, 320, 240), base=700.0, rad=50.0, sign="halves")]
    frames = [synth.synth_frame(1000 + s, "crop" if s % 2 else "full") for s in range(40)]
    frames += [synth.synth_variant(300 + k, **variants[k % len(variants)]) for k in range(20)]
    for h, d in frames:
        nv, mn, mx = oracle.aabb(d, h)
        nv2, mn2, mx2 = onp.aabb(d, h)
        assert nv == nv2
        np.testing.assert_array_equal(mn, mn2)
        np.testing.assert_array_equal(mx, mx2)
        mid, max_l, vl, tr, ori = onp.glue(mn, mx, 32)
        grid, ori_c = oracle.glue(mn, mx, 32)
        np.testing.assert_array_equal(ori, ori_c)
        assert (grid[3], grid[4], grid[5]) == (max_l, vl, tr)
        out_c, pm_c = oracle.voxels(d, h, ori, vl, tr, want_pixmap=True)
        out_n, pm_n = onp.voxels(d, h, ori, vl, tr)
        np.testing.assert_array_equal(pm_c, pm_n)
        np.testing.assert_array_equal(out_c, out_n)


def test_augmented_oracle_identity_equals_plain_path(pkg, synth):
    """Re-specified augmented form (parity unpinned): with the identity map it must equal the pinned
    plain path bit for bit; the packed inverse must invert the forward map."""
    d, o, h = synth.synth_batch(4, "crop", 31)
    ident = pkg.augment.identity_affines(4)
    a = oracle.voxelize_aug(d, o, h, ident, R=32)
    b = oracle.voxelize(d, o, h, R=32)
    np.testing.assert_array_equal(a["tsdf"], b["tsdf"])
    np.testing.assert_array_equal(a["max_l"], b["max_l"])
    np.testing.assert_array_equal(a["mid_p"], b["mid_p"])
    xf, prm = pkg.augment.random_affines(b["mid_p"], rng=3)
    assert xf.shape == (4, 24) and np.all((prm["stretch"] >= 2 / 3) & (prm["stretch"] <= 3 / 2))
    assert np.all((prm["rot_xy"] >= -30) & (prm["rot_xy"] < 30))
    pts = np.random.default_rng(0).normal(0, 100, (4, 21, 3))
    fwd = pkg.augment.apply_affine(pts, xf)
    inv = xf.reshape(4, 2, 3, 4)[:, 1]
    back = np.einsum("nij,nkj->nki", inv[:, :, :3], fwd) + inv[:, None, :, 3]
    np.testing.assert_allclose(back, pts, atol=1e-9)
    # the centre is a fixed point of the map
    np.testing.assert_allclose(pkg.augment.apply_affine(b["mid_p"][:, None, :].astype(np.float64), xf)[:, 0],
                               b["mid_p"], atol=1e-9)
    # a pure translation moves the grid and leaves the volume unchanged
    t = np.array([12.5, -7.25, 30.0])
    tr = pkg.augment.pack_affine(np.tile(np.eye(3), (4, 1, 1)), np.tile(t, (4, 1)))
    c = oracle.voxelize_aug(d, o, h, tr, R=32)
    np.testing.assert_allclose(c["mid_p"], b["mid_p"] + t, atol=1e-4)
    assert np.abs(c["tsdf"] - b["tsdf"]).max() < 1e-3  # float32 re-rounding of the moved AABB only
