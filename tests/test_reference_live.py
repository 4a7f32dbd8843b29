"""CPU tier, build container only: the oracle against the reference RUN LIVE on frames that are NOT among the committed
fixtures.  tests/golden/ pins 22 frames; this runs the reference's own loop (pre/tsdf_for.py::tsdf_f / tsdf_cal through
tools/make_goldens.run_reference: as it runs today and on float64-typed parameters = the numba typing) on 33 more seeded
frames of every family — benchmark distributions, near / far hands, corner bboxes, sparse, dense, negative and mixed-sign
depths — and asks the oracle for the same bits.  Skipped where /root/reference does not exist (the GPU box: nothing there may
read it); nothing here touches a GPU."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/pre"), reason="the reference is only in the build container")


@pytest.fixture(scope="module")
def mg():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        return importlib.import_module("make_goldens")     # imports /root/reference/pre/{tsdf_for,process,joint_nor}.py
    finally:
        sys.path.pop(0)


def _frames(mg, synth):
    out = []
    for s in (40, 41):
        out.append((f"full_{s}", *synth.synth_frame(s, "full")))
    for s in (60, 61, 62, 63):
        out.append((f"crop_{s}", *synth.synth_frame(s, "crop")))
    for name, _, kw in mg.VARIANTS:                        # two unseen seeds of every family the fixtures hold
        for s in (200, 201):
            out.append((f"{name}#{s}", *synth.synth_variant(s, **kw)))
    # mixed signs at two different distances: the camera plane off the grid's centre
    h, d = synth.synth_variant(202, bbox=(60, 30, 260, 210), base=500.0, rad=80.0)
    xs = np.arange(d.size) % 200
    out.append(("mixed_two_distances", h, np.where(xs < 100, -0.3 * d, d).astype(np.float32)))
    return out


def test_oracle_equals_the_reference_loop_on_unseen_frames(mg, synth):
    import oracle

    n_flip_frames = 0
    frames = _frames(mg, synth)
    assert len(frames) >= 33
    for name, h, d in frames:
        with np.errstate(all="ignore"):
            g = mg.run_reference(h, d)
        out, pm = oracle.voxels(d, h, g["vox_ori"], g["voxel_len"], g["trunc"], R=32, layout=0, want_pixmap=True)
        np.testing.assert_array_equal(out, g["loop64"], err_msg=name)                 # a4, numba typing: bit for bit
        bad = (np.abs(out - g["loop32"]) > 1e-5).any(axis=0)
        assert bad.sum() == int(g["n_flip"]), name                                     # a4', the loop as it runs
        n_flip_frames += int(g["n_flip"]) > 0
        grid, ori = oracle.glue(g["aabb_min"], g["aabb_max"], 32)                     # a3: the reference's own glue
        np.testing.assert_array_equal(grid[:3], g["mid_p"], err_msg=name)
        assert grid[3] == g["max_l"] and grid[4] == g["voxel_len"] and grid[5] == g["trunc"], name
        np.testing.assert_array_equal(ori, g["vox_ori"], err_msg=name)
        nv, mn, mx = oracle.aabb(d, h)                                                 # a2': CPU witness, 1 ulp
        assert nv == int(g["pc_n"])
        np.testing.assert_allclose(mn, g["pc_min"], rtol=3e-7, atol=0, err_msg=name)
        np.testing.assert_allclose(mx, g["pc_max"], rtol=3e-7, atol=0, err_msg=name)
        # the search emulation of tools/make_goldens.py stays the reference's float32 loop
        np.testing.assert_array_equal(mg.loop32_emulation(d, h, g["vox_ori"], g["voxel_len"], g["trunc"]), g["loop32"])
    print(f"{len(frames)} unseen frames through the reference loop; {n_flip_frames} of them with float32/float64 flips")
