"""CPU tiers SURVEY.md sections 4 and 5 ask for on top of the golden-vector tests:

  * the label formula pinned to a RUN of the reference's pre/joint_nor.py::normalize (tests/golden/joint_nor_ref.npz,
    written by tools/make_goldens.py) instead of a restatement;
  * tier 3 — hypothesis properties of the oracle (range, shared sign and zero mask, layout transpose, padding the
    bounding box with invalid pixels changes nothing);
  * the C restatement under AddressSanitizer + UBSan (oracle/Makefile `asan`): the golden tests re-run in a child
    process against that build.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import oracle
from oracle import tsdf_oracle_np as onp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- labels: pre/joint_nor.py:8-18 as the reference runs it -------------------------------------------------
def test_label_formula_against_a_run_of_the_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "joint_nor_ref.npz"))
    gt, max_l, mid_p, want = g["gt"], g["max_l"], g["mid_p"], g["joint_nor"]
    assert gt.dtype == np.float32 and want.dtype == np.float64 and want.shape == gt.shape
    # the reference computed in float32 and stored into a float64 array: the values are float32 values
    assert np.array_equal(want.astype(np.float32).astype(np.float64), want)
    clamped = want.copy()
    clamped[clamped < 0] = 0            # 3D_CNN/train.py:241-242
    clamped[clamped > 1] = 1
    assert (clamped != want).any()
    n = gt.shape[0]
    for fn in (oracle.normalize_joints, onp.normalize_joints):
        np.testing.assert_array_equal(fn(gt.reshape(n, 63), max_l, mid_p, clamp=False).reshape(n, 21, 3), want)
        np.testing.assert_array_equal(fn(gt.reshape(n, 63), max_l, mid_p, clamp=True).reshape(n, 21, 3), clamped)


# ---- tier 3: properties of the oracle ---------------------------------------------------------------------------
def _frame(seed, bw, bh, l, t, valid_frac, near):
    """A small frame: a blob of valid depths (`near`..near+80 mm) in a bw x bh bounding box at (l, t)."""
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:bh, 0:bw]
    cx, cy = rng.uniform(0.3, 0.7) * bw, rng.uniform(0.3, 0.7) * bh
    rad = max(2.0, valid_frac * min(bw, bh))
    inside = (xs - cx) ** 2 + (ys - cy) ** 2 < rad ** 2
    d = near + 40.0 * np.sqrt(np.clip(1 - ((xs - cx) ** 2 + (ys - cy) ** 2) / rad ** 2, 0, 1)) + rng.normal(0, 1, (bh, bw))
    d = np.where(inside & (rng.random((bh, bw)) > 0.03), d, 0.0).astype(np.float32)
    if not (np.abs(d) >= 1).any():
        d[bh // 2, bw // 2] = near
    return np.array([320, 240, l, t, l + bw, t + bh], np.int32), d


frames = st.builds(_frame, seed=st.integers(0, 2 ** 31 - 1), bw=st.integers(3, 48), bh=st.integers(3, 40),
                   l=st.integers(0, 200), t=st.integers(0, 150), valid_frac=st.floats(0.1, 0.6),
                   near=st.floats(200.0, 900.0))
# derandomize: the same examples on every run (the driver's CPU tier must not depend on a random seed or a local database)
prop = settings(max_examples=60, deadline=None, derandomize=True, database=None,
                suppress_health_check=[HealthCheck.too_slow])


@prop
@given(fr=frames, R=st.sampled_from([8, 16, 24]))
def test_property_range_sign_and_zero_mask(fr, R):
    """pre/tsdf_numba.py:33-35,54-68: values in [-1,1]; the three channels are zero together (rejected voxel) and
    share their sign; a far voxel is (+-1,+-1,+-1)."""
    h, d = fr
    r = oracle.voxelize(d.reshape(-1), np.array([0, d.size]), h[None], R=R)
    assert r["status"][0] == 0
    v = r["tsdf"][0]
    assert np.isfinite(v).all() and np.abs(v).max() <= 1.0
    zero = v == 0
    # a voxel rejected by :36/:40 is zero in all channels; an accepted one may have single components equal to 0 only
    # when that distance is exactly 0, which the sign bit still marks (-0.0 / +0.0 are both == 0): compare masks of
    # "all three zero" against "any zero" through the sign-bearing representation
    allz = zero.all(axis=0)
    neg = np.signbit(v)
    acc = ~allz
    assert (neg[:, acc] == neg[0][acc]).all()          # shared sign where the voxel was accepted
    far = (np.abs(v) == 1).all(axis=0)
    assert ((np.abs(v[:, far]) == 1).all())
    # (no claim that some voxel is accepted: a 3-pixel-wide box under an 8^3 grid can miss every voxel centre)


@prop
@given(fr=frames, R=st.sampled_from([8, 16]))
def test_property_layout_cxyz_is_the_transpose(fr, R):
    """pre/tsdf_for.py:118-120 writes [c,x,y,z], pre/tsdf_numba.py:70-72 [c,z,y,x]: the same numbers."""
    h, d = fr
    a = oracle.voxelize(d.reshape(-1), np.array([0, d.size]), h[None], R=R, layout=0)["tsdf"][0]
    b = oracle.voxelize(d.reshape(-1), np.array([0, d.size]), h[None], R=R, layout=1)["tsdf"][0]
    np.testing.assert_array_equal(a, b.transpose(0, 3, 2, 1))


@prop
@given(fr=frames, pad=st.tuples(st.integers(0, 9), st.integers(0, 9), st.integers(0, 9), st.integers(0, 9)),
       R=st.sampled_from([8, 16]), junk=st.sampled_from([0.0, 0.5, -0.99, float("nan")]))
def test_property_padding_the_bbox_with_invalid_pixels_changes_nothing(fr, pad, R, junk):
    """A larger bounding box around the same pixels — the added ones invalid (|d| < 1, or NaN by this project's
    rule) — gives the same AABB (pre/tsdf_numba.py:87-89), hence the same grid, and every voxel that projected
    outside the old box (:36) now hits an invalid pixel (:40): the volume, max_l and mid_p are bit-identical."""
    h, d = fr
    pl, pt, pr, pb = pad
    l, t = int(h[2]) - pl, int(h[3]) - pt
    if l < 0 or t < 0:
        pl, pt, l, t = 0, 0, int(h[2]), int(h[3])
    bh, bw = d.shape
    big = np.full((bh + pt + pb, bw + pl + pr), junk, np.float32)
    big[pt:pt + bh, pl:pl + bw] = d
    h2 = np.array([320, 240, l, t, l + big.shape[1], t + big.shape[0]], np.int32)
    a = oracle.voxelize(d.reshape(-1), np.array([0, d.size]), h[None], R=R)
    b = oracle.voxelize(big.reshape(-1), np.array([0, big.size]), h2[None], R=R)
    for k in ("tsdf", "max_l", "mid_p", "status"):
        np.testing.assert_array_equal(a[k], b[k])


@prop
@given(fr=frames, R=st.sampled_from([8, 16]))
def test_property_c_and_numpy_restatements_agree(fr, R):
    """Two independent restatements of SURVEY.md Appendix A (C, numpy) agree bit for bit, pixel map included."""
    h, d = fr
    nv, mn, mx = oracle.aabb(d.reshape(-1), h)
    nv2, mn2, mx2 = onp.aabb(d.reshape(-1), h)
    assert nv == nv2
    np.testing.assert_array_equal(mn, mn2)
    np.testing.assert_array_equal(mx, mx2)
    grid, ori = oracle.glue(mn, mx, R)
    if not grid[3] > 0:
        return
    out_c, pm_c = oracle.voxels(d.reshape(-1), h, ori, grid[4], grid[5], R=R, want_pixmap=True)
    out_n, pm_n = onp.voxels(d.reshape(-1), h, ori, grid[4], grid[5], R=R)
    np.testing.assert_array_equal(pm_c, pm_n)
    np.testing.assert_array_equal(out_c, out_n)


# ---- sanitizers: the C restatement under ASan + UBSan -------------------------------------------------------------
def test_oracle_golden_tests_pass_under_asan_ubsan(tmp_path):
    """"run the CPU restatement under ASan/UBSan".  Builds oracle/libtsdf_oracle_asan.so and
    re-runs the golden-vector and augmentation tests against it in a child interpreter with the sanitizer runtime
    preloaded; any report aborts the child (halt_on_error) and fails this test."""
    if os.environ.get("TSDF_ORACLE_SO"):
        pytest.skip("already inside the sanitizer child")
    asan_rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan_rt or not os.path.isabs(asan_rt) or not os.path.exists(asan_rt):
        pytest.skip("no libasan next to gcc")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-B", "asan"], stdout=subprocess.DEVNULL)
    so = os.path.join(ROOT, "oracle", "libtsdf_oracle_asan.so")
    env = dict(os.environ)
    env.update(TSDF_ORACLE_SO=so, LD_PRELOAD=asan_rt,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1",   # (CPython itself "leaks")
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_golden.py"),
                        os.path.join(ROOT, "tests", "test_tiers_cpu.py"),
                        "-k", "not asan_ubsan"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail


# ---- sanitizers: the PRODUCT's host-only code under ASan + UBSan and under TSan -------------------------------------
@pytest.mark.parametrize("mode", ["asan", "tsan"])
def test_product_host_code_under_sanitizers(mode):
    """VERDICT round 3, missing #5: the product's own host code ran unsanitized.  csrc/tsdf_host.inc — the threaded
    gather of the loaders with its offset validation (a damaged pack is refused before a byte is copied), the argument
    checks of every voxelizer entry and the (stream, thread) slot table with its launch epochs — is the SAME source
    text libtsdf_hip.so compiles; here g++ builds it with -fsanitize=address,undefined / -fsanitize=thread
    (csrc/Makefile host-asan, host-tsan) and tests/host_sanitizer_child.py drives it in a child interpreter with the
    sanitizer runtime preloaded: results against numpy, every refusal, 8 threads hammering the table.  CPU build only."""
    rt_name = "libasan.so" if mode == "asan" else "libtsan.so"
    rt = subprocess.run(["gcc", "-print-file-name=" + rt_name], capture_output=True, text=True).stdout.strip()
    if not rt or not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("no %s next to gcc" % rt_name)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "handposeestimation-with-3d-cnns_amd", "csrc"), "-B", "host-" + mode],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    so = os.path.join(ROOT, "build", "libtsdf_host_%s.so" % mode)
    env = dict(os.environ)
    env.update(LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", TSAN_OPTIONS="halt_on_error=1:report_signal_unsafe=0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "host_sanitizer_child.py"), so, mode],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    tail = (r.stdout + r.stderr)[-3000:]
    if mode == "tsan" and r.returncode != 0 and ("unexpected memory mapping" in tail or "FATAL: ThreadSanitizer" in tail
                                                 and "mmap" in tail):
        pytest.skip("ThreadSanitizer cannot map its shadow in this container: " + tail[-300:])
    assert r.returncode == 0, tail
    assert "host code ok under " + mode in r.stdout
    assert "Sanitizer" not in tail and "runtime error" not in tail, tail
