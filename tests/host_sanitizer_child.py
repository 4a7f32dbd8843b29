"""Child of tests/test_tiers_cpu.py::test_product_host_code_under_sanitizers — runs with the sanitizer runtime preloaded
and drives build/libtsdf_host_{asan,tsan}.so: the PRODUCT's host-only code (csrc/tsdf_host.inc) compiled by g++ with
-fsanitize=address,undefined or -fsanitize=thread.  Any sanitizer report aborts the process; a wrong result is an
AssertionError.  argv[1] = the library, argv[2] = "asan" | "tsan"."""
import ctypes
import sys

import numpy as np

L = ctypes.CDLL(sys.argv[1])
mode = sys.argv[2]
vp, i64 = ctypes.c_void_p, ctypes.c_int64
L.tsdf_host_gather_frames.restype = ctypes.c_int
L.tsdf_host_gather_frames.argtypes = [vp, vp, i64, vp, i64, vp, i64, vp, ctypes.c_int]
L.tsdf_host_gather_frames_n.restype = ctypes.c_int
L.tsdf_host_gather_frames_n.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, vp, ctypes.c_int]
L.tsdf_test_slot_hammer.restype = ctypes.c_int
L.tsdf_test_slot_hammer.argtypes = [ctypes.c_int] * 3
L.tsdf_test_slot_acquire.restype = ctypes.c_int
L.tsdf_test_slot_acquire.argtypes = [ctypes.c_size_t, ctypes.c_int]
L.tsdf_test_slot_release.restype = None
L.tsdf_test_slot_release.argtypes = [ctypes.c_size_t, ctypes.c_int]
L.tsdf_test_slot_next_epoch.restype = ctypes.c_uint32
L.tsdf_test_slot_next_epoch.argtypes = [ctypes.c_int]


class Cam(ctypes.Structure):
    _fields_ = [("focal", ctypes.c_double), ("cx", ctypes.c_double), ("cy", ctypes.c_double),
                ("invalid_eps", ctypes.c_float), ("trunc_voxels", ctypes.c_float)]


class Labels(ctypes.Structure):
    _fields_ = [("d_gt", vp), ("n_joints", ctypes.c_int), ("clamp", ctypes.c_int), ("d_out_gt_nor", vp), ("d_out_gt_aug", vp)]


L.tsdf_test_check_run_args.restype = ctypes.c_int
L.tsdf_test_check_run_args.argtypes = [vp, i64, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(Cam), ctypes.c_int, vp,
                                       ctypes.c_int, ctypes.POINTER(Labels)]

rng = np.random.default_rng(20261005)
# ---- the threaded gather against numpy, small (one thread) and large (every worker) -------------------------------
for n_src, n, lo, hi, threads in ((40, 25, 1, 200, 8), (300, 220, 2000, 9000, 16), (300, 220, 2000, 9000, 64), (5, 0, 1, 10, 4)):
    lens = rng.integers(lo, hi, n_src)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    src = rng.normal(400, 30, int(off[-1])).astype(np.float32)
    idx = rng.integers(0, n_src, n).astype(np.int64)
    want = np.concatenate([src[off[i]:off[i + 1]] for i in idx]) if n else np.zeros(0, np.float32)
    dst = np.full(want.size + 64, -7.0, np.float32)
    doff = np.zeros(n + 1, np.int64)
    for entry in (0, 1):
        dst[:] = -7.0
        if entry == 0:
            rc = L.tsdf_host_gather_frames(src.ctypes.data, off.ctypes.data, n_src, idx.ctypes.data, n, dst.ctypes.data,
                                           dst.size, doff.ctypes.data, threads)
        else:
            rc = L.tsdf_host_gather_frames_n(src.ctypes.data, src.size, off.ctypes.data, n_src, idx.ctypes.data, n,
                                             dst.ctypes.data, dst.size, doff.ctypes.data, threads)
        assert rc == 0, rc
        assert np.array_equal(dst[:want.size], want) and (dst[want.size:] == -7.0).all()
        assert np.array_equal(np.diff(doff), lens[idx])
    # ---- damaged packs and bad requests: refused before a byte is copied (the destination stays untouched) ----
    if n:
        def refused(o=off, ix=idx, cap=dst.size, src_len=src.size, ns=n_src):
            dst[:] = -7.0
            rc = L.tsdf_host_gather_frames_n(src.ctypes.data, src_len, o.ctypes.data, ns, ix.ctypes.data, ix.size,
                                             dst.ctypes.data, cap, doff.ctypes.data, threads)
            return rc == -1 and (dst == -7.0).all()
        bad = idx.copy(); bad[n // 2] = n_src
        assert refused(ix=bad)
        bad[n // 2] = -1
        assert refused(ix=bad)
        assert refused(cap=want.size - 1)
        assert refused(src_len=int(off[idx.max() + 1]) - 1)          # the last frame taken leaves the source
        assert refused(src_len=-3)
        o2 = off.copy(); f = int(idx[0]); o2[f + 1] = o2[f] - 1          # a frame that runs backwards
        assert refused(o=o2)
        o3 = off.copy(); o3[int(idx[1])] = -5
        assert refused(o=o3)
        o4 = off.copy(); o4[int(idx[0]) + 1] = np.iinfo(np.int64).max   # sizes whose sum would overflow
        assert refused(o=o4, src_len=np.iinfo(np.int64).max, ix=np.array([idx[0], idx[0]], np.int64))
assert L.tsdf_host_gather_frames_n(None, 0, None, 0, None, 0, None, 0, None, 4) == 0        # nothing asked for, nothing needed
z = np.zeros(1, np.int64)
assert L.tsdf_host_gather_frames_n(None, 0, None, 0, None, 0, None, 0, z.ctypes.data, 4) == 0 and z[0] == 0

# ---- the argument checks every voxelizer entry starts with ----------------------------------------------------------
one, null = vp(64), vp(0)
ok = lambda **kw: L.tsdf_test_check_run_args(*[kw.get(k, d) for k, d in (
    ("depth", one), ("depth_len", 16), ("offsets", one), ("headers", one), ("n", 1), ("R", 32), ("cam", None),
    ("layout", 0), ("out", one), ("aabb_only", 0), ("labels", None))])
assert ok() == 0 and ok(n=0) == 1 and ok(n=0, depth=null, out=null) == 1
for kw in (dict(n=-1), dict(R=30), dict(R=132), dict(layout=2), dict(depth=null), dict(offsets=null), dict(headers=null),
           dict(depth_len=-1), dict(out=null), dict(out=vp(72))):
    assert ok(**kw) == -1, kw
assert ok(out=null, aabb_only=1) == 0
nan = float("nan")
for cam in (Cam(0.0, 160, 120, 1, 3), Cam(nan, 160, 120, 1, 3), Cam(241.42, 160, 120, 0, 3), Cam(241.42, 160, 120, 1, nan)):
    assert ok(cam=ctypes.pointer(cam)) == -1
assert ok(cam=ctypes.pointer(Cam(241.42, 160, 120, 1, 3))) == 0
assert ok(labels=ctypes.pointer(Labels(64, 21, 1, 64, 0))) == 0 and ok(labels=ctypes.pointer(Labels(64, 0, 1, 64, 0))) == -1
assert ok(labels=ctypes.pointer(Labels(0, 21, 1, 64, 0))) == -1 and ok(labels=ctypes.pointer(Labels(0, 21, 1, 0, 0)), n=0) == 1
assert ok(labels=ctypes.pointer(Labels(64, 171, 1, 64, 0)), n=0) == -1

# ---- the per-stream slot table -----------------------------------------------------------------------------------------
a, b = L.tsdf_test_slot_acquire(0xA0, 0), L.tsdf_test_slot_acquire(0xB0, 0)
assert a >= 0 and b >= 0 and a != b and L.tsdf_test_slot_acquire(0xA0, 0) == a
e = [L.tsdf_test_slot_next_epoch(a) for _ in range(3)]
assert e == [e[0], e[0] + 1, e[0] + 2] and e[0] >= 1
L.tsdf_test_slot_release(0xA0, 0)
assert L.tsdf_test_slot_acquire(0xC0, 0) == a                          # the freed slot is handed out again ...
assert L.tsdf_test_slot_next_epoch(a) == e[2] + 1                      # ... and its epochs go on where they were
for s in (0xB0, 0xC0):
    L.tsdf_test_slot_release(s, 0)
got = [L.tsdf_test_slot_acquire(0x100000 + 16 * k, 0) for k in range(70)]
assert sorted(g for g in got if g >= 0) == list(range(64)) and got.count(-1) == 6    # full: -1, never a shared slot
for k in range(70):
    L.tsdf_test_slot_release(0x100000 + 16 * k, 0)
bad = L.tsdf_test_slot_hammer(8 if mode == "asan" else 6, 5, 300 if mode == "asan" else 120)
assert bad == 0, bad
print("host code ok under", mode)
