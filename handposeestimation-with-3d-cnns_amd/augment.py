"""3-D augmentation parameters for the fused augmented voxelizer (host side, numpy).

The reference's ``DataProcess.data_aug`` (pre/process.py:202-261) draws a stretch ``s ~ U(2/3, 3/2)``
for x and y (z = 1), integer rotation angles in [-30, 30) degrees, builds ``R = Rx(t)·Ry(t)·Rz(t)`` and
maps points as ``(p·S - m)·R + m``.  It cannot run as written (``np.mean(..., axis=2)`` on 2-D input,
SURVEY.md App. B#8) and has further defects there listed: ``R_z`` reuses the x/y angle (``rot_z`` is
drawn but unused), the "centre" ``m`` is each point's own coordinate mean.  Here the same
distributions are drawn, ``rot_z`` is used for ``Rz`` and ``m`` is a real centre (the un-augmented
grid centre ``mid_p``) about which the cloud is stretched AND rotated, giving the affine map
``T(p) = Rᵀ·S·(p - m) + m = A·p + b``  with ``A = Rᵀ·S``, ``b = m - A·m`` (column-vector form of the
reference's row-vector expression, with ``m`` a fixed point of the map).
The same ``T`` applies to the joint labels (:232-249): :func:`apply_affine`.
"""
from __future__ import annotations

import numpy as np


def pack_affine(A: np.ndarray, b: np.ndarray) -> np.ndarray:
    """(A [..,3,3], b [..,3]) -> float64[..,24]: forward rows {A_i0,A_i1,A_i2,b_i} then the inverse map."""
    A = np.asarray(A, np.float64)
    b = np.asarray(b, np.float64)
    Ai = np.linalg.inv(A)
    bi = -np.einsum("...ij,...j->...i", Ai, b)
    fwd = np.concatenate([A, b[..., None]], axis=-1).reshape(*A.shape[:-2], 12)
    inv = np.concatenate([Ai, bi[..., None]], axis=-1).reshape(*A.shape[:-2], 12)
    return np.ascontiguousarray(np.concatenate([fwd, inv], axis=-1))


def identity_affines(n: int) -> np.ndarray:
    return pack_affine(np.tile(np.eye(3), (n, 1, 1)), np.zeros((n, 3)))


def rotation_xyz(rx_deg, ry_deg, rz_deg) -> np.ndarray:
    """R = Rx·Ry·Rz with the reference's matrix conventions (pre/process.py:218-224)."""
    ax, ay, az = (np.deg2rad(np.asarray(v, np.float64)) for v in (rx_deg, ry_deg, rz_deg))
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    z, o = np.zeros_like(cx), np.ones_like(cx)
    Rx = np.stack([np.stack([o, z, z], -1), np.stack([z, cx, sx], -1), np.stack([z, -sx, cx], -1)], -2)
    Ry = np.stack([np.stack([cy, z, -sy], -1), np.stack([z, o, z], -1), np.stack([sy, z, cy], -1)], -2)
    Rz = np.stack([np.stack([cz, sz, z], -1), np.stack([-sz, cz, z], -1), np.stack([z, z, o], -1)], -2)
    return Rx @ Ry @ Rz


def reference_draw(rs: "np.random.RandomState"):
    """The three draws of one ``data_aug`` call, in the reference's order and with its calls
    (pre/process.py:209, 215, 216): ``uniform(2/3, 3/2)``, ``randint(-30, 30)``, ``randint(-30, 30)`` on numpy's
    legacy generator (``np.random.seed(k)`` + module functions == ``RandomState(k)`` methods).
    Returns (stretch_xy, rot_xy, rot_z) — the last one is drawn by the reference but never used."""
    stretch = rs.uniform(2 / 3, 3 / 2)
    rot_xy = rs.randint(-30, 30)
    rot_z = rs.randint(-30, 30)
    return float(stretch), int(rot_xy), int(rot_z)


def reference_matrices(stretch_xy: float, rot_xy: int):
    """``S`` and ``R`` exactly as ``data_aug`` builds them (pre/process.py:210-224): S = diag(s, s, 1),
    R = Rx(t)·Ry(t)·Rz(t) with t = rot_xy degrees — R_z reuses the x/y angle."""
    S = np.diag(np.array([stretch_xy, stretch_xy, 1.0]))
    R = rotation_xyz(rot_xy, rot_xy, rot_xy)
    return S, R


def reference_data_aug(points: np.ndarray, S: np.ndarray, R: np.ndarray) -> np.ndarray:
    """The reference's point mapping as written (pre/process.py:251-259 for the cloud, :237-249 for the joints):
    ``(p·S - m)·R + m`` in row-vector form, where the "centre" m repeats the mean of the point's OWN three
    stretched coordinates (``np.mean(..., axis=2)``) — the defect ``random_affines`` fixes by using a real
    centre.  points [..., 3]; used by the tests that pin this module's conventions to the reference's output."""
    p = np.asarray(points, np.float64)
    ps = p @ S
    m = ps.mean(axis=-1, keepdims=True)
    return (ps - m) @ R + m


def draw_params(n: int, rng=None):
    """The augmentation parameters of n frames with the reference's distributions (pre/process.py:209-216):
    ``(stretch float64[n], rot_xy int64[n], rot_z int64[n])``.  ``rng``: seed / ``np.random.Generator``; or a legacy
    ``np.random.RandomState``, in which case every frame takes the reference's own three draws in its order
    (:func:`reference_draw`)."""
    if isinstance(rng, np.random.RandomState):
        draws = [reference_draw(rng) for _ in range(n)]
        return (np.array([d[0] for d in draws], np.float64), np.array([d[1] for d in draws], np.int64),
                np.array([d[2] for d in draws], np.int64))
    rng = np.random.default_rng(rng)
    return rng.uniform(2 / 3, 3 / 2, n), rng.integers(-30, 30, n), rng.integers(-30, 30, n)


def affines_from_params(centres: np.ndarray, stretch, rot_xy, rot_z) -> np.ndarray:
    """The maps ``T(p) = Rᵀ·S·(p - m) + m`` of the module docstring for centres m [n,3] and per-frame parameters:
    float64[n,24] (forward rows then inverse rows)."""
    m = np.asarray(centres, np.float64).reshape(-1, 3)
    n = m.shape[0]
    S = np.zeros((n, 3, 3))
    S[:, 0, 0] = stretch
    S[:, 1, 1] = stretch
    S[:, 2, 2] = 1.0
    Rt = np.swapaxes(rotation_xyz(rot_xy, rot_xy, rot_z), -1, -2)
    A = Rt @ S
    b = m - np.einsum("nij,nj->ni", A, m)
    return pack_affine(A, b)


def random_affines(centres: np.ndarray, rng=None):
    """One augmentation per frame with the reference's distributions (pre/process.py:209-216).

    centres  [n,3]  the point each frame is stretched/rotated about (use the un-augmented ``mid_p``).
    rng      seed / ``np.random.Generator``; or a legacy ``np.random.RandomState``, in which case every frame
             takes the reference's own three draws in its order (:func:`reference_draw`).
    Returns (xforms float64[n,24], params dict(stretch, rot_xy, rot_z)).
    """
    m = np.asarray(centres, np.float64).reshape(-1, 3)
    stretch, rot_xy, rot_z = draw_params(m.shape[0], rng)
    return affines_from_params(m, stretch, rot_xy, rot_z), dict(stretch=stretch, rot_xy=rot_xy, rot_z=rot_z)


def apply_affine(points: np.ndarray, xforms: np.ndarray) -> np.ndarray:
    """T(p) for points [n,k,3] (or [n,63] joint rows) with xforms [n,24]; returns the input's shape."""
    pts = np.asarray(points, np.float64)
    shp = pts.shape
    p = pts.reshape(shp[0], -1, 3)
    f = np.asarray(xforms, np.float64).reshape(-1, 24)[:, :12].reshape(-1, 3, 4)
    out = np.einsum("nij,nkj->nki", f[:, :, :3], p) + f[:, None, :, 3]
    return out.reshape(shp)
