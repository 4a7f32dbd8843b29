/*
 * tsdf_oracle.c — CPU restatement of the reference's projective-TSDF path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only
 * as the checker / the reported CPU baseline.  Nothing under
 * handposeestimation-with-3d-cnns_amd/ imports, links or calls it; the product
 * path is the HIP library and has no CPU fallback.
 *
 * What it restates (all citations into /root/reference):
 *   - AABB over all valid pixels      pre/tsdf_numba.py:84-96 (per pixel), :140-141 (final min/max)
 *   - grid placement ("host glue")    pre/tsdf_numba.py:142-147  (== pre/tsdf_for.py:11-16)
 *   - per-voxel TSDF                  pre/tsdf_numba.py:15-72    (== pre/tsdf_for.py:62-120 formula)
 * with the arithmetic types numba infers for that file:
 * float32 parameters, float64 intermediates (FOCAL is a Python float), unfused
 * multiply-then-add for the pixel index, int() truncation toward zero, float32
 * store.  Every division here is a true IEEE division.
 *
 * Parity pinning: tests/test_oracle_golden.py checks this file against golden
 * vectors produced by running the reference's own pre/tsdf_for.py::tsdf_cal
 * (the only runnable implementation) on float64-typed copies of the float32
 * parameters — which makes the reference loop evaluate exactly the numba
 * typing — and on the loop as it runs today (float32 scalars under numpy 2).
 * The AABB half (min_max_kernel) has no runnable reference: it is pinned by
 * restatement only ("numba path: restated, not executed").
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/tsdf.h"

#ifdef _OPENMP
#include <omp.h>
#endif

/* int() of a float64: truncate toward zero.  Out-of-range values saturate and NaN
 * gives 0 (the gfx950 v_cvt_i32_f64 rule); both only arise for grids that touch the
 * camera plane z = 0, which is outside the depth-camera domain. */
static int32_t trunc_i32(double v) {
  if (v != v) return 0;
  if (v >= 2147483647.0) return INT32_MAX;
  if (v <= -2147483648.0) return INT32_MIN;
  return (int32_t)v;
}

static const tsdf_cam k_default_cam = {241.42, 160.0, 120.0, 1.0f, 3.0f};

/*
 * A.1 — pre/tsdf_numba.py:84-96 per pixel, :140-141 reduction.  Bounds-checked over
 * exactly N = b_w*b_h pixels (the reference's out-of-bounds read at :83-86 and the
 * "drop last block" hack at :138-139 are defects).
 * Returns the number of valid pixels; min_p/max_p are untouched when it is 0.
 */
long tsdf_oracle_aabb(const float *depth, const int32_t *header, const tsdf_cam *cam,
                      float *min_p, float *max_p) {
  if (!cam) cam = &k_default_cam;
  const int l = header[2], t = header[3], r = header[4], b = header[5];
  const int bw = r - l, bh = b - t;
  long n_valid = 0;
  float mn[3] = {INFINITY, INFINITY, INFINITY};
  float mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int row = 0; row < bh; ++row) {
    for (int col = 0; col < bw; ++col) {
      const long pos = (long)row * bw + col;
      const int x = col + l;          /* pos % b_w + l   :84 */
      const int y = row + t;          /* pos // b_w + t  :85 (integer division, App. B#3) */
      const float d = depth[pos];     /* :86 */
      if (!(fabsf(d) >= cam->invalid_eps)) continue; /* :87; NaN is invalid (include/tsdf.h) */
      const double q = (double)d / cam->focal;                 /* :91 */
      const float cxp = (float)(q * ((double)x - cam->cx));    /* :92, rounded by the f32 smem store :95 */
      const float cyp = (float)(-q * ((double)y - cam->cy));   /* :93 */
      const float czp = -d;                                    /* :94 */
      if (cxp < mn[0]) mn[0] = cxp;
      if (cyp < mn[1]) mn[1] = cyp;
      if (czp < mn[2]) mn[2] = czp;
      if (cxp > mx[0]) mx[0] = cxp;
      if (cyp > mx[1]) mx[1] = cyp;
      if (czp > mx[2]) mx[2] = czp;
      ++n_valid;
    }
  }
  if (n_valid) {
    memcpy(min_p, mn, sizeof mn);
    memcpy(max_p, mx, sizeof mx);
  }
  return n_valid;
}

/*
 * A.2 — pre/tsdf_numba.py:142-147, all float32, evaluated left to right.
 * grid[8] = mid_p[3], max_l, voxel_len, trunc_dis, 0, 0 ; ori[3] = vox_ori.
 */
void tsdf_oracle_glue(const float *min_p, const float *max_p, int R, const tsdf_cam *cam,
                      float *grid, float *ori) {
  if (!cam) cam = &k_default_cam;
  float mid[3], len[3];
  for (int a = 0; a < 3; ++a) {
    mid[a] = (min_p[a] + max_p[a]) / 2.0f; /* :142 */
    len[a] = max_p[a] - min_p[a];          /* :143 */
  }
  float max_l = len[0];                    /* :144 */
  if (len[1] > max_l) max_l = len[1];
  if (len[2] > max_l) max_l = len[2];
  const float voxel_len = max_l / (float)R;              /* :145 */
  const float trunc_dis = voxel_len * cam->trunc_voxels; /* :146 */
  for (int a = 0; a < 3; ++a) {
    float v = mid[a] - max_l / 2.0f;       /* :147 */
    ori[a] = v + voxel_len / 2.0f;
  }
  grid[0] = mid[0]; grid[1] = mid[1]; grid[2] = mid[2];
  grid[3] = max_l; grid[4] = voxel_len; grid[5] = trunc_dis;
  grid[6] = 0.0f; grid[7] = 0.0f;
}

/*
 * A.3 — pre/tsdf_numba.py:15-72 for every voxel of an R^3 grid.
 * out: float32[3][R][R][R] in `layout`.  pixmap (optional, int32[R^3] indexed
 * [z][y][x]): the gathered element index (pix_y-t)*b_w+pix_x-l, or -1 if the
 * voxel projects outside the bbox (:36-37), or -2 - idx if the pixel there is
 * invalid (:40-41).  It lets tests compare pixel maps exactly.
 */
void tsdf_oracle_voxels(const float *depth, const int32_t *header, const float *ori,
                        float voxel_len, float trunc_dis, int R, const tsdf_cam *cam, int layout,
                        float *out, int32_t *pixmap) {
  if (!cam) cam = &k_default_cam;
  const int l = header[2], t = header[3], r = header[4], b = header[5];
  const int bw = r - l;
  const size_t R3 = (size_t)R * R * R;
  memset(out, 0, 3 * R3 * sizeof(float)); /* :33-35 */
  const double F = cam->focal;
  for (int z = 0; z < R; ++z) {
    const double v_z = (double)ori[2] + (double)z * (double)voxel_len; /* :28 */
    const double q = -F / v_z;                                         /* :30 */
    for (int y = 0; y < R; ++y) {
      const double v_y = (double)ori[1] + (double)y * (double)voxel_len; /* :27 */
      const double py_f = (-v_y * q) + cam->cy;                          /* :32, mul then add */
      const int32_t pix_y = trunc_i32(py_f);
      for (int x = 0; x < R; ++x) {
        const size_t vi = ((size_t)z * R + y) * R + x;
        const double v_x = (double)ori[0] + (double)x * (double)voxel_len; /* :26 */
        const double px_f = (v_x * q) + cam->cx;                           /* :31 */
        const int32_t pix_x = trunc_i32(px_f);
        if (pix_x < l || pix_x >= r || pix_y < t || pix_y >= b) {          /* :36 */
          if (pixmap) pixmap[vi] = -1;
          continue;
        }
        const int32_t idx = (pix_y - t) * bw + pix_x - l;                  /* :38 */
        const float pd = depth[idx];                                       /* :39 */
        if (!(fabsf(pd) >= cam->invalid_eps)) {                            /* :40; NaN is invalid */
          if (pixmap) pixmap[vi] = -2 - idx;
          continue;
        }
        if (pixmap) pixmap[vi] = idx;
        const double q2 = (double)pd / F;                                  /* :43 */
        const double w_x = ((double)pix_x - cam->cx) * q2;                 /* :44 */
        const double w_y = -((double)pix_y - cam->cy) * q2;                /* :45 */
        const double w_z = -(double)pd;                                    /* :46 */
        double ts[3];
        ts[0] = fabs(v_x - w_x) / (double)trunc_dis;                       /* :47 */
        ts[1] = fabs(v_y - w_y) / (double)trunc_dis;                       /* :48 */
        ts[2] = fabs(v_z - w_z) / (double)trunc_dis;                       /* :49 */
        const double dist = sqrt(ts[0] * ts[0] + ts[1] * ts[1] + ts[2] * ts[2]); /* :51-52 */
        if (dist > 1.0) ts[0] = ts[1] = ts[2] = 1.0;                       /* :54-57 */
        for (int a = 0; a < 3; ++a)
          if (1.0 < ts[a]) ts[a] = 1.0;                                    /* :58-60 min(ts,1) */
        if (w_z > v_z)                                                     /* :65 */
          for (int a = 0; a < 3; ++a) ts[a] = -ts[a];
        for (int a = 0; a < 3; ++a) {                                      /* :70-72 */
          size_t o;
          if (layout == TSDF_LAYOUT_CXYZ)
            o = (size_t)a * R3 + ((size_t)x * R + y) * R + z;              /* pre/tsdf_for.py:118-120 */
          else
            o = (size_t)a * R3 + vi;
          out[o] = (float)ts[a];
        }
      }
    }
  }
}

/* Degenerate-frame rule of include/tsdf.h applied to a placed grid: the extent must be positive and finite
 * and the centre finite (a +-inf depth passes |d| >= eps and poisons the AABB).  Returns the status and
 * clears what the header says is cleared. */
static int grid_status(float *grid) {
  const int ext_ok = grid[3] > 0.0f && isfinite(grid[3]);
  const int mid_ok = isfinite(grid[0]) && isfinite(grid[1]) && isfinite(grid[2]);
  if (ext_ok && mid_ok) return TSDF_FRAME_OK;
  grid[3] = grid[4] = grid[5] = 0.0f;
  /* a zero extent keeps its (finite) centre; a non-finite centre is reported as 0 */
  if (!mid_ok) grid[0] = grid[1] = grid[2] = 0.0f;
  return TSDF_FRAME_DEGENERATE;
}

/* One frame end to end: cal_tsdf_cuda (pre/tsdf_numba.py:119-161) with the
 * degenerate-frame convention of include/tsdf.h.  Returns the frame status. */
static int oracle_frame(const float *depth, int64_t n_elem, const int32_t *header, int R,
                        const tsdf_cam *cam, int layout, float *out, float *max_l, float *mid_p,
                        float *aabb, float *grid_out, float *ori_out) {
  const size_t R3 = (size_t)R * R * R;
  const int bw = header[4] - header[2], bh = header[5] - header[3];
  float grid[8] = {0}, ori[3] = {0}, mn[3] = {0}, mx[3] = {0};
  int status = TSDF_FRAME_OK;
  if (bw <= 0 || bh <= 0 || (int64_t)bw * bh != n_elem) {
    status = TSDF_FRAME_BAD_HEADER;
  } else {
    long nv = tsdf_oracle_aabb(depth, header, cam, mn, mx);
    if (nv == 0) {
      status = TSDF_FRAME_DEGENERATE;
    } else {
      tsdf_oracle_glue(mn, mx, R, cam, grid, ori);
      status = grid_status(grid);
    }
  }
  if (status == TSDF_FRAME_OK) {
    if (out) tsdf_oracle_voxels(depth, header, ori, grid[4], grid[5], R, cam, layout, out, NULL);
  } else {
    if (out) memset(out, 0, 3 * R3 * sizeof(float));
  }
  if (max_l) *max_l = grid[3];
  if (mid_p) { mid_p[0] = grid[0]; mid_p[1] = grid[1]; mid_p[2] = grid[2]; }
  if (aabb) { memcpy(aabb, mn, sizeof mn); memcpy(aabb + 3, mx, sizeof mx); }
  if (grid_out) memcpy(grid_out, grid, sizeof grid);
  if (ori_out) memcpy(ori_out, ori, sizeof ori);
  return status;
}

/*
 * Batch form with the same argument meaning as tsdf_voxelize_hip (include/tsdf.h),
 * host pointers, OpenMP over frames.  n_threads <= 0 means 1.  Any output may be NULL.
 * Returns the number of threads actually used.
 */
int tsdf_oracle_voxelize(const float *depth, const int64_t *offsets, const int32_t *headers, int n,
                         int R, const tsdf_cam *cam, int layout, int n_threads, float *out_tsdf,
                         float *out_max_l, float *out_mid_p, int32_t *out_status, float *out_aabb,
                         float *out_grid, float *out_ori) {
  if (!cam) cam = &k_default_cam;
  if (n_threads <= 0) n_threads = 1;
  const size_t vol = (size_t)3 * R * R * R;
  int used = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
#endif
  for (int i = 0; i < n; ++i) {
#ifdef _OPENMP
    if (i == 0) used = omp_get_num_threads();
#endif
    int st = oracle_frame(depth + offsets[i], offsets[i + 1] - offsets[i], headers + 6 * (size_t)i,
                          R, cam, layout, out_tsdf ? out_tsdf + vol * i : NULL,
                          out_max_l ? out_max_l + i : NULL, out_mid_p ? out_mid_p + 3 * (size_t)i : NULL,
                          out_aabb ? out_aabb + 6 * (size_t)i : NULL,
                          out_grid ? out_grid + 8 * (size_t)i : NULL,
                          out_ori ? out_ori + 3 * (size_t)i : NULL);
    if (out_status) out_status[i] = st;
  }
  return used;
}

/* ------------------------------------------------------------------------------------------------
 * Augmented form (BASELINE.json configs[4]).  The reference's data_aug
 * (pre/process.py:202-261) raises AxisError on its own input and, where it runs (cut_version), only
 * moves the grid while the TSDF still samples the un-augmented depth image (App. B#8-9), so there is
 * NO reference output to match: this is a re-specification, parity unpinned, checked by construction
 * (identity transform == plain path: grid, pixel map, zero mask, sign and z component bit for bit, x / y to the
 * float32 rounding) and against the HIP kernel.
 *
 * xf = double[24] per frame: forward affine T(p) = A p + b as rows {A_i0, A_i1, A_i2, b_i} (12 values),
 * then its inverse in the same form (12 values).  Every product/sum below is a separately rounded
 * float64 operation, in the order written (see affine3 / affine3_fwd).
 *   cloud:  p' = T(p) for the back-projected point p of every valid pixel (A.1, before the float32
 *           rounding), then rounded to float32 for the AABB -> glue as in the plain path;
 *   voxel:  centre v' lives in the augmented frame; v = T^-1(v') is projected (pre/tsdf_numba.py:30-32) and the
 *           pixel gathered and tested (:36-41) exactly as in the plain path.  The distances (:43-49) are those
 *           between v' and w' = T(w), w the surface point of that pixel, written out for an affine T so that
 *           the surface point itself never has to be formed: with dxi = pix_x - cx, dyi = pix_y - cy (:44-45),
 *           w = pd * (dxi/F, -dyi/F, -1) and therefore, per axis i,
 *               v'_i - w'_i = (v'_i - b_i) + pd * c_i,    c_i = -A_i0 dxi / F + A_i1 dyi / F + A_i2.
 *           Evaluated (float64, one rounding per operation, fma where written) as
 *               iF = 1 / F,  it = 1 / trunc_dis                                         (per frame)
 *               g_i0 = -(A_i0 * iF),  g_i1 = A_i1 * iF                                  (per frame)
 *               c_i = fma(g_i0, dxi, fma(g_i1, dyi, A_i2))
 *               u_i = fma(pd, c_i, v'_i - b_i)                                          (mm)
 *               t_i = u_i * it                                                          (:47-49, |.| taken below)
 *               near  iff  fma(t_z, t_z, fma(t_y, t_y, t_x * t_x)) <= 1                 (:51-57, without the sqrt)
 *               value_i = near ? min(|t_i|, 1) : 1 ;  negated iff u_z < 0               (:58-68: w'_z > v'_z)
 *           With the identity map c = (-dxi * iF, dyi * iF, 1): u_z = pd + v_z exactly as rounded once, so the
 *           sign and the z component are those of the plain path bit for bit; x and y differ from it in the
 *           association of one product (<= 2 ulp of float64, i.e. nothing after the float32 store except at
 *           the exact tie of a rounding).  Round 2 formed w and T(w) explicitly (about twice the float64
 *           operations per voxel), and the kernel that implemented it was bound by instruction issue — which is
 *           why the contract, a re-specification this project owns, is written in its cheapest exact form.
 *           (Since round 4 that kernel is bound by its volume stores first — without any per-voxel arithmetic it
 *           would be 11 % faster, DESIGN.md (d) —, as include/tsdf.h says too.)
 */
/* Inverse map (voxel centre back into the camera frame): every product and sum rounded separately, grouped as
 * (A_i0 x + A_i1 y) + (A_i2 z + b_i) — both brackets depend on grid indices only, so an implementation may
 * tabulate / hoist them and pay one addition per voxel and row.  The identity map returns p exactly. */
static void affine3(const double *m, const double *p, double *o) {
  for (int i = 0; i < 3; ++i) {
    volatile double a = m[4 * i] * p[0], b = m[4 * i + 1] * p[1], c = m[4 * i + 2] * p[2];
    volatile double ab = a + b, cd = c + m[4 * i + 3];
    o[i] = ab + cd;
  }
}

/* Forward map (camera-frame point into the augmented frame): a fused chain, one rounding per step.  With the
 * identity row {1,0,0,0} it returns p[i] exactly, which is what makes the identity map equal the plain path. */
static void affine3_fwd(const double *m, const double *p, double *o) {
  for (int i = 0; i < 3; ++i) o[i] = fma(m[4 * i], p[0], fma(m[4 * i + 1], p[1], fma(m[4 * i + 2], p[2], m[4 * i + 3])));
}

long tsdf_oracle_aabb_aug(const float *depth, const int32_t *header, const tsdf_cam *cam, const double *xf,
                          float *min_p, float *max_p) {
  if (!cam) cam = &k_default_cam;
  const int l = header[2], t = header[3], r = header[4], b = header[5];
  const int bw = r - l, bh = b - t;
  long n_valid = 0;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int row = 0; row < bh; ++row)
    for (int col = 0; col < bw; ++col) {
      const float d = depth[(long)row * bw + col];
      if (!(fabsf(d) >= cam->invalid_eps)) continue;
      const double q = (double)d / cam->focal;
      const double p[3] = {q * ((double)(col + l) - cam->cx), -q * ((double)(row + t) - cam->cy), -(double)d};
      double o[3];
      affine3_fwd(xf, p, o);
      for (int a = 0; a < 3; ++a) {
        const float v = (float)o[a];
        if (v < mn[a]) mn[a] = v;
        if (v > mx[a]) mx[a] = v;
      }
      ++n_valid;
    }
  if (n_valid) {
    memcpy(min_p, mn, sizeof mn);
    memcpy(max_p, mx, sizeof mx);
  }
  return n_valid;
}

void tsdf_oracle_voxels_aug(const float *depth, const int32_t *header, const float *ori, float voxel_len,
                            float trunc_dis, int R, const tsdf_cam *cam, int layout, const double *xf,
                            float *out) {
  if (!cam) cam = &k_default_cam;
  const int l = header[2], t = header[3], r = header[4], b = header[5];
  const int bw = r - l;
  const size_t R3 = (size_t)R * R * R;
  memset(out, 0, 3 * R3 * sizeof(float));
  const double F = cam->focal;
  const double *fwd = xf, *inv = xf + 12;
  /* per-frame constants of the distance terms (see the contract above) */
  const double iF = 1.0 / F, it = 1.0 / (double)trunc_dis;
  double g0[3], g1[3];
  for (int a = 0; a < 3; ++a) {
    volatile double p0 = fwd[4 * a] * iF, p1 = fwd[4 * a + 1] * iF;
    g0[a] = -p0;
    g1[a] = p1;
  }
  for (int z = 0; z < R; ++z)
    for (int y = 0; y < R; ++y)
      for (int x = 0; x < R; ++x) {
        const double vp[3] = {(double)ori[0] + (double)x * (double)voxel_len,
                              (double)ori[1] + (double)y * (double)voxel_len,
                              (double)ori[2] + (double)z * (double)voxel_len};
        double v[3];
        affine3(inv, vp, v);
        const double q = -F / v[2];
        const int32_t pix_x = trunc_i32((v[0] * q) + cam->cx);
        const int32_t pix_y = trunc_i32((-v[1] * q) + cam->cy);
        if (pix_x < l || pix_x >= r || pix_y < t || pix_y >= b) continue;
        const float pd = depth[(pix_y - t) * bw + pix_x - l];
        if (!(fabsf(pd) >= cam->invalid_eps)) continue;
        const double dxi = (double)pix_x - cam->cx, dyi = (double)pix_y - cam->cy;
        double u[3], ts[3];
        for (int a = 0; a < 3; ++a) {
          const double c = fma(g0[a], dxi, fma(g1[a], dyi, fwd[4 * a + 2]));
          volatile double vb = vp[a] - fwd[4 * a + 3];
          u[a] = fma((double)pd, c, vb);                                  /* v'_a - w'_a, mm */
          volatile double ta = u[a] * it;
          ts[a] = ta;
        }
        volatile double xx = ts[0] * ts[0];
        const double s2 = fma(ts[2], ts[2], fma(ts[1], ts[1], xx));
        const int nearv = s2 <= 1.0;
        for (int a = 0; a < 3; ++a) {
          double m = fabs(ts[a]);
          if (!(m < 1.0)) m = 1.0;             /* min(|t|, 1); a NaN distance counts as far */
          ts[a] = nearv ? m : 1.0;
        }
        if (u[2] < 0.0)
          for (int a = 0; a < 3; ++a) ts[a] = -ts[a];
        const size_t vi = ((size_t)z * R + y) * R + x;
        for (int a = 0; a < 3; ++a) {
          const size_t o = layout == TSDF_LAYOUT_CXYZ ? (size_t)a * R3 + ((size_t)x * R + y) * R + z
                                                      : (size_t)a * R3 + vi;
          out[o] = (float)ts[a];
        }
      }
}

int tsdf_oracle_voxelize_aug(const float *depth, const int64_t *offsets, const int32_t *headers, int n, int R,
                             const tsdf_cam *cam, int layout, int n_threads, const double *xforms,
                             float *out_tsdf, float *out_max_l, float *out_mid_p, int32_t *out_status) {
  if (!cam) cam = &k_default_cam;
  if (n_threads <= 0) n_threads = 1;
  const size_t vol = (size_t)3 * R * R * R;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
#endif
  for (int i = 0; i < n; ++i) {
    const float *d = depth + offsets[i];
    const int32_t *h = headers + 6 * (size_t)i;
    const double *xf = xforms + 24 * (size_t)i;
    const int bw = h[4] - h[2], bh = h[5] - h[3];
    float grid[8] = {0}, ori[3] = {0}, mn[3] = {0}, mx[3] = {0};
    int status = TSDF_FRAME_OK;
    if (bw <= 0 || bh <= 0 || (int64_t)bw * bh != offsets[i + 1] - offsets[i]) {
      status = TSDF_FRAME_BAD_HEADER;
    } else if (tsdf_oracle_aabb_aug(d, h, cam, xf, mn, mx) == 0) {
      status = TSDF_FRAME_DEGENERATE;
    } else {
      tsdf_oracle_glue(mn, mx, R, cam, grid, ori);
      status = grid_status(grid);
    }
    if (out_tsdf) {
      if (status == TSDF_FRAME_OK)
        tsdf_oracle_voxels_aug(d, h, ori, grid[4], grid[5], R, cam, layout, xf, out_tsdf + vol * i);
      else
        memset(out_tsdf + vol * i, 0, vol * sizeof(float));
    }
    if (out_max_l) out_max_l[i] = grid[3];
    if (out_mid_p) memcpy(out_mid_p + 3 * (size_t)i, grid, 3 * sizeof(float));
    if (out_status) out_status[i] = status;
  }
  return n_threads;
}

/* ------------------------------------------------------------------------------------------------
 * Label normalisation — pre/joint_nor.py:8-18 and the per-sample loop of 3D_CNN/train.py:236-244:
 *   joint_nor = (joint - mid_p) / max_l + 0.5      float32, three separately rounded operations
 *   joint_nor[joint_nor < 0] = 0 ; joint_nor[joint_nor > 1] = 1      (train.py:241-242; NaN stays NaN)
 * gt float32[n][3*J] (x,y,z per joint), out the same shape.  A frame whose status is not OK (max_l == 0;
 * the reference returns None for it, tsdf_numba.py:162-171) gets 0.5 everywhere — the cube centre —
 * instead of the reference's division by zero (include/tsdf.h).
 */
void tsdf_oracle_normalize_joints(const float *gt, const float *max_l, const float *mid_p, int n, int J,
                                  int clamp, float *out) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < 3 * J; ++j) {
      const size_t e = (size_t)i * 3 * J + j;
      if (!(max_l[i] > 0.0f)) { out[e] = 0.5f; continue; }
      volatile float a = gt[e] - mid_p[3 * (size_t)i + j % 3];
      volatile float b = a / max_l[i];
      float v = b + 0.5f;
      if (clamp) {
        if (v < 0.0f) v = 0.0f;
        if (v > 1.0f) v = 1.0f;
      }
      out[e] = v;
    }
}

/* Augmented labels (pre/process.py:232-249 maps the joints with the cloud's own S and R): T(joint) with the
 * forward map of the frame's xform, fused chain in float64 (affine3_fwd), rounded to float32. */
void tsdf_oracle_transform_joints(const float *gt, const double *xforms, int n, int J, float *out) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < J; ++j) {
      const float *g = gt + ((size_t)i * J + j) * 3;
      const double p[3] = {(double)g[0], (double)g[1], (double)g[2]};
      double o[3];
      affine3_fwd(xforms + 24 * (size_t)i, p, o);
      for (int a = 0; a < 3; ++a) out[((size_t)i * J + j) * 3 + a] = (float)o[a];
    }
}
