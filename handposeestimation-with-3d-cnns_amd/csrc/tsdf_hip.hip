// tsdf_hip.hip — fused projective-TSDF voxelizer for gfx950 (MI355X), and its C ABI.
//
// Replaces, for a whole batch in ONE launch, what the reference does per frame with two
// numba kernels, host numpy glue and four PCIe copies (pre/tsdf_numba.py:119-161):
//   phase 1  min_max_kernel  (pre/tsdf_numba.py:75-116,140-141)  AABB of all valid pixels
//   glue     host numpy      (pre/tsdf_numba.py:142-147)          grid placement, float32
//   phase 2  tsdf_kernel     (pre/tsdf_numba.py:15-72)            per-voxel project/gather/TSDF
// Arithmetic contract: SURVEY.md Appendix A (float32 parameters, float64 intermediates,
// unfused multiply-then-add for the pixel index, float32 store).
//
// Design (DESIGN.md has the numbers):
//   * one 256-thread workgroup (4 wave64) per frame; the AABB never leaves the chip;
//   * phase 1 streams the crop once with 16-byte loads, lane <-> 4 consecutive columns,
//     wave <-> rows.  It does NOT back-project every pixel (one float64 division each):
//     f32(f64(d)/F * (x-cx)) is monotone in d for a fixed column x (and likewise per row),
//     so the AABB is the extreme of the formula applied to each column's / row's (min,max)
//     valid depth — 2(b_w+b_h) divisions per wave instead of b_w*b_h, bit-identical result;
//   * phase 2: each lane owns 4 consecutive voxels along the layout's fastest axis, so every
//     wave store is 1 KiB contiguous (global_store_dwordx4); q = -F/v_z comes from a 1-per-z
//     LDS table (true division), the per-voxel chain uses reciprocals (<= 2 ulp64 from the
//     divisions it replaces — 9 orders of magnitude inside the 1e-5 parity bound);
//   * memory-bound streaming read + streaming write; the gather re-reads lines the same
//     workgroup has just streamed (L2 / Infinity Cache), counted once in the roofline.
// No MFMA (gather/scatter, not a contraction), no CPU fallback, gfx950 only.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tsdf.h"

namespace {

constexpr int kWG = 256;            // threads per workgroup
constexpr int kWaves = kWG / 64;    // wave64
constexpr int kRowUnroll = 4;       // rows in flight per wave in phase 1
constexpr int kMaxR = 128;

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned 16-B access

struct CamK {
  double focal, cx, cy, inv_focal;
  float eps, trunc_vox;
};

#define TSDF_INF __builtin_inff()

// ---- wave64 reductions on DPP (result valid in every lane after the readlane) -------------
template <int Ctrl>
__device__ __forceinline__ float dpp_get(float v) {
  // lanes whose DPP source is out of range keep their own value ("old" = v)
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), Ctrl, 0xf,
                                                    0xf, false));
}

__device__ __forceinline__ float wave_min(float v) {
  v = fminf(v, dpp_get<0x111>(v));  // row_shr:1
  v = fminf(v, dpp_get<0x112>(v));  // row_shr:2
  v = fminf(v, dpp_get<0x114>(v));  // row_shr:4
  v = fminf(v, dpp_get<0x118>(v));  // row_shr:8   -> lane 15 of each row holds the row's min
  v = fminf(v, dpp_get<0x142>(v));  // row_bcast:15
  v = fminf(v, dpp_get<0x143>(v));  // row_bcast:31 -> lane 63 holds the wave's min
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_get<0x111>(v));
  v = fmaxf(v, dpp_get<0x112>(v));
  v = fmaxf(v, dpp_get<0x114>(v));
  v = fmaxf(v, dpp_get<0x118>(v));
  v = fmaxf(v, dpp_get<0x142>(v));
  v = fmaxf(v, dpp_get<0x143>(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// ---- exact float64 helpers (never contracted) ----------------------------------------------
// int() of a float64, toward zero; v_cvt_i32_f64 saturates out-of-range values and maps NaN to 0
// (same rule as oracle/tsdf_oracle.c::trunc_i32).
__device__ __forceinline__ int trunc_i32(double v) {
  int r;
  asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(v));
  return r;
}

// (a * b) + c with two roundings: pre/tsdf_numba.py:31-32 as numba types it (App. A.3).
__device__ __forceinline__ double mul_then_add(double a, double b, double c) {
#pragma clang fp contract(off)
  double p = a * b;
  return p + c;
}

// A.1 x: f32( (f64(d)/F) * (x - cx) )      pre/tsdf_numba.py:91-92,95
__device__ __forceinline__ float backproject_x(float d, int x, const CamK &k) {
  const double q = (double)d / k.focal;
  return (float)(q * ((double)x - k.cx));
}
// A.1 y: f32( (-(f64(d)/F)) * (y - cy) )   pre/tsdf_numba.py:91,93,95
__device__ __forceinline__ float backproject_y(float d, int y, const CamK &k) {
  const double q = (double)d / k.focal;
  return (float)((-q) * ((double)y - k.cy));
}

// 4 consecutive pixels of one row starting at column c; columns >= bw read as NaN (never valid).
__device__ __forceinline__ f4 load_row4(const float *__restrict__ rp, int c, int bw) {
  const float nan = __builtin_nanf("");
  f4 v = {nan, nan, nan, nan};
  if (c + 3 < bw) {
    v = *reinterpret_cast<const f4u *>(rp + c);
  } else {
    if (c < bw) v.x = rp[c];
    if (c + 1 < bw) v.y = rp[c + 1];
    if (c + 2 < bw) v.z = rp[c + 2];
  }
  return v;
}

__device__ __forceinline__ void acc1(float d, float eps, float &cmin, float &cmax, float &rmin,
                                     float &rmax) {
  const bool ok = __builtin_fabsf(d) >= eps;  // NaN -> invalid
  const float lo = ok ? d : TSDF_INF;
  const float hi = ok ? d : -TSDF_INF;
  cmin = fminf(cmin, lo);
  cmax = fmaxf(cmax, hi);
  rmin = fminf(rmin, lo);
  rmax = fmaxf(rmax, hi);
}

__device__ __forceinline__ void acc4(f4 v, float eps, float (&cmin)[4], float (&cmax)[4], float &rmin,
                                     float &rmax) {
  acc1(v.x, eps, cmin[0], cmax[0], rmin, rmax);
  acc1(v.y, eps, cmin[1], cmax[1], rmin, rmax);
  acc1(v.z, eps, cmin[2], cmax[2], rmin, rmax);
  acc1(v.w, eps, cmin[3], cmax[3], rmin, rmax);
}

struct Frame {
  const float *depth;  // frame-local base
  int l, t, r, b, bw, bh;
};

struct Grid {
  float mid[3];
  float max_l, voxel_len, trunc;
  float ori[3];
};

// ---- phase 1: AABB of all valid back-projected pixels ---------------------------------------
// Returns (in every thread) min/max xyz; `any` is false when the frame has no valid pixel.
__device__ __forceinline__ void phase1_aabb(const Frame &f, const CamK &k, float (&red)[kWaves][8],
                                            float (&mn)[3], float (&mx)[3], bool &any) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float xmn = TSDF_INF, xmx = -TSDF_INF, ymn = TSDF_INF, ymx = -TSDF_INF;
  float dmn = TSDF_INF, dmx = -TSDF_INF;
  // per-wave stash of reduced row extremes: lane i keeps the i-th non-empty row piece
  float s_rmin = TSDF_INF, s_rmax = -TSDF_INF;
  int s_row = 0, cnt = 0;

  auto flush_rows = [&]() {
    if (s_rmin <= s_rmax) {
      const int y = f.t + s_row;
      const float a = backproject_y(s_rmin, y, k), b = backproject_y(s_rmax, y, k);
      ymn = fminf(ymn, fminf(a, b));
      ymx = fmaxf(ymx, fmaxf(a, b));
    }
    s_rmin = TSDF_INF;
    s_rmax = -TSDF_INF;
    cnt = 0;
  };

  for (int cbase = 0; cbase < f.bw; cbase += 512) {
    float cmin[2][4], cmax[2][4];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        cmin[kk][j] = TSDF_INF;
        cmax[kk][j] = -TSDF_INF;
      }
    const int c0 = cbase + 4 * lane, c1 = c0 + 256;
    const bool has1 = cbase + 256 < f.bw;  // wave-uniform

    for (int row0 = wave; row0 < f.bh; row0 += kWaves * kRowUnroll) {
      f4 va[kRowUnroll], vb[kRowUnroll];
      const float nan = __builtin_nanf("");
#pragma unroll
      for (int u = 0; u < kRowUnroll; ++u) {
        const int row = row0 + kWaves * u;
        va[u] = f4{nan, nan, nan, nan};
        vb[u] = f4{nan, nan, nan, nan};
        if (row < f.bh) {
          const float *rp = f.depth + (int64_t)row * f.bw;
          va[u] = load_row4(rp, c0, f.bw);
          if (has1) vb[u] = load_row4(rp, c1, f.bw);
        }
      }
      if (cnt > 64 - kRowUnroll) flush_rows();  // wave-uniform (cnt is)
#pragma unroll
      for (int u = 0; u < kRowUnroll; ++u) {
        const int row = row0 + kWaves * u;
        float rmin = TSDF_INF, rmax = -TSDF_INF;
        acc4(va[u], k.eps, cmin[0], cmax[0], rmin, rmax);
        if (has1) acc4(vb[u], k.eps, cmin[1], cmax[1], rmin, rmax);
        if (__any(rmin <= rmax)) {  // skip the reduction for rows without a valid pixel
          const float wmin = wave_min(rmin), wmax = wave_max(rmax);
          if (lane == cnt) {
            s_rmin = wmin;
            s_rmax = wmax;
            s_row = row;
          }
          ++cnt;
        }
      }
    }
    // column extremes of this wave's rows -> x extent, depth extremes -> z extent
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (cmin[kk][j] <= cmax[kk][j]) {
          const int x = f.l + c0 + 256 * kk + j;
          const float a = backproject_x(cmin[kk][j], x, k), b = backproject_x(cmax[kk][j], x, k);
          xmn = fminf(xmn, fminf(a, b));
          xmx = fmaxf(xmx, fmaxf(a, b));
          dmn = fminf(dmn, cmin[kk][j]);
          dmx = fmaxf(dmx, cmax[kk][j]);
        }
      }
  }
  flush_rows();

  xmn = wave_min(xmn);
  ymn = wave_min(ymn);
  dmn = wave_min(dmn);
  xmx = wave_max(xmx);
  ymx = wave_max(ymx);
  dmx = wave_max(dmx);
  if (lane == 0) {
    red[wave][0] = xmn;
    red[wave][1] = ymn;
    red[wave][2] = dmn;
    red[wave][3] = xmx;
    red[wave][4] = ymx;
    red[wave][5] = dmx;
  }
  __syncthreads();
  xmn = ymn = dmn = TSDF_INF;
  xmx = ymx = dmx = -TSDF_INF;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) {
    xmn = fminf(xmn, red[w][0]);
    ymn = fminf(ymn, red[w][1]);
    dmn = fminf(dmn, red[w][2]);
    xmx = fmaxf(xmx, red[w][3]);
    ymx = fmaxf(ymx, red[w][4]);
    dmx = fmaxf(dmx, red[w][5]);
  }
  any = dmn <= dmx;
  mn[0] = xmn;
  mn[1] = ymn;
  mn[2] = -dmx;  // cam_z = -d   pre/tsdf_numba.py:94
  mx[0] = xmx;
  mx[1] = ymx;
  mx[2] = -dmn;
}

// ---- glue: pre/tsdf_numba.py:142-147, float32, left to right --------------------------------
__device__ __forceinline__ Grid glue(const float (&mn)[3], const float (&mx)[3], int R, const CamK &k) {
  Grid g;
  float len[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    g.mid[a] = __fdiv_rn(__fadd_rn(mn[a], mx[a]), 2.0f);
    len[a] = __fsub_rn(mx[a], mn[a]);
  }
  g.max_l = fmaxf(len[0], fmaxf(len[1], len[2]));
  g.voxel_len = __fdiv_rn(g.max_l, (float)R);
  g.trunc = __fmul_rn(g.voxel_len, k.trunc_vox);
#pragma unroll
  for (int a = 0; a < 3; ++a)
    g.ori[a] = __fadd_rn(__fsub_rn(g.mid[a], __fdiv_rn(g.max_l, 2.0f)), __fdiv_rn(g.voxel_len, 2.0f));
  return g;
}

// ---- phase 2 per-voxel value: pre/tsdf_numba.py:43-68 ---------------------------------------
struct VoxK {
  double cx, cy, inv_focal, inv_trunc;
  float eps;
};

__device__ __forceinline__ void tsdf_value(double vx, double vy, double vz, int pix_x, int pix_y, float pd,
                                           bool inb, const VoxK &k, float &ox, float &oy, float &oz) {
  const bool ok = inb && (__builtin_fabsf(pd) >= k.eps);         // :36,:40
  const double pd64 = (double)pd;
  const double q2 = pd64 * k.inv_focal;                           // :43 (reciprocal form)
  const double wx = ((double)pix_x - k.cx) * q2;                  // :44
  const double wy = -((double)pix_y - k.cy) * q2;                 // :45
  const double tx = __builtin_fabs(vx - wx) * k.inv_trunc;        // :47
  const double ty = __builtin_fabs(vy - wy) * k.inv_trunc;        // :48
  const double tz = __builtin_fabs(vz + pd64) * k.inv_trunc;      // :49, w_z = -pd :46
  const double s = __builtin_fma(tz, tz, __builtin_fma(ty, ty, tx * tx));  // dist^2 :51-52
  const bool far = s > 1.0;                                       // :54  (sqrt is monotone, sqrt(1)=1)
  float fx = (far || tx > 1.0) ? 1.0f : (float)tx;                // :55-60, f32 store :70-72
  float fy = (far || ty > 1.0) ? 1.0f : (float)ty;
  float fz = (far || tz > 1.0) ? 1.0f : (float)tz;
  const bool neg = (-pd64) > vz;                                  // :65
  fx = neg ? -fx : fx;
  fy = neg ? -fy : fy;
  fz = neg ? -fz : fz;
  ox = ok ? fx : 0.0f;                                            // :33-41
  oy = ok ? fy : 0.0f;
  oz = ok ? fz : 0.0f;
}

__device__ __forceinline__ void zero_volume(float *__restrict__ out, int R) {
  const int n4 = 3 * R * R * R / 4;
  f4 *o4 = reinterpret_cast<f4 *>(out);
  const f4 z = {0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < n4; i += kWG) o4[i] = z;
}

// LAYOUT 0: o[c][z][y][x], lanes run along x.   LAYOUT 1: o[c][x][y][z], lanes run along z.
template <int RT, int LAYOUT>
__global__ __launch_bounds__(kWG) void tsdf_fused_kernel(
    const float *__restrict__ depth, const int64_t *__restrict__ offsets,
    const int32_t *__restrict__ headers, int n, int Rrt, CamK cam, float *__restrict__ out_tsdf,
    float *__restrict__ out_max_l, float *__restrict__ out_mid_p, int32_t *__restrict__ out_status,
    float *__restrict__ out_aabb, float *__restrict__ out_grid, float *__restrict__ out_ori,
    int aabb_only) {
  __shared__ float red[kWaves][8];
  __shared__ double qtab[kMaxR];

  const int R = RT ? RT : Rrt;
  const int frame = blockIdx.x;
  if (frame >= n) return;
  const int tid = threadIdx.x;

  const int32_t *h = headers + 6 * (int64_t)frame;
  Frame f;
  f.l = h[2];
  f.t = h[3];
  f.r = h[4];
  f.b = h[5];
  f.bw = f.r - f.l;
  f.bh = f.b - f.t;
  const int64_t off0 = offsets[frame], off1 = offsets[frame + 1];
  f.depth = depth + off0;
  float *out = out_tsdf ? out_tsdf + (int64_t)frame * 3 * R * R * R : nullptr;

  int status = TSDF_FRAME_OK;
  float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
  Grid g;
  g.mid[0] = g.mid[1] = g.mid[2] = 0.f;
  g.max_l = g.voxel_len = g.trunc = 0.f;
  g.ori[0] = g.ori[1] = g.ori[2] = 0.f;

  if (f.bw <= 0 || f.bh <= 0 || (int64_t)f.bw * (int64_t)f.bh != off1 - off0) {
    status = TSDF_FRAME_BAD_HEADER;  // block-uniform
  } else {
    bool any;
    phase1_aabb(f, cam, red, mn, mx, any);
    if (!any) {
      status = TSDF_FRAME_DEGENERATE;
      mn[0] = mn[1] = mn[2] = mx[0] = mx[1] = mx[2] = 0.f;
    } else {
      g = glue(mn, mx, R, cam);
      if (!(g.max_l > 0.f) || !(g.max_l < TSDF_INF)) {
        status = TSDF_FRAME_DEGENERATE;
        g.max_l = g.voxel_len = g.trunc = 0.f;
      }
    }
  }

  if (tid == 0) {
    if (out_max_l) out_max_l[frame] = g.max_l;
    if (out_mid_p) {
      out_mid_p[3 * (int64_t)frame + 0] = g.mid[0];
      out_mid_p[3 * (int64_t)frame + 1] = g.mid[1];
      out_mid_p[3 * (int64_t)frame + 2] = g.mid[2];
    }
    if (out_status) out_status[frame] = status;
    if (out_aabb) {
      float *a = out_aabb + 6 * (int64_t)frame;
      a[0] = mn[0]; a[1] = mn[1]; a[2] = mn[2];
      a[3] = mx[0]; a[4] = mx[1]; a[5] = mx[2];
    }
    if (out_grid) {
      float *q = out_grid + 8 * (int64_t)frame;
      q[0] = g.mid[0]; q[1] = g.mid[1]; q[2] = g.mid[2];
      q[3] = g.max_l; q[4] = g.voxel_len; q[5] = g.trunc; q[6] = 0.f; q[7] = 0.f;
    }
    if (out_ori) {
      float *q = out_ori + 3 * (int64_t)frame;
      q[0] = g.ori[0]; q[1] = g.ori[1]; q[2] = g.ori[2];
    }
  }
  if (aabb_only || !out) return;
  if (status != TSDF_FRAME_OK) {
    zero_volume(out, R);
    return;
  }

  // ---- phase 2 ----
  const double vl = (double)g.voxel_len;
  const double ox = (double)g.ori[0], oy = (double)g.ori[1], oz = (double)g.ori[2];
  if (tid < R) {
    const double v_z = oz + (double)tid * vl;  // :28
    qtab[tid] = -cam.focal / v_z;              // :30, true division, once per z
  }
  __syncthreads();

  VoxK vk;
  vk.cx = cam.cx;
  vk.cy = cam.cy;
  vk.inv_focal = cam.inv_focal;
  vk.inv_trunc = 1.0 / (double)g.trunc;
  vk.eps = cam.eps;

  const int R4 = R / 4;
  const int groups = R * R4;          // (y, fast/4) pairs per slow-axis slice
  const int64_t R3 = (int64_t)R * R * R;
  const float *__restrict__ fd = f.depth;

  for (int gi = tid; gi < groups; gi += kWG) {
    const int f4i = (gi % R4) * 4;    // first of 4 consecutive indices on the fast axis
    const int y = gi / R4;
    const double vy = oy + (double)y * vl;  // :27
    const double nvy = -vy;

    if constexpr (LAYOUT == 0) {
      // fast axis = x.  v_x fixed per lane; loop over z.
      double vx[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) vx[j] = ox + (double)(f4i + j) * vl;  // :26
      for (int z = 0; z < R; ++z) {
        const double vz = oz + (double)z * vl;
        const double q = qtab[z];
        const int pix_y = trunc_i32(mul_then_add(nvy, q, cam.cy));      // :32
        const bool row_ok = pix_y >= f.t && pix_y < f.b;
        const int rowoff = (pix_y - f.t) * f.bw - f.l;
        int pix_x[4];
        bool inb[4];
        float pd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pix_x[j] = trunc_i32(mul_then_add(vx[j], q, cam.cx));         // :31
          inb[j] = row_ok && pix_x[j] >= f.l && pix_x[j] < f.r;         // :36
          pd[j] = fd[inb[j] ? rowoff + pix_x[j] : 0];                   // :38-39
        }
        f4 o0, o1, o2;
        { float a_, b_, c_; tsdf_value(vx[0], vy, vz, pix_x[0], pix_y, pd[0], inb[0], vk, a_, b_, c_); o0.x = a_; o1.x = b_; o2.x = c_; }
        { float a_, b_, c_; tsdf_value(vx[1], vy, vz, pix_x[1], pix_y, pd[1], inb[1], vk, a_, b_, c_); o0.y = a_; o1.y = b_; o2.y = c_; }
        { float a_, b_, c_; tsdf_value(vx[2], vy, vz, pix_x[2], pix_y, pd[2], inb[2], vk, a_, b_, c_); o0.z = a_; o1.z = b_; o2.z = c_; }
        { float a_, b_, c_; tsdf_value(vx[3], vy, vz, pix_x[3], pix_y, pd[3], inb[3], vk, a_, b_, c_); o0.w = a_; o1.w = b_; o2.w = c_; }
        const int64_t e = ((int64_t)z * R + y) * R + f4i;               // o[c][z][y][x] :70-72
        *reinterpret_cast<f4 *>(out + e) = o0;
        *reinterpret_cast<f4 *>(out + R3 + e) = o1;
        *reinterpret_cast<f4 *>(out + 2 * R3 + e) = o2;
      }
    } else {
      // fast axis = z.  q, v_z and pix_y fixed per lane; loop over x.
      double vz[4], q[4];
      int pix_y[4], rowoff[4];
      bool row_ok[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        vz[j] = oz + (double)(f4i + j) * vl;
        q[j] = qtab[f4i + j];
        pix_y[j] = trunc_i32(mul_then_add(nvy, q[j], cam.cy));
        row_ok[j] = pix_y[j] >= f.t && pix_y[j] < f.b;
        rowoff[j] = (pix_y[j] - f.t) * f.bw - f.l;
      }
      for (int x = 0; x < R; ++x) {
        const double vx = ox + (double)x * vl;
        int pix_x[4];
        bool inb[4];
        float pd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pix_x[j] = trunc_i32(mul_then_add(vx, q[j], cam.cx));
          inb[j] = row_ok[j] && pix_x[j] >= f.l && pix_x[j] < f.r;
          pd[j] = fd[inb[j] ? rowoff[j] + pix_x[j] : 0];
        }
        f4 o0, o1, o2;
        { float a_, b_, c_; tsdf_value(vx, vy, vz[0], pix_x[0], pix_y[0], pd[0], inb[0], vk, a_, b_, c_); o0.x = a_; o1.x = b_; o2.x = c_; }
        { float a_, b_, c_; tsdf_value(vx, vy, vz[1], pix_x[1], pix_y[1], pd[1], inb[1], vk, a_, b_, c_); o0.y = a_; o1.y = b_; o2.y = c_; }
        { float a_, b_, c_; tsdf_value(vx, vy, vz[2], pix_x[2], pix_y[2], pd[2], inb[2], vk, a_, b_, c_); o0.z = a_; o1.z = b_; o2.z = c_; }
        { float a_, b_, c_; tsdf_value(vx, vy, vz[3], pix_x[3], pix_y[3], pd[3], inb[3], vk, a_, b_, c_); o0.w = a_; o1.w = b_; o2.w = c_; }
        const int64_t e = ((int64_t)x * R + y) * R + f4i;               // o[c][x][y][z] tsdf_for.py:118-120
        *reinterpret_cast<f4 *>(out + e) = o0;
        *reinterpret_cast<f4 *>(out + R3 + e) = o1;
        *reinterpret_cast<f4 *>(out + 2 * R3 + e) = o2;
      }
    }
  }
}

const tsdf_cam kDefaultCam = {241.42, 160.0, 120.0, 1.0f, 3.0f};

int check_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return TSDF_ERR_NO_DEVICE;
  }
  return TSDF_OK;
}

template <int RT, int LAYOUT>
hipError_t launch(hipStream_t s, const float *d, const int64_t *o, const int32_t *h, int n, int R, CamK ck,
                  float *t, float *ml, float *mp, int32_t *st, float *ab, float *gr, float *orr,
                  int aabb_only) {
  hipLaunchKernelGGL((tsdf_fused_kernel<RT, LAYOUT>), dim3(n), dim3(kWG), 0, s, d, o, h, n, R, ck, t, ml,
                     mp, st, ab, gr, orr, aabb_only);
  return hipGetLastError();
}

int run(const float *d_depth, const int64_t *d_offsets, const int32_t *d_headers, int n, int R,
        const tsdf_cam *cam, int layout, void *hip_stream, float *t, float *ml, float *mp, int32_t *st,
        float *ab, float *gr, float *orr, int aabb_only) {
  if (n < 0 || !tsdf_resolution_supported(R)) return TSDF_ERR_INVALID_ARG;
  if (layout != TSDF_LAYOUT_CZYX && layout != TSDF_LAYOUT_CXYZ) return TSDF_ERR_INVALID_ARG;
  if (n == 0) return TSDF_OK;
  if (!d_depth || !d_offsets || !d_headers) return TSDF_ERR_INVALID_ARG;
  if (!aabb_only && (!t || (reinterpret_cast<uintptr_t>(t) & 15))) return TSDF_ERR_INVALID_ARG;
  if (!cam) cam = &kDefaultCam;
  if (!(cam->focal > 0.0) || !(cam->invalid_eps > 0.0f) || !(cam->trunc_voxels > 0.0f))
    return TSDF_ERR_INVALID_ARG;
  int rc = check_device();
  if (rc != TSDF_OK) return rc;
  CamK ck;
  ck.focal = cam->focal;
  ck.cx = cam->cx;
  ck.cy = cam->cy;
  ck.inv_focal = 1.0 / cam->focal;
  ck.eps = cam->invalid_eps;
  ck.trunc_vox = cam->trunc_voxels;
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  hipError_t e;
  if (layout == TSDF_LAYOUT_CZYX) {
    if (R == 32) e = launch<32, 0>(s, d_depth, d_offsets, d_headers, n, R, ck, t, ml, mp, st, ab, gr, orr, aabb_only);
    else if (R == 64) e = launch<64, 0>(s, d_depth, d_offsets, d_headers, n, R, ck, t, ml, mp, st, ab, gr, orr, aabb_only);
    else e = launch<0, 0>(s, d_depth, d_offsets, d_headers, n, R, ck, t, ml, mp, st, ab, gr, orr, aabb_only);
  } else {
    if (R == 32) e = launch<32, 1>(s, d_depth, d_offsets, d_headers, n, R, ck, t, ml, mp, st, ab, gr, orr, aabb_only);
    else if (R == 64) e = launch<64, 1>(s, d_depth, d_offsets, d_headers, n, R, ck, t, ml, mp, st, ab, gr, orr, aabb_only);
    else e = launch<0, 1>(s, d_depth, d_offsets, d_headers, n, R, ck, t, ml, mp, st, ab, gr, orr, aabb_only);
  }
  return e == hipSuccess ? TSDF_OK : TSDF_ERR_LAUNCH;
}

}  // namespace

extern "C" {

void tsdf_default_cam(tsdf_cam *cam) {
  if (cam) *cam = kDefaultCam;
}

int tsdf_version(void) { return TSDF_ABI_VERSION; }

const char *tsdf_strerror(int status) {
  switch (status) {
    case TSDF_OK: return "ok";
    case TSDF_ERR_INVALID_ARG: return "invalid argument";
    case TSDF_ERR_NO_DEVICE: return "no usable HIP device (this library is gfx950-only and has no CPU fallback)";
    case TSDF_ERR_LAUNCH: return "HIP kernel launch failed";
    default: return "unknown tsdf status";
  }
}

int tsdf_resolution_supported(int R) { return R >= 4 && R <= kMaxR && (R % 4) == 0; }

int tsdf_voxelize_hip(const float *d_depth, const int64_t *d_offsets, const int32_t *d_headers, int n,
                      int R, const tsdf_cam *cam, int layout, void *hip_stream, float *d_out_tsdf,
                      float *d_out_max_l, float *d_out_mid_p, int32_t *d_out_status) {
  if (n > 0 && (!d_out_tsdf || !d_out_max_l || !d_out_mid_p)) return TSDF_ERR_INVALID_ARG;
  return run(d_depth, d_offsets, d_headers, n, R, cam, layout, hip_stream, d_out_tsdf, d_out_max_l,
             d_out_mid_p, d_out_status, nullptr, nullptr, nullptr, 0);
}

int tsdf_aabb_hip(const float *d_depth, const int64_t *d_offsets, const int32_t *d_headers, int n, int R,
                  const tsdf_cam *cam, void *hip_stream, float *d_out_aabb, float *d_out_grid,
                  float *d_out_ori, int32_t *d_out_status) {
  return run(d_depth, d_offsets, d_headers, n, R, cam, TSDF_LAYOUT_CZYX, hip_stream, nullptr, nullptr,
             nullptr, d_out_status, d_out_aabb, d_out_grid, d_out_ori, 1);
}

}  // extern "C"
