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
//   * one 1024-thread workgroup (16 wave64) per frame, one workgroup per CU (it owns the LDS);
//     the AABB and the grid placement never leave the chip;
//   * phase 1 streams the crop once with 16-byte loads, lane <-> 4 consecutive columns,
//     wave <-> rows.  It does NOT back-project every pixel (one float64 division each):
//     f32(f64(d)/F * (x-cx)) is monotone in d for a fixed column x (and likewise per row),
//     so the AABB is the extreme of the formula applied to each column's / row's (min,max)
//     valid depth — 2(b_w+b_h) divisions per wave instead of b_w*b_h, bit-identical result.
//     The same pass yields the pixel rectangle that holds every valid pixel;
//   * staging: that rectangle (the only pixels phase 2 can ever use — everything outside it is
//     rejected by pre/tsdf_numba.py:36 or :40) is copied into LDS (up to 144 KiB), so the
//     per-voxel gather is an LDS read: no vector-memory latency, and stores never block loads;
//     a rectangle that does not fit falls back to gathering from global memory (L2);
//   * phase 2: each lane owns 4 consecutive voxels along the layout's fastest axis, so every
//     wave store is 1 KiB contiguous (global_store_dwordx4); q = -F/v_z comes from a 1-per-z
//     LDS table (true division), the per-voxel chain uses reciprocals (<= a few ulp64 from the
//     divisions it replaces — 9 orders of magnitude inside the 1e-5 parity bound); a wave whose
//     256 voxels are all rejected or farther than the truncation distance along z skips the
//     x/y terms (the result is then (+-1,+-1,+-1) or 0 by pre/tsdf_numba.py:54-57).
// HBM-bound streaming read + streaming write.  No MFMA (gather/scatter, not a contraction),
// no CPU fallback, gfx950 only.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tsdf.h"

namespace {

constexpr int kWG = 1024;               // threads per workgroup
constexpr int kWaves = kWG / 64;        // wave64
constexpr int kRowUnroll = 4;           // rows in flight per wave in phase 1
constexpr int kMaxR = 128;
constexpr int kStageFloats = 36 * 1024; // 144 KiB depth stage in LDS
constexpr int kRedStride = 12;

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned 16-B access

struct CamK {
  double focal, cx, cy, inv_focal;
  float eps, trunc_vox;
};

#define TSDF_INF __builtin_inff()

// ---- raw VALU min/max (no canonicalising v_max x,x,x in front; operands here are never NaN) ----
__device__ __forceinline__ float vmin(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmin3(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// max(|a|,|b|,|c|,|d|); NaN operands are ignored (IEEE maxNum), all-NaN gives NaN.
__device__ __forceinline__ float vmaxabs4(f4 v) {
  float r;
  asm("v_max3_f32 %0, |%1|, |%2|, |%3|\n\tv_max_f32 %0, %0, |%4|"
      : "=&v"(r)
      : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
  return r;
}

// ---- wave64 reductions: one DPP VALU op per step (s_nop 1 covers the VALU-write -> DPP-read hazard;
// lanes whose DPP source is out of range are write-disabled and keep their value) ----------------
#define TSDF_DPP_REDUCE(OP)                                                             \
  asm("s_nop 1\n\t" OP " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"           \
      "s_nop 1\n\t" OP " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"           \
      "s_nop 1\n\t" OP " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"           \
      "s_nop 1\n\t" OP " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"           \
      "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xf bank_mask:0xf\n\t"        \
      "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xf bank_mask:0xf\n\t"        \
      "s_nop 1"                                                                         \
      : "+v"(v))
#define TSDF_DPP_REDUCE16(OP)                                                           \
  asm("s_nop 1\n\t" OP " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"           \
      "s_nop 1\n\t" OP " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"           \
      "s_nop 1\n\t" OP " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"           \
      "s_nop 1\n\t" OP " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"           \
      "s_nop 1"                                                                         \
      : "+v"(v))

__device__ __forceinline__ float wave_min(float v) {  // result in every lane (wave-uniform)
  TSDF_DPP_REDUCE("v_min_f32_dpp");
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  TSDF_DPP_REDUCE("v_max_f32_dpp");
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// reduce lanes 0..15 only (first DPP row); result wave-uniform
__device__ __forceinline__ float row0_min(float v) {
  TSDF_DPP_REDUCE16("v_min_f32_dpp");
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 15));
}
__device__ __forceinline__ float row0_max(float v) {
  TSDF_DPP_REDUCE16("v_max_f32_dpp");
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 15));
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

__device__ __forceinline__ void acc4(f4 v, float eps, float (&cmin)[4], float (&cmax)[4], float &rmin,
                                     float &rmax) {
  float lo[4], hi[4];
  const float d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool ok = __builtin_fabsf(d[j]) >= eps;  // pre/tsdf_numba.py:87 (NaN -> invalid)
    lo[j] = ok ? d[j] : TSDF_INF;
    hi[j] = ok ? d[j] : -TSDF_INF;
    cmin[j] = vmin(cmin[j], lo[j]);
    cmax[j] = vmax(cmax[j], hi[j]);
  }
  rmin = vmin3(vmin3(rmin, lo[0], lo[1]), lo[2], lo[3]);
  rmax = vmax3(vmax3(rmax, hi[0], hi[1]), hi[2], hi[3]);
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

struct Aabb {
  float mn[3], mx[3];
  int c0, c1, r0, r1;  // bbox-relative rectangle holding every valid pixel (inclusive)
  bool any;
};

// ---- phase 1: AABB of all valid back-projected pixels ---------------------------------------
__device__ __forceinline__ Aabb phase1_aabb(const Frame &f, const CamK &k, float *red) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float xmn = TSDF_INF, xmx = -TSDF_INF, ymn = TSDF_INF, ymx = -TSDF_INF;
  float dmn = TSDF_INF, dmx = -TSDF_INF;
  float cimn = TSDF_INF, cimx = -TSDF_INF, rimn = TSDF_INF, rimx = -TSDF_INF;  // indices (exact in f32)
  // per-wave stash of reduced row extremes: lane i keeps the i-th non-empty row piece
  float s_rmin = TSDF_INF, s_rmax = -TSDF_INF;
  int s_row = 0, cnt = 0;

  auto flush_rows = [&]() {
    if (s_rmin <= s_rmax) {
      const int y = f.t + s_row;
      const float a = backproject_y(s_rmin, y, k), b = backproject_y(s_rmax, y, k);
      ymn = vmin3(ymn, a, b);
      ymx = vmax3(ymx, a, b);
      rimn = vmin(rimn, (float)s_row);
      rimx = vmax(rimx, (float)s_row);
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
        // most 256-pixel segments hold no valid pixel at all: skip them wave-wide
        if (__any(vmaxabs4(va[u]) >= k.eps)) acc4(va[u], k.eps, cmin[0], cmax[0], rmin, rmax);
        if (has1 && __any(vmaxabs4(vb[u]) >= k.eps)) acc4(vb[u], k.eps, cmin[1], cmax[1], rmin, rmax);
        if (__any(rmin <= rmax)) {
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
    // column extremes of this wave's rows -> x extent; depth extremes -> z extent
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (cmin[kk][j] <= cmax[kk][j]) {
          const int col = c0 + 256 * kk + j;
          const int x = f.l + col;
          const float a = backproject_x(cmin[kk][j], x, k), b = backproject_x(cmax[kk][j], x, k);
          xmn = vmin3(xmn, a, b);
          xmx = vmax3(xmx, a, b);
          dmn = vmin(dmn, cmin[kk][j]);
          dmx = vmax(dmx, cmax[kk][j]);
          cimn = vmin(cimn, (float)col);
          cimx = vmax(cimx, (float)col);
        }
      }
  }
  flush_rows();

  // wave -> LDS -> every wave reduces the 16 partials itself (no second barrier)
  float part[10];
  part[0] = wave_min(xmn);
  part[1] = wave_min(ymn);
  part[2] = wave_min(dmn);
  part[3] = wave_min(cimn);
  part[4] = wave_min(rimn);
  part[5] = wave_max(xmx);
  part[6] = wave_max(ymx);
  part[7] = wave_max(dmx);
  part[8] = wave_max(cimx);
  part[9] = wave_max(rimx);
  if (lane < 10) {
    float v = part[0];
#pragma unroll
    for (int i = 1; i < 10; ++i) v = (lane == i) ? part[i] : v;
    red[wave * kRedStride + lane] = v;
  }
  __syncthreads();
  float fin[10];
  const int src = (lane & 15) * kRedStride;
#pragma unroll
  for (int i = 0; i < 5; ++i) fin[i] = row0_min(red[src + i]);
#pragma unroll
  for (int i = 5; i < 10; ++i) fin[i] = row0_max(red[src + i]);

  Aabb a;
  a.any = fin[2] <= fin[7];
  a.mn[0] = fin[0];
  a.mn[1] = fin[1];
  a.mn[2] = -fin[7];  // cam_z = -d   pre/tsdf_numba.py:94
  a.mx[0] = fin[5];
  a.mx[1] = fin[6];
  a.mx[2] = -fin[2];
  a.c0 = a.any ? (int)fin[3] : 0;
  a.c1 = a.any ? (int)fin[8] : -1;
  a.r0 = a.any ? (int)fin[4] : 0;
  a.r1 = a.any ? (int)fin[9] : -1;
  return a;
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

// ---- phase 2 ---------------------------------------------------------------------------------
struct VoxK {
  double cx, cy;
  double it;   // 1 / trunc_dis
  double kq;   // (1/F) * it
  double ncx;  // -cx
  float eps;
  int px0, px1, py0, py1;  // image-coordinate rectangle (inclusive) holding every valid pixel
  int stride;              // gather source: elements per row ...
  int base;                // ... and index of pixel (px0, py0)
};

__device__ __forceinline__ void zero_volume(float *__restrict__ out, int R) {
  const int n4 = 3 * R * R * R / 4;
  f4 *o4 = reinterpret_cast<f4 *>(out);
  const f4 z = {0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < n4; i += kWG) o4[i] = z;
}

// One group of 4 voxels along the fast axis (pre/tsdf_numba.py:26-72 for each).
// LAYOUT 0: the 4 voxels differ in x (same y, z) and use vx[0..3], vz[0], q[0], negthr[0];
// LAYOUT 1: they differ in z (same x, y)    and use vx[0], vz[0..3], q[0..3], negthr[0..3].
// Coordinates are pre-scaled by it = 1/trunc_dis:
//   tx = (v_x - w_x)/trunc = v_x*it - (pix_x-cx)*(pd*kq),  tz = v_z*it + pd*it  (w_z = -pd).
template <int LAYOUT>
__device__ __forceinline__ void voxel_group4(const double (&vx)[4], const double vy, const double (&vz)[4],
                                             const double (&q)[4], const float (&negthr)[4], const VoxK &k,
                                             const float *__restrict__ src, f4 &o0, f4 &o1, f4 &o2) {
  int pix_x[4], pix_y[4];
  bool inb[4];
  float pd[4];
  const double nvy = -vy;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int jx = LAYOUT == 0 ? j : 0, jz = LAYOUT == 0 ? 0 : j;
    if (LAYOUT == 0 && j > 0) {
      pix_y[j] = pix_y[0];
    } else {
      pix_y[j] = trunc_i32(mul_then_add(nvy, q[jz], k.cy));                 // :32
    }
    pix_x[j] = trunc_i32(mul_then_add(vx[jx], q[jz], k.cx));                // :31
    // :36 restricted to the rectangle that holds every valid pixel (outside it :36 or :40 rejects)
    inb[j] = (unsigned)(pix_x[j] - k.px0) <= (unsigned)(k.px1 - k.px0) &&
             (unsigned)(pix_y[j] - k.py0) <= (unsigned)(k.py1 - k.py0);
    const int idx = inb[j] ? (pix_y[j] - k.py0) * k.stride + (pix_x[j] - k.px0) + k.base : k.base;
    pd[j] = src[idx];                                                       // :38-39
  }
  bool ok[4], neg[4];
  double pd64[4], tz[4];
  bool any_near = false;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int jz = LAYOUT == 0 ? 0 : j;
    ok[j] = inb[j] && (__builtin_fabsf(pd[j]) >= k.eps);                    // :40
    pd64[j] = (double)pd[j];
    tz[j] = __builtin_fma(pd64[j], k.it, vz[jz] * k.it);                    // :46,:49
    neg[j] = pd[j] < negthr[jz];                                            // w_z > v_z  :65
    any_near |= ok[j] && (__builtin_fabs(tz[j]) <= 1.0);
  }
  float r0[4], r1[4], r2[4];
  if (__any(any_near)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int jx = LAYOUT == 0 ? j : 0;
      const double a = pd64[j] * k.kq;                                      // pd/F/trunc      :43
      const double wx = ((double)pix_x[j] + k.ncx) * a;                     // w_x/trunc       :44
      const double wy = (k.cy - (double)pix_y[j]) * a;                      // w_y/trunc       :45
      const double tx = __builtin_fma(vx[jx], k.it, -wx);                   // (v_x-w_x)/trunc :47
      const double ty = __builtin_fma(vy, k.it, -wy);                       // :48
      const double s = __builtin_fma(tz[j], tz[j], __builtin_fma(ty, ty, tx * tx));  // dist^2 :51-52
      const bool far = !(s <= 1.0);                                         // :54 (sqrt monotone, sqrt(1)=1)
      r0[j] = far ? 1.0f : fminf(__builtin_fabsf((float)tx), 1.0f);         // :55-60, f32 store :70-72
      r1[j] = far ? 1.0f : fminf(__builtin_fabsf((float)ty), 1.0f);
      r2[j] = far ? 1.0f : fminf(__builtin_fabsf((float)tz[j]), 1.0f);
    }
  } else {
    // every voxel of this wave is rejected or beyond the truncation distance along z alone:
    // dist >= |tz| > 1  ->  (1,1,1)   pre/tsdf_numba.py:54-57
#pragma unroll
    for (int j = 0; j < 4; ++j) r0[j] = r1[j] = r2[j] = 1.0f;
  }
  float *p0 = reinterpret_cast<float *>(&o0), *p1 = reinterpret_cast<float *>(&o1),
        *p2 = reinterpret_cast<float *>(&o2);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float sg = neg[j] ? -1.0f : 1.0f;
    p0[j] = ok[j] ? sg * r0[j] : 0.0f;                                      // :33-41, :65-68
    p1[j] = ok[j] ? sg * r1[j] : 0.0f;
    p2[j] = ok[j] ? sg * r2[j] : 0.0f;
  }
}

// smallest float32 >= t  (so that for a float32 p:  p < t  <=>  p < result)
__device__ __forceinline__ float f32_round_up(double t) {
  float f = (float)t;
  if ((double)f < t) f = nextafterf(f, TSDF_INF);
  return f;
}

template <int LAYOUT>
__device__ __forceinline__ void phase2(const Grid &g, const VoxK &vk, int R, const double *qtab,
                                       const float *negtab, const float *__restrict__ src,
                                       float *__restrict__ out) {
  const int tid = threadIdx.x;
  const double vl = (double)g.voxel_len;
  const double ox = (double)g.ori[0], oy = (double)g.ori[1], oz = (double)g.ori[2];
  const int R4 = R / 4;
  const int G = R * R4;  // groups of 4 voxels per slow-axis slice
  const int64_t R3 = (int64_t)R * R * R;

  // slow axis s (z for LAYOUT 0, x for LAYOUT 1); group gi -> (y, fast4)
  int g0, gstep, s0, sstep;
  if (G <= kWG && (kWG % G) == 0) {
    g0 = tid % G;
    gstep = G;  // one group per thread, kWG/G slices at a time
    s0 = tid / G;
    sstep = kWG / G;
  } else {
    g0 = tid;
    gstep = kWG;
    s0 = 0;
    sstep = 1;
  }
  for (int gi = g0; gi < G; gi += gstep) {
    const int f4i = (gi % R4) * 4;
    const int y = gi / R4;
    const double vy = oy + (double)y * vl;                                  // :27
    double vx[4], vz[4], q[4];
    float negthr[4];
    if constexpr (LAYOUT == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) vx[j] = ox + (double)(f4i + j) * vl;      // :26
      for (int z = s0; z < R; z += sstep) {
        vz[0] = oz + (double)z * vl;                                        // :28
        q[0] = qtab[z];
        negthr[0] = negtab[z];
        f4 o0, o1, o2;
        voxel_group4<0>(vx, vy, vz, q, negthr, vk, src, o0, o1, o2);
        const int64_t e = ((int64_t)z * R + y) * R + f4i;                   // o[c][z][y][x] :70-72
        *reinterpret_cast<f4 *>(out + e) = o0;
        *reinterpret_cast<f4 *>(out + R3 + e) = o1;
        *reinterpret_cast<f4 *>(out + 2 * R3 + e) = o2;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        vz[j] = oz + (double)(f4i + j) * vl;
        q[j] = qtab[f4i + j];
        negthr[j] = negtab[f4i + j];
      }
      for (int x = s0; x < R; x += sstep) {
        vx[0] = ox + (double)x * vl;
        f4 o0, o1, o2;
        voxel_group4<1>(vx, vy, vz, q, negthr, vk, src, o0, o1, o2);
        const int64_t e = ((int64_t)x * R + y) * R + f4i;                   // o[c][x][y][z] tsdf_for.py:118-120
        *reinterpret_cast<f4 *>(out + e) = o0;
        *reinterpret_cast<f4 *>(out + R3 + e) = o1;
        *reinterpret_cast<f4 *>(out + 2 * R3 + e) = o2;
      }
    }
  }
}

template <int RT, int LAYOUT>
__global__ __launch_bounds__(kWG) void tsdf_fused_kernel(
    const float *__restrict__ depth, const int64_t *__restrict__ offsets,
    const int32_t *__restrict__ headers, int n, int Rrt, CamK cam, float *__restrict__ out_tsdf,
    float *__restrict__ out_max_l, float *__restrict__ out_mid_p, int32_t *__restrict__ out_status,
    float *__restrict__ out_aabb, float *__restrict__ out_grid, float *__restrict__ out_ori,
    int aabb_only) {
  __shared__ __attribute__((aligned(16))) float stage[kStageFloats];
  __shared__ double qtab[kMaxR];
  __shared__ float negtab[kMaxR];
  __shared__ float red[kWaves * kRedStride];

  const int R = RT ? RT : Rrt;
  const int frame = blockIdx.x;
  if (frame >= n) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

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
  Aabb ab;
  ab.any = false;
  ab.mn[0] = ab.mn[1] = ab.mn[2] = ab.mx[0] = ab.mx[1] = ab.mx[2] = 0.f;
  ab.c0 = ab.r0 = 0;
  ab.c1 = ab.r1 = -1;
  Grid g;
  g.mid[0] = g.mid[1] = g.mid[2] = 0.f;
  g.max_l = g.voxel_len = g.trunc = 0.f;
  g.ori[0] = g.ori[1] = g.ori[2] = 0.f;

  if (f.bw <= 0 || f.bh <= 0 || (int64_t)f.bw * (int64_t)f.bh != off1 - off0) {
    status = TSDF_FRAME_BAD_HEADER;  // block-uniform
  } else {
    ab = phase1_aabb(f, cam, red);
    if (!ab.any) {
      status = TSDF_FRAME_DEGENERATE;
      ab.mn[0] = ab.mn[1] = ab.mn[2] = ab.mx[0] = ab.mx[1] = ab.mx[2] = 0.f;
    } else {
      g = glue(ab.mn, ab.mx, R, cam);
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
      a[0] = ab.mn[0]; a[1] = ab.mn[1]; a[2] = ab.mn[2];
      a[3] = ab.mx[0]; a[4] = ab.mx[1]; a[5] = ab.mx[2];
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

  // ---- per-z tables (true divisions, once per z) ----
  const double vl = (double)g.voxel_len;
  if (tid < R) {
    const double v_z = (double)g.ori[2] + (double)tid * vl;  // :28
    qtab[tid] = -cam.focal / v_z;                            // :30
    negtab[tid] = f32_round_up(-v_z);                        // pd < -v_z  <=>  w_z > v_z  (:65)
  }

  // ---- stage the rectangle of valid pixels into LDS ----
  const int sw = ab.c1 - ab.c0 + 1, sh = ab.r1 - ab.r0 + 1;
  const bool staged = (int64_t)sw * sh <= kStageFloats;  // block-uniform
  if (staged) {
    const float *__restrict__ srcp = f.depth + (int64_t)ab.r0 * f.bw + ab.c0;
    for (int r0 = wave; r0 < sh; r0 += kWaves * 4) {
      for (int cb = 0; cb < sw; cb += 256) {
        float v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int r = r0 + kWaves * u, c = cb + lane + 64 * kk;
            v[u][kk] = (r < sh && c < sw) ? srcp[(int64_t)r * f.bw + c] : 0.f;
          }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int r = r0 + kWaves * u, c = cb + lane + 64 * kk;
            if (r < sh && c < sw) stage[r * sw + c] = v[u][kk];
          }
      }
    }
  }
  __syncthreads();

  VoxK vk;
  vk.cx = cam.cx;
  vk.cy = cam.cy;
  vk.it = 1.0 / (double)g.trunc;
  vk.kq = cam.inv_focal * vk.it;
  vk.ncx = -cam.cx;
  vk.eps = cam.eps;
  vk.px0 = f.l + ab.c0;
  vk.px1 = f.l + ab.c1;
  vk.py0 = f.t + ab.r0;
  vk.py1 = f.t + ab.r1;
  if (staged) {
    vk.stride = sw;
    vk.base = 0;
    phase2<LAYOUT>(g, vk, R, qtab, negtab, stage, out);
  } else {
    vk.stride = f.bw;
    vk.base = ab.r0 * f.bw + ab.c0;
    phase2<LAYOUT>(g, vk, R, qtab, negtab, f.depth, out);
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
