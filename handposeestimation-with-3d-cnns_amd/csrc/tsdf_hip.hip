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
//   * one persistent launch; one 1024-thread workgroup (16 wave64) per CU, which owns the CU's LDS.
//     Its two 512-thread halves ("groups") each process whole frames — the first one positional, the
//     rest from an atomic work queue — and take turns on the single 128 KiB LDS stage, so one group's
//     row streaming overlaps the other group's voxel arithmetic and stores.  Groups synchronise on LDS
//     counters (s_barrier would span both).  The AABB and the grid placement never leave the chip;
//   * phase 1 streams the crop once with 16-byte loads, lane <-> 4 consecutive columns, wave <-> rows,
//     two register buffers in ping-pong behind counted vmcnt waits.  It does NOT back-project every
//     pixel (one float64 division each): f32(f64(d)/F * (x-cx)) is monotone in d for a fixed column x
//     (and likewise per row), so the AABB is the extreme of the formula applied to each column's /
//     row's (min,max) valid depth — bit-identical result, 2(b_w+b_h) evaluations per wave instead of
//     b_w*b_h; d/F uses Markstein's correction (exact quotient in 3 ops).  The same pass yields the pixel
//     rectangle that holds every valid pixel;
//   * staging: that rectangle (the only pixels phase 2 can ever use — everything outside it is
//     rejected by pre/tsdf_numba.py:36 or :40) is copied into LDS by LDS-DMA (global_load_lds_dwordx4),
//     so the per-voxel gather is an LDS read: no vector-memory latency, and stores never block loads;
//     a rectangle over 32 Ki pixels falls back to gathering from global memory (L2);
//   * phase 2: pix_x depends on (x,z) only and pix_y on (y,z) only -> both tabulated per frame in LDS
//     (true division for q = -F/v_z, unfused multiply-add, v_cvt_i32_f64 truncation).  Each lane owns 4
//     consecutive voxels along the layout's fastest axis, so every wave store is 1 KiB contiguous
//     (global_store_dwordx4 nt: the volume is written once; non-temporal stores keep the streamed depth
//     rows cached for the staging copy).  The per-voxel chain uses reciprocals (<= a few ulp64 from the
//     divisions it replaces — 10 orders of magnitude inside the 1e-5 parity bound); a wave whose 256
//     voxels are all rejected or farther than the truncation distance along z skips the x/y terms (the
//     result is then (+-1,+-1,+-1) or 0 by pre/tsdf_numba.py:54-57);
//   * the augmented form (template AUG) maps every valid pixel / voxel centre / surface point through a
//     per-frame affine transform instead (tsdf_voxelize_aug_hip, re-specified: see include/tsdf.h).
// HBM-bound streaming read + streaming write.  No MFMA (gather/scatter, not a contraction), no
// inter-workgroup communication (XCD placement is irrelevant), no CPU fallback, gfx950 only.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <type_traits>

#include "../../include/tsdf.h"

namespace {

constexpr int kWG = 1024;               // threads per workgroup (16 wave64; needs <= 128 VGPRs)
#ifndef TSDF_GROUPS
#define TSDF_GROUPS 2
#endif
constexpr int kGroups = TSDF_GROUPS;    // half-workgroups: each walks its own frames, see the kernel
constexpr int kGW = kWG / kGroups;      // threads per group
constexpr int kGWaves = kGW / 64;       // waves per group
constexpr int kMaxR = 128;
constexpr int kStageFloats = 32 * 1024; // 128 KiB depth stage in LDS (>= 181 x 181 pixels)
constexpr int kTabR = 32;               // projection tables for R <= kTabR (2 x 4 KiB)
constexpr int kRedStride = 12;

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned 16-B access

struct CamK {
  double focal, cx, cy, inv_focal;
  float eps, trunc_vox;
};

#define TSDF_INF __builtin_inff()

// In-kernel timeline stamps: compiled only into the diagnostic library (make stamps), never into
// libtsdf_hip.so.  Lane 0 of wave 0 of each workgroup records s_memrealtime (100 MHz) per phase.
#ifdef TSDF_STAMPS
constexpr int kStampSlots = 16, kStampFrames = 8, kStampBlocks = 512;
__device__ unsigned long long g_stamps[kStampBlocks * kStampFrames * kStampSlots];
#define TSDF_STAMP(iter, slot)                                                                       \
  do {                                                                                               \
    if ((threadIdx.x & (kGW - 1)) == 0 && blockIdx.x < kStampBlocks && (iter) < kStampFrames)                      \
      g_stamps[(blockIdx.x * kStampFrames + (iter)) * kStampSlots + (slot)] =                        \
          __builtin_amdgcn_s_memrealtime();                                                          \
  } while (0)
#define TSDF_STAMP_VAL(iter, slot, val)                                                              \
  do {                                                                                               \
    if ((threadIdx.x & (kGW - 1)) == 0 && blockIdx.x < kStampBlocks && (iter) < kStampFrames)                      \
      g_stamps[(blockIdx.x * kStampFrames + (iter)) * kStampSlots + (slot)] = (val);                 \
  } while (0)
#else
#define TSDF_STAMP(iter, slot) do { } while (0)
#define TSDF_STAMP_VAL(iter, slot, val) do { } while (0)
#endif

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

// Forward map, one row {A_i0, A_i1, A_i2, b_i}: o = fma(A_i0, x, fma(A_i1, y, fma(A_i2, z, b_i))) — the augmented
// form's arithmetic contract (oracle/tsdf_oracle.c::affine3_fwd).  Three instructions per row; with the
// identity row {1,0,0,0} it returns x exactly, so the identity map reproduces the plain path bit for bit.
// (The inverse map is specified with separately rounded products instead, because those can be tabulated
// per grid index: see phase2_aug.)
__device__ __forceinline__ double affine_row(const double *m, double px, double py, double pz) {
  return __builtin_fma(m[0], px, __builtin_fma(m[1], py, __builtin_fma(m[2], pz, m[3])));
}

// fl64(d / F) without the hardware division sequence (~12 dependent float64 instructions):
// with y = RN(1/F) computed by a true division on the host, q0 = RN(d*y) is within 1 ulp of d/F,
// r = d - q0*F is exact in one fma, and RN(q0 + r*y) is the correctly rounded quotient (Markstein's
// correction step).  The AABB parity tests compare the result bit for bit with the oracle's division.
__device__ __forceinline__ double div_by_focal(double d, const CamK &k) {
  const double q0 = d * k.inv_focal;
  const double r = __builtin_fma(-q0, k.focal, d);
  return __builtin_fma(r, k.inv_focal, q0);
}

// A.1 x: f32( (f64(d)/F) * (x - cx) )      pre/tsdf_numba.py:91-92,95
__device__ __forceinline__ float backproject_x(float d, int x, const CamK &k) {
  const double q = div_by_focal((double)d, k);
  return (float)(q * ((double)x - k.cx));
}
// A.1 y: f32( (-(f64(d)/F)) * (y - cy) )   pre/tsdf_numba.py:91,93,95
__device__ __forceinline__ float backproject_y(float d, int y, const CamK &k) {
  const double q = div_by_focal((double)d, k);
  return (float)((-q) * ((double)y - k.cy));
}

// P consecutive pixels of one lane, P in 1..8 (16-byte + 12/8/4-byte pieces, 4-byte aligned).
template <int P>
struct PixN {
  float d[P];
};

typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));

template <int P>
__device__ __forceinline__ PixN<P> load_pix(const float *__restrict__ p) {
  PixN<P> r;
  constexpr int Q = P / 4, T = P % 4;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const f4 v = *reinterpret_cast<const f4u *>(p + 4 * q);
    r.d[4 * q] = v.x;
    r.d[4 * q + 1] = v.y;
    r.d[4 * q + 2] = v.z;
    r.d[4 * q + 3] = v.w;
  }
  if constexpr (T == 1) {
    r.d[4 * Q] = p[4 * Q];
  } else if constexpr (T == 2) {
    const auto v = *reinterpret_cast<const f2u *>(p + 4 * Q);
    r.d[4 * Q] = v.x;
    r.d[4 * Q + 1] = v.y;
  } else if constexpr (T == 3) {
    const auto v = *reinterpret_cast<const f3u *>(p + 4 * Q);
    r.d[4 * Q] = v.x;
    r.d[4 * Q + 1] = v.y;
    r.d[4 * Q + 2] = v.z;
  }
  return r;
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

// The 10 extents phase 1 produces (per workgroup, then per frame):
//   [0..4] minima: cam x, cam y, depth, valid column index, valid row index   [5..9] the maxima
constexpr int kExt = 10;

// ---- phase 1: extents of all valid back-projected pixels of rows [rbeg, rend) ----------------
// NW waves cooperate (row = rbeg + wave + NW*i); the result is wave-uniform in every thread.
// `wave` is the (scalar) index of this wave among the NW cooperating waves, `sync` their barrier.
// AUG: the extents are those of the affinely mapped cloud p' = A p + b (xf = 12 doubles), which needs
// every valid pixel transformed (the monotone shortcut does not survive a rotation).
template <int NW, bool AUG, typename SYNC>
__device__ __forceinline__ void phase1_extents(const Frame &f, const CamK &k, int rbeg, int rend, float *red,
                                               float (&fin)[kExt], const int wave, SYNC sync,
                                               int stamp_iter = 0, const double *xf = nullptr) {
  (void)stamp_iter;
  constexpr int kWaves = NW;
  const int lane = threadIdx.x & 63;
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

  // One pass over the rows for the columns [cbase, cbase + 64*P): lane <-> P consecutive columns, so a
  // row of up to 320 pixels is ONE visit with every lane busy (P = ceil(width / 64), at most 5).  Loads are
  // unconditional vector loads in straight-line code (rows past the band are clamped and ignored), two
  // register buffers in ping-pong: while one is reduced the other one's rows stream in behind a counted
  // vmcnt.  A lane whose window crosses the row end reads into the next row (masked); only in the very
  // last row of the frame would that leave the buffer, so that row alone takes guarded element loads.
  auto row_pass = [&](int cbase, auto p_tag) {
    constexpr int P = decltype(p_tag)::value;
    constexpr int kU = P <= 4 ? 4 : 3;           // rows per register buffer (bytes in flight vs VGPRs)
    constexpr int kStep = kWaves * kU;
    float cmin[P], cmax[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
      cmin[j] = TSDF_INF;
      cmax[j] = -TSDF_INF;
    }
    const int c0 = cbase + P * lane;
    bool mine[P];                                 // the lane owns column c0 + j
#pragma unroll
    for (int j = 0; j < P; ++j) mine[j] = c0 + j < f.bw;
    const int cload = mine[0] ? c0 : f.bw - P;    // lanes past the row reload its last P pixels (all masked)
    const bool ragged = (f.bw - cbase) % P != 0 && cbase + 64 * P >= f.bw;  // some lane straddles the row end
    const float nan = __builtin_nanf("");

    auto load_rows = [&](int row0, PixN<P> (&v)[kU]) {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int row = row0 + kWaves * u;
        const int rc = row < rend ? row : rend - 1;  // scalar
        const float *rp = f.depth + (int64_t)rc * f.bw;
        if (ragged && rc == f.bh - 1) {  // scalar, last row of the frame only: nothing may be read past it
#pragma unroll
          for (int j = 0; j < P; ++j) v[u].d[j] = mine[j] ? rp[c0 + j] : nan;
        } else {
          v[u] = load_pix<P>(rp + cload);
        }
      }
    };

    auto reduce_rows = [&](int row0, const PixN<P> (&v)[kU]) {
      if (cnt > 64 - kU) flush_rows();  // wave-uniform (cnt is)
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int row = row0 + kWaves * u;
        bool ok[P];
        bool any_ok = false;
#pragma unroll
        for (int j = 0; j < P; ++j) {
          ok[j] = (__builtin_fabsf(v[u].d[j]) >= k.eps) & mine[j];  // pre/tsdf_numba.py:87 (NaN -> invalid)
          any_ok |= ok[j];
        }
        // most row segments hold no valid pixel at all: skip them wave-wide
        if (row < rend && __any(any_ok)) {
          if constexpr (AUG) {
            const double ym = (double)(f.t + row) - k.cy;
#pragma unroll
            for (int j = 0; j < P; ++j) {
              if (ok[j]) {
                const int col = c0 + j;
                const float dj = v[u].d[j];
                const double q = div_by_focal((double)dj, k);               // :91
                const double px = q * ((double)(f.l + col) - k.cx);         // :92
                const double py = (-q) * ym;                                // :93
                const double pz = -(double)dj;                              // :94
                const float ax = (float)affine_row(xf + 0, px, py, pz);
                const float ay = (float)affine_row(xf + 4, px, py, pz);
                const float az = (float)affine_row(xf + 8, px, py, pz);
                xmn = vmin(xmn, ax);
                xmx = vmax(xmx, ax);
                ymn = vmin(ymn, ay);
                ymx = vmax(ymx, ay);
                dmn = vmin(dmn, -az);  // stored negated: aabb_from_extents flips z back
                dmx = vmax(dmx, -az);
                cimn = vmin(cimn, (float)col);
                cimx = vmax(cimx, (float)col);
                rimn = vmin(rimn, (float)row);
                rimx = vmax(rimx, (float)row);
              }
            }
          } else {
            float rmin = TSDF_INF, rmax = -TSDF_INF;
#pragma unroll
            for (int j = 0; j < P; ++j) {
              const float lo = ok[j] ? v[u].d[j] : TSDF_INF;
              const float hi = ok[j] ? v[u].d[j] : -TSDF_INF;
              cmin[j] = vmin(cmin[j], lo);
              cmax[j] = vmax(cmax[j], hi);
              rmin = vmin(rmin, lo);
              rmax = vmax(rmax, hi);
            }
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
    };

    PixN<P> bufA[kU], bufB[kU];
    load_rows(rbeg + wave, bufA);
    for (int row0 = rbeg + wave; row0 < rend; row0 += 2 * kStep) {
      load_rows(row0 + kStep, bufB);
      reduce_rows(row0, bufA);
      load_rows(row0 + 2 * kStep, bufA);
      reduce_rows(row0 + kStep, bufB);
    }
    if constexpr (!AUG) {
      // column extremes of this wave's rows -> x extent; depth extremes -> z extent
#pragma unroll
      for (int j = 0; j < P; ++j) {
        if (cmin[j] <= cmax[j]) {
          const int col = c0 + j;
          const int x = f.l + col;
          const float a = backproject_x(cmin[j], x, k), b = backproject_x(cmax[j], x, k);
          xmn = vmin3(xmn, a, b);
          xmx = vmax3(xmx, a, b);
          dmn = vmin(dmn, cmin[j]);
          dmx = vmax(dmx, cmax[j]);
          cimn = vmin(cimn, (float)col);
          cimx = vmax(cimx, (float)col);
        }
      }
    }
  };
  // P is capped at 5 (320 columns per pass, the MSRA sensor width): wider windows only cost registers
  // (the whole kernel lives in 128 VGPRs) and would spill.
  if (rbeg < rend) {
    for (int cbase = 0; cbase < f.bw; cbase += 320) {
      const int w = f.bw - cbase;  // columns left (scalar)
      if (w <= 64) row_pass(cbase, std::integral_constant<int, 1>{});
      else if (w <= 128) row_pass(cbase, std::integral_constant<int, 2>{});
      else if (w <= 192) row_pass(cbase, std::integral_constant<int, 3>{});
      else if (w <= 256) row_pass(cbase, std::integral_constant<int, 4>{});
      else row_pass(cbase, std::integral_constant<int, 5>{});
    }
  }
  TSDF_STAMP(stamp_iter, 1);
  flush_rows();
  TSDF_STAMP(stamp_iter, 2);

  // wave -> LDS -> every wave reduces the partials itself (no second barrier)
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
  TSDF_STAMP(stamp_iter, 3);
  sync();
  static_assert(kWaves <= 16, "the cross-wave reduction uses one 16-lane DPP row");
  const bool has = (lane & 15) < kWaves;
  const int src = has ? (lane & 15) * kRedStride : 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) fin[i] = row0_min(has ? red[src + i] : TSDF_INF);
#pragma unroll
  for (int i = 5; i < 10; ++i) fin[i] = row0_max(has ? red[src + i] : -TSDF_INF);
}

__device__ __forceinline__ Aabb aabb_from_extents(const float (&fin)[kExt]) {
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
  int px0, py0;  // image coordinates of the first pixel of the rectangle holding every valid pixel
  int dx, dy;    // its extent - 1 (inclusive upper bounds of rectangle-relative coordinates)
  int stride;    // gather source: elements per row ...
  int base;      // ... and index of the rectangle's first pixel
};

__device__ __forceinline__ void zero_volume(float *__restrict__ out, int R, int gtid) {
  const int n4 = 3 * R * R * R / 4;
  f4 *o4 = reinterpret_cast<f4 *>(out);
  const f4 z = {0.f, 0.f, 0.f, 0.f};
  for (int i = gtid; i < n4; i += kGW) o4[i] = z;
}

// Projection of a voxel coordinate onto a pixel coordinate, pre/tsdf_numba.py:30-32:
//   pix = int(v * q + c)  with q = -F / v_z   (multiply, round, add, round, truncate)
// returned relative to the valid-pixel rectangle, or -1 when outside it (then :36 or :40 rejects).
__device__ __forceinline__ int project_rel(double v, double q, double c, int p0, int dmax) {
  const int rel = trunc_i32(mul_then_add(v, q, c)) - p0;
  return (unsigned)rel <= (unsigned)dmax ? rel : -1;
}

// smallest float32 >= t  (so that for a float32 p:  p < t  <=>  p < result)
__device__ __forceinline__ float f32_round_up(double t) {
  float f = (float)t;
  if ((double)f < t) f = nextafterf(f, TSDF_INF);
  return f;
}

// Per-voxel value, pre/tsdf_numba.py:36-68, for the 4 voxels of one lane.  Coordinates are pre-scaled
// by it = 1/trunc_dis:  tx = v_x*it - (pix_x-cx)*(pd*kq),  ty likewise,  tz = v_z*it + pd*it (w_z = -pd).
//   ex[j], ry[j]          rectangle-relative pixel of voxel j (or -1: rejected)
//   vxs[j], vys, vzs[j]   pre-scaled voxel centre;   negthr[j] = f32_round_up(-v_z)
// The gather source is either the LDS stage or (rectangles too large for it) the frame in global memory.
// It is passed with its address space in the type: a generic pointer would make every gather a FLAT load,
// which is counted in vmcnt together with the volume stores, so each loop iteration would wait for the
// previous iteration's stores to be acknowledged by memory.  As ds_read the gather only touches lgkmcnt
// and the stores stay in flight.
typedef const __attribute__((address_space(3))) float *LdsSrc;
typedef const __attribute__((address_space(1))) float *GlobalSrc;
typedef __attribute__((address_space(1))) float *GlobalOut;  // the output volume

template <class SrcP>
__device__ __forceinline__ void voxel_values4(const int (&ex)[4], const int (&ry)[4], const double (&vxs)[4],
                                              const double vys, const double (&vzs)[4],
                                              const float (&negthr)[4], const VoxK &k,
                                              const SrcP src, f4 &o0, f4 &o1, f4 &o2) {
  float pd[4];
  bool inb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    inb[j] = (ex[j] | ry[j]) >= 0;                                          // :36 (both in range)
    int idx = __mul24(ry[j], k.stride) + ex[j] + k.base;
    idx = inb[j] ? idx : k.base;
    pd[j] = src[idx];                                                       // :38-39 (always in bounds)
  }
  bool ok[4], neg[4];
  double pd64[4], tz[4];
  bool any_near = false;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ok[j] = inb[j] & (__builtin_fabsf(pd[j]) >= k.eps);                     // :40
    pd64[j] = (double)pd[j];
    tz[j] = __builtin_fma(pd64[j], k.it, vzs[j]);                           // :46,:49
    neg[j] = pd[j] < negthr[j];                                             // w_z > v_z  :65
    any_near |= ok[j] & (__builtin_fabs(tz[j]) <= 1.0);
  }
  // r*: |t| clamped to 1 (:58-60); stays (1,1,1) when dist > 1 (:54-57)
  float r0[4] = {1.f, 1.f, 1.f, 1.f}, r1[4] = {1.f, 1.f, 1.f, 1.f}, r2[4] = {1.f, 1.f, 1.f, 1.f};
  if (__any(any_near)) {
    // otherwise every voxel of this wave is rejected or beyond the truncation distance along z
    // alone: dist >= |tz| > 1 -> (1,1,1), and the x/y terms are not needed
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double a = pd64[j] * k.kq;                                      // pd/F/trunc         :43
      const double dxi = (double)(ex[j] + k.px0) + k.ncx;                   // pix_x - cx         :44
      const double dyi = (double)(ry[j] + k.py0) - k.cy;                    // pix_y - cy         :45
      const double tx = __builtin_fma(-dxi, a, vxs[j]);                     // (v_x - w_x)/trunc  :47
      const double ty = __builtin_fma(dyi, a, vys);                         // (v_y - w_y)/trunc  :48, w_y = -dyi*q
      const double s = __builtin_fma(tz[j], tz[j], __builtin_fma(ty, ty, tx * tx));  // dist^2 :51-52
      const bool nearv = s <= 1.0;                                          // :54 (sqrt monotone, sqrt(1)=1)
      const float m0 = vmin(__builtin_fabsf((float)tx), 1.0f);              // f32 store :70-72
      const float m1 = vmin(__builtin_fabsf((float)ty), 1.0f);
      const float m2 = vmin(__builtin_fabsf((float)tz[j]), 1.0f);
      r0[j] = nearv ? m0 : 1.0f;
      r1[j] = nearv ? m1 : 1.0f;
      r2[j] = nearv ? m2 : 1.0f;
    }
  }
  float *p0 = reinterpret_cast<float *>(&o0), *p1 = reinterpret_cast<float *>(&o1),
        *p2 = reinterpret_cast<float *>(&o2);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // sign :65-68 and zero for rejected voxels :33-41 as bit masks
    const unsigned sg = neg[j] ? 0x80000000u : 0u;
    const unsigned keep = ok[j] ? 0xffffffffu : 0u;
    p0[j] = __uint_as_float((__float_as_uint(r0[j]) | sg) & keep);
    p1[j] = __uint_as_float((__float_as_uint(r1[j]) | sg) & keep);
    p2[j] = __uint_as_float((__float_as_uint(r2[j]) | sg) & keep);
  }
}

#ifndef TSDF_NT_STORE
#define TSDF_NT_STORE 1
#endif
#ifndef TSDF_TAIL_HELP
#define TSDF_TAIL_HELP 1
#endif
// One 16-byte store of the output volume.  It is written once and never re-read here, so it goes out
// non-temporal: the depth rows this CU has just streamed stay in L2 / Infinity Cache for the staging
// copy instead of being evicted by 393 KB of output per frame (measured: 180 -> 155 us per 1024 frames).
__device__ __forceinline__ void store_vol4(GlobalOut p, f4 v) {
#if TSDF_NT_STORE
  __builtin_nontemporal_store(v, (__attribute__((address_space(1))) f4 *)p);
#else
  *(__attribute__((address_space(1))) f4 *)p = v;
#endif
}

// LDS-resident per-frame tables.  The pixel a voxel projects to factorises: pix_x depends on (x, z)
// only and pix_y on (y, z) only, so for R <= kTabR both are tabulated once per frame (R*R entries
// each, one pair per thread) instead of 3 float64 operations + a range test per voxel.
struct ZEntry {
  double q;      // -F / v_z                     :30
  double vzs;    // v_z / trunc_dis
  float negthr;  // f32_round_up(-v_z)
  float pad;
};

// table index of (fast, slow) coordinates: the 4 entries a lane needs are contiguous
template <int LAYOUT>
__device__ __forceinline__ int tab_index(int x_or_y, int z, int R) {
  return LAYOUT == 0 ? z * R + x_or_y : x_or_y * R + z;
}

template <int LAYOUT, int T, class SrcP>
__device__ __forceinline__ void phase2(const Grid &g, const CamK &cam, const VoxK &vk, int R,
                                       const ZEntry *ztab, const int *pxtab, const int *pytab,
                                       const bool use_tab, const SrcP src,
                                       const GlobalOut out, const int tid) {
  // tid in [0, T): T = kGW when the group works alone, 2*kGW when the CU's other group helps (its
  // threads come in as kGW + gtid)
  const double vl = (double)g.voxel_len;
  const double ox = (double)g.ori[0], oy = (double)g.ori[1];
  const int R4 = R / 4;
  const int G = R * R4;  // groups of 4 voxels per slow-axis slice
  const int64_t R3 = (int64_t)R * R * R;

  // slow axis s (z for LAYOUT 0, x for LAYOUT 1); group gi -> (y, fast4)
  int g0, gstep, s0, sstep;
  if (G <= T && (T % G) == 0) {
    g0 = tid % G;
    gstep = G;  // one group of 4 voxels per thread, T/G slices at a time
    s0 = tid / G;
    sstep = T / G;
  } else {
    g0 = tid;
    gstep = T;
    s0 = 0;
    sstep = 1;
  }
  for (int gi = g0; gi < G; gi += gstep) {
    const int f4i = (gi % R4) * 4;
    const int y = gi / R4;
    const double vy = oy + (double)y * vl;                                  // :27
    const double vys = vy * vk.it;
    if constexpr (LAYOUT == 0) {
      // lanes run along x: v_x fixed per lane, loop over z
      double vx[4], vxs[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        vx[j] = ox + (double)(f4i + j) * vl;                                // :26
        vxs[j] = vx[j] * vk.it;
      }
      for (int z = s0; z < R; z += sstep) {
        const ZEntry ze = ztab[z];
        int ex[4], ry[4];
        if (use_tab) {
          const int4 e = *reinterpret_cast<const int4 *>(pxtab + z * R + f4i);
          ex[0] = e.x; ex[1] = e.y; ex[2] = e.z; ex[3] = e.w;
          ry[0] = pytab[z * R + y];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) ex[j] = project_rel(vx[j], ze.q, cam.cx, vk.px0, vk.dx);  // :31
          ry[0] = project_rel(-vy, ze.q, cam.cy, vk.py0, vk.dy);                                 // :32
        }
        ry[1] = ry[2] = ry[3] = ry[0];
        const double vzs[4] = {ze.vzs, ze.vzs, ze.vzs, ze.vzs};
        const float negthr[4] = {ze.negthr, ze.negthr, ze.negthr, ze.negthr};
        f4 o0, o1, o2;
        voxel_values4(ex, ry, vxs, vys, vzs, negthr, vk, src, o0, o1, o2);
        const int64_t e = ((int64_t)z * R + y) * R + f4i;                   // o[c][z][y][x] :70-72
        store_vol4(out + e, o0);
        store_vol4(out + R3 + e, o1);
        store_vol4(out + 2 * R3 + e, o2);
      }
    } else {
      // lanes run along z: q, v_z and pix_y fixed per lane, loop over x
      double q[4], vzs[4];
      float negthr[4];
      int ry[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const ZEntry ze = ztab[f4i + j];
        q[j] = ze.q;
        vzs[j] = ze.vzs;
        negthr[j] = ze.negthr;
      }
      if (use_tab) {
        const int4 e = *reinterpret_cast<const int4 *>(pytab + y * R + f4i);
        ry[0] = e.x; ry[1] = e.y; ry[2] = e.z; ry[3] = e.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) ry[j] = project_rel(-vy, q[j], cam.cy, vk.py0, vk.dy);
      }
      for (int x = s0; x < R; x += sstep) {
        const double vx = ox + (double)x * vl;
        const double vx1 = vx * vk.it;
        const double vxs[4] = {vx1, vx1, vx1, vx1};
        int ex[4];
        if (use_tab) {
          const int4 e = *reinterpret_cast<const int4 *>(pxtab + x * R + f4i);
          ex[0] = e.x; ex[1] = e.y; ex[2] = e.z; ex[3] = e.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) ex[j] = project_rel(vx, q[j], cam.cx, vk.px0, vk.dx);
        }
        f4 o0, o1, o2;
        voxel_values4(ex, ry, vxs, vys, vzs, negthr, vk, src, o0, o1, o2);
        const int64_t e = ((int64_t)x * R + y) * R + f4i;                   // o[c][x][y][z] tsdf_for.py:118-120
        store_vol4(out + e, o0);
        store_vol4(out + R3 + e, o1);
        store_vol4(out + 2 * R3 + e, o2);
      }
    }
  }
}

// ---- synchronisation inside one half-workgroup (group) -------------------------------------------
// s_barrier spans all 16 waves, so the 8 waves of a group meet on an LDS counter instead: monotonic
// count, lane 0 of each wave adds 1 and polls until the group's epoch target is reached.  LDS
// operations of a wave execute in order, so everything a wave wrote to LDS before its arrival is
// visible to whoever sees the count.  Only LDS is ordered here (no vmcnt wait: output stores stay
// in flight across these barriers).
struct FrameHdr {
  int frame;  // -1: no more work
  int l, t, r, b;
  int pad;
  int64_t off0, off1;
};

// Tail help: a group that finds the queue empty does not leave at once.  It raises idle[] and waits; the
// CU's other group, on reaching phase 2 of what is then necessarily its last frame, sees the flag, posts
// the frame's voxel parameters here and both groups split the slow axis of the volume (the stage and the
// tables are in LDS, which the two share).  Nothing has to be handed back: the helper leaves when done.
struct HelpReq {
  Grid g;
  VoxK vk;
  const float *src;  // the frame in global memory (gather source when the rectangle is not staged)
  float *out;
  int frame, use_tab, staged, pad;
};

struct GroupCtl {
  int bar[kGroups];
  int lock;  // 0 free, 1 held: the LDS stage + tables are one resource the two groups take turns on
  int help_for;  // 0: none; g+1: group g is asked to help with the frame in `help`
  int idle[kGroups];
  FrameHdr hdr[kGroups];  // mailbox: the group's first wave fetches the next frame for the others
  HelpReq help;
};

__device__ __forceinline__ int lds_load(const int *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store(int *p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Work queue: frames beyond the first one per group are handed out dynamically (frame cost varies
// ~3x with the hand's size; a static 4-frames-per-CU split left a 25 % tail).  One slot per launch in
// flight; `next` and `done` return to 0 when the launch's last group leaves, so a slot needs no reset.
constexpr int kQueueSlots = 1024;  // launches that may be in flight at once (8 KiB of device memory)
__device__ unsigned int g_queue[kQueueSlots][2];

__device__ __forceinline__ void group_barrier(int *cnt, int &target) {
  target += kGWaves;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0) {
    __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target)
      __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}

// Phase 2 of the augmented form (oracle/tsdf_oracle.c::tsdf_oracle_voxels_aug): the voxel centre v'
// lives in the augmented frame, v = T^-1(v') is projected, the surface point w is mapped forward and the
// distances are taken between v' and T(w).  The projection no longer factorises, so there are no pixel
// tables and q = -F / v_z is one true division per voxel; what does factorise is the inverse map: its
// three products per row, A_i0*v'_x, A_i1*v'_y, A_i2*v'_z, depend on one grid index each and are
// tabulated per frame (atab, 9*R doubles), leaving ((a + b) + c) + d — the oracle's exact rounding order.
// atab layout: [axis][index][row] = fl(inv[4*row + axis] * (ori_axis + index*voxel_len)).
template <int LAYOUT, int T, class SrcP>
__device__ __forceinline__ void phase2_aug(const Grid &g, const CamK &cam, const VoxK &vk, int R,
                                           const double *xf, const double *atab, const SrcP src,
                                           const GlobalOut out, const int tid) {
  const double vl = (double)g.voxel_len;
  const double ox = (double)g.ori[0], oy = (double)g.ori[1], oz = (double)g.ori[2];
  const double *fwd = xf, *inv = xf + 12;
  const double bi0 = inv[3], bi1 = inv[7], bi2 = inv[11];
  const double *tabx = atab, *taby = atab + 3 * R, *tabz = atab + 6 * R;
  const int R4 = R / 4;
  const int G = R * R4;
  const int64_t R3 = (int64_t)R * R * R;
  int g0, gstep, s0, sstep;
  if (G <= T && (T % G) == 0) {
    g0 = tid % G;
    gstep = G;
    s0 = tid / G;
    sstep = T / G;
  } else {
    g0 = tid;
    gstep = T;
    s0 = 0;
    sstep = 1;
  }
  for (int gi = g0; gi < G; gi += gstep) {
    const int f4i = (gi % R4) * 4;
    const int y = gi / R4;
    const double vpy = oy + (double)y * vl;
    const double ty0 = taby[3 * y], ty1 = taby[3 * y + 1], ty2 = taby[3 * y + 2];
    for (int sl = s0; sl < R; sl += sstep) {
      // ---- project the 4 voxels and gather their depths ----
      int ex[4], ry[4];
      float pd[4];
      bool ok[4];
      double vpx[4], vpz[4], az[4], tz[4], q2[4];
      bool any_near = false;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int x = LAYOUT == 0 ? f4i + j : sl, z = LAYOUT == 0 ? sl : f4i + j;
        vpx[j] = ox + (double)x * vl;
        vpz[j] = oz + (double)z * vl;
        const double *tx = tabx + 3 * x, *tzp = tabz + 3 * z;
        const double vx = ((tx[0] + ty0) + tzp[0]) + bi0;                        // v = T^-1(v')
        const double vy = ((tx[1] + ty1) + tzp[1]) + bi1;
        const double vz = ((tx[2] + ty2) + tzp[2]) + bi2;
        const double q = -cam.focal / vz;                                        // :30
        ex[j] = project_rel(vx, q, cam.cx, vk.px0, vk.dx);                       // :31
        ry[j] = project_rel(-vy, q, cam.cy, vk.py0, vk.dy);                      // :32
        const bool inb = (ex[j] | ry[j]) >= 0;                                   // :36
        int idx = __mul24(ry[j], vk.stride) + ex[j] + vk.base;
        idx = inb ? idx : vk.base;
        pd[j] = src[idx];                                                        // :38-39
        ok[j] = inb & (__builtin_fabsf(pd[j]) >= vk.eps);                        // :40
      }
      // ---- z component first: a wave whose voxels are all beyond the truncation distance along z' alone
      // needs nothing else (dist >= |tz| > 1 -> (1,1,1)); one whose voxels are all rejected needs nothing ----
      double wx[4], wy[4];
      float o0[4], o1[4], o2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o0[j] = o1[j] = o2[j] = 1.0f;
        az[j] = vpz[j];
      }
      if (__any(ok[0] | ok[1] | ok[2] | ok[3])) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          q2[j] = div_by_focal((double)pd[j], cam);                                // :43
          wx[j] = ((double)(ex[j] + vk.px0) - cam.cx) * q2[j];                     // :44
          wy[j] = -((double)(ry[j] + vk.py0) - cam.cy) * q2[j];                    // :45
          az[j] = affine_row(fwd + 8, wx[j], wy[j], -(double)pd[j]);               // w'_z, w_z = -pd :46
          tz[j] = (vpz[j] - az[j]) * vk.it;                                        // :49
          any_near |= ok[j] & (__builtin_fabs(tz[j]) <= 1.0);
        }
        if (__any(any_near)) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const double wz = -(double)pd[j];
            const double ax = affine_row(fwd + 0, wx[j], wy[j], wz);
            const double ay = affine_row(fwd + 4, wx[j], wy[j], wz);
            const double tx = (vpx[j] - ax) * vk.it, ty = (vpy - ay) * vk.it;      // :47-48
            const double s2 = __builtin_fma(tz[j], tz[j], __builtin_fma(ty, ty, tx * tx));
            const bool nearv = s2 <= 1.0;                                          // :54
            const float m0 = vmin(__builtin_fabsf((float)tx), 1.0f);
            const float m1 = vmin(__builtin_fabsf((float)ty), 1.0f);
            const float m2 = vmin(__builtin_fabsf((float)tz[j]), 1.0f);
            o0[j] = nearv ? m0 : 1.0f;
            o1[j] = nearv ? m1 : 1.0f;
            o2[j] = nearv ? m2 : 1.0f;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned keep = ok[j] ? 0xffffffffu : 0u;
        const unsigned sg = (az[j] > vpz[j] ? 0x80000000u : 0u) & keep;         // w'_z > v'_z  :65
        o0[j] = __uint_as_float((__float_as_uint(o0[j]) & keep) | sg);
        o1[j] = __uint_as_float((__float_as_uint(o1[j]) & keep) | sg);
        o2[j] = __uint_as_float((__float_as_uint(o2[j]) & keep) | sg);
      }
      const int64_t e = ((int64_t)sl * R + y) * R + f4i;  // o[c][slow][y][fast]
      store_vol4(out + e, f4{o0[0], o0[1], o0[2], o0[3]});
      store_vol4(out + R3 + e, f4{o1[0], o1[1], o1[2], o1[3]});
      store_vol4(out + 2 * R3 + e, f4{o2[0], o2[1], o2[2], o2[3]});
    }
  }
}

// Persistent kernel, one 1024-thread workgroup per CU.  Its two 512-thread groups each walk their
// own frames  (frame = blockIdx.x + gridDim.x * (group + 2*i))  through
//     stream rows -> extents -> glue        (no shared resource; the memory-bound part)
//     [ tables -> stage -> phase 2 ]        (holds the LDS stage; the VALU/store-bound part)
// and take turns on the single 128 KiB LDS stage, so one group's row streaming overlaps the other
// group's voxel arithmetic and stores on the same CU.
template <int RT, int LAYOUT, bool AUG>
__global__ __launch_bounds__(kWG) void tsdf_fused_kernel(
    const float *__restrict__ depth, const int64_t *__restrict__ offsets,
    const int32_t *__restrict__ headers, int n, int Rrt, CamK cam, float *__restrict__ out_tsdf,
    float *__restrict__ out_max_l, float *__restrict__ out_mid_p, int32_t *__restrict__ out_status,
    float *__restrict__ out_aabb, float *__restrict__ out_grid, float *__restrict__ out_ori,
    int aabb_only, const float *__restrict__ grid_in, int qslot, const double *__restrict__ xforms,
    int64_t depth_len) {
  __shared__ __attribute__((aligned(16))) float stage[kStageFloats];
  __shared__ __attribute__((aligned(16))) int pxtab[kTabR * kTabR];
  __shared__ __attribute__((aligned(16))) int pytab[kTabR * kTabR];
  __shared__ __attribute__((aligned(16))) ZEntry ztab[kMaxR];
  __shared__ __attribute__((aligned(16))) double atab[AUG ? 9 * kMaxR : 1];  // inverse-map products (AUG)
  __shared__ float red_all[kGroups][kGWaves * kRedStride];
  __shared__ GroupCtl ctl;

  const int R = RT ? RT : Rrt;
  const int tid = threadIdx.x, lane = tid & 63, gtid = tid & (kGW - 1);
  const int group = __builtin_amdgcn_readfirstlane(tid / kGW);
  const int gwave = __builtin_amdgcn_readfirstlane((tid >> 6) & (kGWaves - 1));
  float *red = red_all[group];

  if (tid == 0) {
    for (int i = 0; i < kGroups; ++i) ctl.bar[i] = ctl.idle[i] = 0;
    ctl.lock = 0;
    ctl.help_for = 0;
  }
  __syncthreads();  // the only workgroup-wide barrier
  int bar_target = 0;
  auto gsync = [&]() { group_barrier(&ctl.bar[group], bar_target); };

  unsigned int *q_next = &g_queue[qslot][0], *q_done = &g_queue[qslot][1];
  const int n_static = gridDim.x * kGroups;  // frames handed out by position (the first one per group)

  int iter = 0;
  (void)iter;
  for (;; ++iter) {
    // ---- the group's first wave fetches the next frame (index + header) and posts it in LDS ----
    if (gwave == 0) {
      int fr;
      if (iter == 0) {
        fr = blockIdx.x + gridDim.x * group;
      } else {
        unsigned int t = 0;
        if (lane == 0) t = atomicAdd(q_next, 1u);
        fr = n_static + (int)__builtin_amdgcn_readfirstlane(t);
      }
      FrameHdr m;
      m.frame = fr < n ? fr : -1;
      m.l = m.t = m.r = m.b = m.pad = 0;
      m.off0 = m.off1 = 0;
      if (fr < n) {
        const int32_t *h = headers + 6 * (int64_t)fr;
        m.l = h[2];
        m.t = h[3];
        m.r = h[4];
        m.b = h[5];
        m.off0 = offsets[fr];
        m.off1 = offsets[fr + 1];
      }
      if (lane == 0) ctl.hdr[group] = m;
    }
    gsync();
    const FrameHdr fh = ctl.hdr[group];
    const int frame = __builtin_amdgcn_readfirstlane(fh.frame);
    if (frame < 0) break;
    TSDF_STAMP(kGroups * iter + group, 0);

    Frame f;
    f.l = __builtin_amdgcn_readfirstlane(fh.l);
    f.t = __builtin_amdgcn_readfirstlane(fh.t);
    f.r = __builtin_amdgcn_readfirstlane(fh.r);
    f.b = __builtin_amdgcn_readfirstlane(fh.b);
    f.bw = f.r - f.l;
    f.bh = f.b - f.t;
    f.depth = depth + fh.off0;
    float *out = out_tsdf ? out_tsdf + (int64_t)frame * 3 * R * R * R : nullptr;
    const bool want_vol = !aabb_only && out;

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

    bool holds_stage = false;  // group-uniform
    // a header that contradicts its payload, or a payload outside the depth buffer, is never read
    if (f.bw <= 0 || f.bh <= 0 || (int64_t)f.bw * (int64_t)f.bh != fh.off1 - fh.off0 || fh.off0 < 0 ||
        fh.off1 > depth_len) {
      status = TSDF_FRAME_BAD_HEADER;  // group-uniform
    } else {
      float fin[kExt];
      // The group's first wave takes the stage lock on its way into the extents barrier, so the
      // wait for the other group's phase 2 hides behind this group's own slowest wave.
      auto sync_and_lock = [&]() {
        if (want_vol && gwave == 0 && lane == 0) {
          while (atomicCAS(&ctl.lock, 0, 1) != 0) __builtin_amdgcn_s_sleep(2);
        }
        gsync();
      };
      phase1_extents<kGWaves, AUG>(f, cam, 0, f.bh, red, fin, gwave, sync_and_lock, kGroups * iter + group,
                                   AUG ? xforms + 24 * (int64_t)frame : nullptr);
      holds_stage = want_vol;
      ab = aabb_from_extents(fin);
      TSDF_STAMP(kGroups * iter + group, 4);
      if (!ab.any) {
        status = TSDF_FRAME_DEGENERATE;
        ab.mn[0] = ab.mn[1] = ab.mn[2] = ab.mx[0] = ab.mx[1] = ab.mx[2] = 0.f;
      } else {
        g = glue(ab.mn, ab.mx, R, cam);
        if (grid_in) {
          // caller-supplied placement (tsdf_cal's vox_ori, voxel_len, truncation arguments)
          const float *gi = grid_in + 8 * (int64_t)frame;
          g.ori[0] = gi[0];
          g.ori[1] = gi[1];
          g.ori[2] = gi[2];
          g.voxel_len = gi[3];
          g.trunc = gi[4];
          if (!(g.trunc > 0.f) || !(g.trunc < TSDF_INF)) status = TSDF_FRAME_DEGENERATE;
        } else if (!(g.max_l > 0.f) || !(g.max_l < TSDF_INF)) {
          status = TSDF_FRAME_DEGENERATE;
          g.max_l = g.voxel_len = g.trunc = 0.f;
        }
      }
    }

    if (gtid == 0) {
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

    bool ran_phase2 = false;
    if (want_vol) {
      if (status != TSDF_FRAME_OK) {
        zero_volume(out, R, gtid);
      } else {
        ran_phase2 = true;
        // The thread's index for the tables and the voxel pass, hidden from loop-invariant code motion:
        // otherwise every constant derived from it (a dozen float64 conversions of voxel indices) is
        // computed once before the frame loop and then occupies registers, or scratch, all through phase 1.
        int vt = gtid;
        asm volatile("" : "+v"(vt));
        VoxK vk;
        vk.cx = cam.cx;
        vk.cy = cam.cy;
        vk.it = 1.0 / (double)g.trunc;
        vk.kq = cam.inv_focal * vk.it;
        vk.ncx = -cam.cx;
        vk.eps = cam.eps;
        vk.px0 = f.l + ab.c0;
        vk.py0 = f.t + ab.r0;
        vk.dx = ab.c1 - ab.c0;
        vk.dy = ab.r1 - ab.r0;
        // LDS image of the rectangle: rows padded to a multiple of 4 pixels (16-byte LDS-DMA pieces)
        const int sw = vk.dx + 1, sh = vk.dy + 1;
        const int sw4 = (sw + 3) & ~3;
        const bool staged = (int64_t)sw4 * sh <= kStageFloats;  // group-uniform

        // ---- stage the rectangle of valid pixels into LDS by LDS-DMA (global_load_lds_dwordx4) ----
        // No VGPR staging and no ds_write pass: each wave instruction moves up to 64 x 16 B straight into
        // the row-major LDS image (lane i lands at base + 16*i, so lanes are laid out as [row][4-pixel
        // group]); all of a wave's pieces are in flight at once.  Sources need only 4-byte alignment
        // and EXEC-masked lanes leave their slot untouched (tools/probes/glds_probe.hip).
        if (staged) {
          const int ng = sw4 >> 2;                       // 4-pixel groups per row
          const int64_t n_frame = fh.off1 - fh.off0;     // elements in this frame's crop
          const int64_t base_idx = (int64_t)ab.r0 * f.bw + ab.c0;
          if (ng <= 64) {
            const int rows_per = 64 / ng;                // rows one wave instruction covers
            const int rsub = lane / ng, cg = lane - rsub * ng;
            const int nblk = (sh + rows_per - 1) / rows_per;
            for (int blk = gwave; blk < nblk; blk += kGWaves) {
              const int R0 = blk * rows_per;             // scalar
              const int row = R0 + rsub;
              const int64_t gi = base_idx + (int64_t)row * f.bw + 4 * cg;
              const bool act = rsub < rows_per && row < sh;
              float *ldst = stage + R0 * sw4;            // wave-uniform LDS base
              if (act && gi + 3 < n_frame) {
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(f.depth + gi),
                    (__attribute__((address_space(3))) void *)ldst, 16, 0, 0);
              } else if (act) {  // the 16-byte piece would run past the end of the frame: element copies
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (gi + e < n_frame) ldst[(rsub * ng + cg) * 4 + e] = f.depth[gi + e];
              }
            }
          } else {
            for (int row = gwave; row < sh; row += kGWaves) {
              for (int c4 = 0; c4 < ng; c4 += 64) {
                const int cg = c4 + lane;
                const int64_t gi = base_idx + (int64_t)row * f.bw + 4 * cg;
                float *ldst = stage + row * sw4 + 4 * c4;  // wave-uniform
                if (cg < ng && gi + 3 < n_frame) {
                  __builtin_amdgcn_global_load_lds(
                      (const __attribute__((address_space(1))) void *)(f.depth + gi),
                      (__attribute__((address_space(3))) void *)ldst, 16, 0, 0);
                } else if (cg < ng) {
#pragma unroll
                  for (int e = 0; e < 4; ++e)
                    if (gi + e < n_frame) ldst[lane * 4 + e] = f.depth[gi + e];
                }
              }
            }
          }
        }
        // ---- per-frame tables (true divisions; (x,z)/(y,z) pairs spread over the group) ----
        const double vl = (double)g.voxel_len;
        const double ox = (double)g.ori[0], oy = (double)g.ori[1], oz = (double)g.ori[2];
        TSDF_STAMP(kGroups * iter + group, 5);
        if (vt < R) {
          const double v_z = oz + (double)vt * vl;  // :28
          ZEntry ze;
          ze.q = -cam.focal / v_z;                    // :30
          ze.vzs = v_z * vk.it;
          ze.negthr = f32_round_up(-v_z);             // pd < -v_z  <=>  w_z > v_z  (:65)
          ze.pad = 0.f;
          ztab[vt] = ze;
        }
        if constexpr (AUG) {
          // products of the inverse map, one per (axis, index, row): see phase2_aug
          const double *inv = xforms + 24 * (int64_t)frame + 12;
          for (int e = vt; e < 9 * R; e += kGW) {
            const int axis = e / (3 * R), rem = e - axis * 3 * R, i = rem / 3, row = rem - 3 * i;
            const double o_a = axis == 0 ? ox : (axis == 1 ? oy : oz);
            atab[e] = inv[4 * row + axis] * (o_a + (double)i * vl);
          }
        }
        const bool use_tab = !AUG && R <= kTabR;  // uniform
        if (use_tab) {
          for (int e = vt; e < R * R; e += kGW) {
            const int z = e / R, i = e - z * R;
            const double q = -cam.focal / (oz + (double)z * vl);                              // :30
            const double vx = ox + (double)i * vl, vy = oy + (double)i * vl;                  // :26-27
            pxtab[tab_index<LAYOUT>(i, z, R)] = project_rel(vx, q, cam.cx, vk.px0, vk.dx);    // :31
            pytab[tab_index<LAYOUT>(i, z, R)] = project_rel(-vy, q, cam.cy, vk.py0, vk.dy);   // :32
          }
        }
        TSDF_STAMP(kGroups * iter + group, 6);

        // the copy was issued before the tables were computed (worth 1.1 % of the launch, tools/ab_precise.py)
        if (staged) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // LDS-DMA completion is counted in vmcnt
        TSDF_STAMP(kGroups * iter + group, 7);
        if constexpr (kGroups == 2 && TSDF_TAIL_HELP) {
          if (gwave == 0 && lane == 0) ctl.hdr[group].pad = lds_load(&ctl.idle[group ^ 1]);
        }
        gsync();
        TSDF_STAMP(kGroups * iter + group, 8);
        vk.stride = staged ? sw4 : f.bw;
        vk.base = staged ? 0 : ab.r0 * f.bw + ab.c0;
        bool helped = false;  // group-uniform
        if constexpr (kGroups == 2 && TSDF_TAIL_HELP) {
          // the other group is idle (so this is the launch's last frame on this CU): split the volume with it
          if (__builtin_amdgcn_readfirstlane(ctl.hdr[group].pad)) {
            helped = true;
            if (gwave == 0 && lane == 0) {
              HelpReq hq;
              hq.g = g;
              hq.vk = vk;
              hq.src = f.depth;
              hq.staged = staged;
              hq.pad = 0;
              hq.out = out;
              hq.frame = frame;
              hq.use_tab = use_tab;
              ctl.help = hq;
              lds_store(&ctl.help_for, (group ^ 1) + 1);
            }
          }
        }
        TSDF_STAMP_VAL(kGroups * iter + group, 10, helped ? 2 : 1);
        auto run2 = [&](auto src) {
          if (__builtin_expect(helped, 0)) {
            if constexpr (AUG) {
              phase2_aug<LAYOUT, 2 * kGW>(g, cam, vk, R, xforms + 24 * (int64_t)frame, atab, src, (GlobalOut)out, vt);
            } else {
              phase2<LAYOUT, 2 * kGW>(g, cam, vk, R, ztab, pxtab, pytab, use_tab, src, (GlobalOut)out, vt);
            }
          } else {
            if constexpr (AUG) {
              phase2_aug<LAYOUT, kGW>(g, cam, vk, R, xforms + 24 * (int64_t)frame, atab, src, (GlobalOut)out, vt);
            } else {
              phase2<LAYOUT, kGW>(g, cam, vk, R, ztab, pxtab, pytab, use_tab, src, (GlobalOut)out, vt);
            }
          }
        };
        if (staged) {
          run2((LdsSrc)stage);
        } else {
          run2((GlobalSrc)f.depth);
        }
      }
    }
    TSDF_STAMP(kGroups * iter + group, 9);
    // Close the frame: every wave of the group has left the LDS it shares (stage/tables when phase 2
    // ran, `red` and the header mailbox always) before the lock is handed over and the next frame
    // rewrites them.
    gsync();
    (void)ran_phase2;
    if (holds_stage && gwave == 0 && lane == 0)
      __hip_atomic_store(&ctl.lock, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  // ---- queue empty: offer help with the other group's last frame before leaving ----
  if constexpr (kGroups == 2 && TSDF_TAIL_HELP) {
    // Queue empty: wait until the other group either asks for help with its last frame or is idle too.
    // (help_for is written before idle[] by the same wave, and LDS operations of a wave stay in order:
    // once idle[other] reads 1, help_for is final.)
    if (gwave == 0 && lane == 0) {
      lds_store(&ctl.idle[group], 1);
      int dec;
      for (;;) {
        if (lds_load(&ctl.help_for) == group + 1) { dec = 1; break; }
        if (lds_load(&ctl.idle[group ^ 1])) { dec = lds_load(&ctl.help_for) == group + 1; break; }
        __builtin_amdgcn_s_sleep(8);
      }
      ctl.hdr[group].pad = dec;
    }
    gsync();
    if (__builtin_amdgcn_readfirstlane(ctl.hdr[group].pad)) {
      const HelpReq hq = ctl.help;
      // pointers that came through LDS are generic: say that they are global, or every access through
      // them is a FLAT instruction (counted in lgkmcnt as well as vmcnt)
      const GlobalOut hout = (GlobalOut)hq.out;
      const GlobalSrc hsrc = (GlobalSrc)hq.src;
      auto run2 = [&](auto src) {
        if constexpr (AUG) {
          phase2_aug<LAYOUT, 2 * kGW>(hq.g, cam, hq.vk, R, xforms + 24 * (int64_t)hq.frame, atab, src, hout,
                                      kGW + gtid);
        } else {
          phase2<LAYOUT, 2 * kGW>(hq.g, cam, hq.vk, R, ztab, pxtab, pytab, hq.use_tab != 0, src, hout, kGW + gtid);
        }
      };
      if (hq.staged) {
        run2((LdsSrc)stage);
      } else {
        run2(hsrc);
      }
    }
  }
  // ---- leave: the last group of the launch returns the queue slot to its initial state ----
  if (gwave == 0 && lane == 0) {
    const unsigned int d = atomicAdd(q_done, 1u);
    if (d + 1 == (unsigned int)(gridDim.x * kGroups)) {
      __hip_atomic_store(q_next, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(q_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

const tsdf_cam kDefaultCam = {241.42, 160.0, 120.0, 1.0f, 3.0f};

// CU count of the current device (cached per device id; a racing first call computes the same value)
int num_cus() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cached[dev] = v;
  }
  return cached[dev];
}

// The code object holds gfx950 kernels only: any other device is "no usable device", not a launch error.
// (Cached per device id; a racing first call computes the same value.)
int check_device() {
  static int arch_state[64] = {0};  // 0 unknown, 1 gfx950, -1 something else
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return TSDF_ERR_NO_DEVICE;
  }
  if (dev < 0 || dev >= 64) return TSDF_OK;  // beyond the cache: let the launch decide
  if (arch_state[dev] == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      (void)hipGetLastError();
      return TSDF_ERR_NO_DEVICE;
    }
    arch_state[dev] = strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : -1;
  }
  return arch_state[dev] == 1 ? TSDF_OK : TSDF_ERR_NO_DEVICE;
}

// everything a launch needs, so that the dispatch over (R, layout, augmented) stays in one place
struct LaunchArgs {
  const float *depth;
  const int64_t *offsets;
  const int32_t *headers;
  int n, R;
  CamK ck;
  float *tsdf, *max_l, *mid_p;
  int32_t *status;
  float *aabb, *grid, *ori;
  int aabb_only;
  const float *grid_in;
  const double *xforms;
  int64_t depth_len;
};

template <int RT, int LAYOUT, bool AUG>
hipError_t launch(hipStream_t s, const LaunchArgs &a) {
  // persistent: one workgroup per CU; with fewer than kGroups frames per CU the later groups idle
  const int grid = a.n < num_cus() ? a.n : num_cus();
  static std::atomic<unsigned int> launch_counter{0};
  const int qslot = (int)(launch_counter.fetch_add(1, std::memory_order_relaxed) % kQueueSlots);
  hipLaunchKernelGGL((tsdf_fused_kernel<RT, LAYOUT, AUG>), dim3(grid), dim3(kWG), 0, s, a.depth, a.offsets,
                     a.headers, a.n, a.R, a.ck, a.tsdf, a.max_l, a.mid_p, a.status, a.aabb, a.grid, a.ori,
                     a.aabb_only, a.grid_in, qslot, a.xforms, a.depth_len);
  return hipGetLastError();
}

template <int LAYOUT, bool AUG>
hipError_t launch_r(hipStream_t s, const LaunchArgs &a) {
  if (a.R == 32) return launch<32, LAYOUT, AUG>(s, a);
  if (a.R == 64) return launch<64, LAYOUT, AUG>(s, a);
  return launch<0, LAYOUT, AUG>(s, a);
}

int run(const float *d_depth, int64_t depth_len, const int64_t *d_offsets, const int32_t *d_headers, int n, int R,
        const tsdf_cam *cam, int layout, void *hip_stream, float *t, float *ml, float *mp, int32_t *st,
        float *ab, float *gr, float *orr, int aabb_only, const float *gin = nullptr,
        const double *xforms = nullptr) {
  if (n < 0 || !tsdf_resolution_supported(R)) return TSDF_ERR_INVALID_ARG;
  if (layout != TSDF_LAYOUT_CZYX && layout != TSDF_LAYOUT_CXYZ) return TSDF_ERR_INVALID_ARG;
  if (n == 0) return TSDF_OK;
  if (!d_depth || !d_offsets || !d_headers || depth_len < 0) return TSDF_ERR_INVALID_ARG;
  if (!aabb_only && (!t || (reinterpret_cast<uintptr_t>(t) & 15))) return TSDF_ERR_INVALID_ARG;
  if (!cam) cam = &kDefaultCam;
  if (!(cam->focal > 0.0) || !(cam->invalid_eps > 0.0f) || !(cam->trunc_voxels > 0.0f))
    return TSDF_ERR_INVALID_ARG;
  int rc = check_device();
  if (rc != TSDF_OK) return rc;
  LaunchArgs a;
  a.depth = d_depth;
  a.offsets = d_offsets;
  a.headers = d_headers;
  a.n = n;
  a.R = R;
  a.ck.focal = cam->focal;
  a.ck.cx = cam->cx;
  a.ck.cy = cam->cy;
  a.ck.inv_focal = 1.0 / cam->focal;
  a.ck.eps = cam->invalid_eps;
  a.ck.trunc_vox = cam->trunc_voxels;
  a.tsdf = t;
  a.max_l = ml;
  a.mid_p = mp;
  a.status = st;
  a.aabb = ab;
  a.grid = gr;
  a.ori = orr;
  a.aabb_only = aabb_only;
  a.grid_in = gin;
  a.xforms = xforms;
  a.depth_len = depth_len;
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  hipError_t e;
  if (xforms) {
    e = layout == TSDF_LAYOUT_CZYX ? launch_r<0, true>(s, a) : launch_r<1, true>(s, a);
  } else {
    e = layout == TSDF_LAYOUT_CZYX ? launch_r<0, false>(s, a) : launch_r<1, false>(s, a);
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

int tsdf_voxelize_hip(const float *d_depth, int64_t depth_len, const int64_t *d_offsets, const int32_t *d_headers, int n,
                      int R, const tsdf_cam *cam, int layout, void *hip_stream, float *d_out_tsdf,
                      float *d_out_max_l, float *d_out_mid_p, int32_t *d_out_status) {
  if (n > 0 && (!d_out_tsdf || !d_out_max_l || !d_out_mid_p)) return TSDF_ERR_INVALID_ARG;
  return run(d_depth, depth_len, d_offsets, d_headers, n, R, cam, layout, hip_stream, d_out_tsdf, d_out_max_l,
             d_out_mid_p, d_out_status, nullptr, nullptr, nullptr, 0);
}

int tsdf_voxelize_grid_hip(const float *d_depth, int64_t depth_len, const int64_t *d_offsets, const int32_t *d_headers, int n,
                           int R, const tsdf_cam *cam, int layout, void *hip_stream, const float *d_grid,
                           float *d_out_tsdf, int32_t *d_out_status) {
  if (n > 0 && (!d_out_tsdf || !d_grid)) return TSDF_ERR_INVALID_ARG;
  return run(d_depth, depth_len, d_offsets, d_headers, n, R, cam, layout, hip_stream, d_out_tsdf, nullptr, nullptr,
             d_out_status, nullptr, nullptr, nullptr, 0, d_grid);
}

int tsdf_voxelize_aug_hip(const float *d_depth, int64_t depth_len, const int64_t *d_offsets, const int32_t *d_headers, int n,
                          int R, const tsdf_cam *cam, int layout, void *hip_stream, const double *d_xforms,
                          float *d_out_tsdf, float *d_out_max_l, float *d_out_mid_p, int32_t *d_out_status) {
  if (n > 0 && (!d_out_tsdf || !d_out_max_l || !d_out_mid_p || !d_xforms)) return TSDF_ERR_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(d_xforms) & 7) return TSDF_ERR_INVALID_ARG;
  return run(d_depth, depth_len, d_offsets, d_headers, n, R, cam, layout, hip_stream, d_out_tsdf, d_out_max_l,
             d_out_mid_p, d_out_status, nullptr, nullptr, nullptr, 0, nullptr, d_xforms);
}

int tsdf_aabb_hip(const float *d_depth, int64_t depth_len, const int64_t *d_offsets, const int32_t *d_headers, int n, int R,
                  const tsdf_cam *cam, void *hip_stream, float *d_out_aabb, float *d_out_grid,
                  float *d_out_ori, int32_t *d_out_status) {
  return run(d_depth, depth_len, d_offsets, d_headers, n, R, cam, TSDF_LAYOUT_CZYX, hip_stream, nullptr, nullptr,
             nullptr, d_out_status, d_out_aabb, d_out_grid, d_out_ori, 1);
}

#ifdef TSDF_STAMPS
// Diagnostic library only: copy the stamp array to the host (synchronises the device).
int tsdf_debug_read_stamps(unsigned long long *host_out, int count) {
  const int total = kStampBlocks * kStampFrames * kStampSlots;
  if (count > total) count = total;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * count) != hipSuccess)
    return -1;
  return count;
}
#endif

}  // extern "C"
