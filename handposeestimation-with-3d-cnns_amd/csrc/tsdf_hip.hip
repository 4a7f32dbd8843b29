// tsdf_hip.hip — fused projective-TSDF voxelizer for gfx950 (MI355X), and its C ABI.
//
// Replaces, for a whole batch in ONE launch, what the reference does per frame with two
// numba kernels, host numpy glue and four PCIe copies (pre/tsdf_numba.py:119-161):
//   phase 1  min_max_kernel  (pre/tsdf_numba.py:75-116,140-141)  AABB of all valid pixels
//   glue     host numpy      (pre/tsdf_numba.py:142-147)          grid placement, float32
//   phase 2  tsdf_kernel     (pre/tsdf_numba.py:15-72)            per-voxel project/gather/TSDF
//   labels   joint_nor       (pre/joint_nor.py:8-18, 3D_CNN/train.py:236-244)  optional, same launch
// Arithmetic contract: SURVEY.md Appendix A (float32 parameters, float64 intermediates,
// unfused multiply-then-add for the pixel index, float32 store).
//
// Design (DESIGN.md has the numbers):
//   * one persistent launch; one 1024-thread workgroup (16 wave64) per CU, which owns the CU's LDS.
//     Its two 512-thread halves ("groups") each process whole frames — the first one positional, the
//     rest from a work queue — and take turns on the single LDS pool, so one group's row streaming
//     (memory-bound) overlaps the other group's voxel arithmetic and stores on the same CU.  Groups
//     synchronise on LDS counters (s_barrier would span both).  The AABB and the grid placement never
//     leave the chip;
//   * phase 1 streams the crop once with vector loads, lane <-> P consecutive columns, wave <-> rows,
//     two register buffers in ping-pong behind counted vmcnt waits.  It does NOT back-project every
//     pixel (one float64 division each): f32(f64(d)/F * (x-cx)) is monotone in d for a fixed column x
//     (and likewise per row), so the AABB is the extreme of the formula applied to each column's /
//     row's (min,max) valid depth — bit-identical result, 2(b_w+b_h) evaluations per wave instead of
//     b_w*b_h; d/F uses Markstein's correction (exact quotient in 3 ops).  The same pass yields the pixel
//     rectangle that holds every valid pixel;
//   * staging (TSDF_FILL 0, the default): that rectangle (the only pixels phase 2 can ever use —
//     everything outside it is rejected by pre/tsdf_numba.py:36 or :40) is copied into the LDS pool by
//     LDS-DMA (global_load_lds_dwordx4), so the per-voxel gather is an LDS read: no vector-memory
//     latency, and stores never block loads; a rectangle larger than the pool gathers from global
//     memory (L2).  The alternative, capturing every row's span of valid pixels into the pool while the
//     row is in registers (TSDF_FILL 1: depth read from HBM exactly once, no lock), is implemented and
//     parity-green but measured slower on every workload; the split kernel uses it;
//   * phase 2: pix_x depends on (x,z) only and pix_y on (y,z) only -> both tabulated per frame in LDS
//     (true division for q = -F/v_z, unfused multiply-add, v_cvt_i32_f64 truncation); the y table holds
//     the row's pool entry directly.  Each lane owns 4 consecutive voxels along the layout's fastest
//     axis, so every wave store is 1 KiB contiguous (global_store_dwordx4 nt: written once, never
//     re-read).  The per-voxel chain uses reciprocals (<= a few ulp64 from the divisions it replaces —
//     10 orders of magnitude inside the 1e-5 parity bound); a wave whose 256 voxels are all rejected or
//     farther than the truncation distance along z skips the x/y terms (the result is then
//     (+-1,+-1,+-1) or 0 by pre/tsdf_numba.py:54-57);
//   * small batches (n <= CUs/2) take the split kernel instead: S workgroups per frame, each streams
//     the frame (16 waves, L2 hits after the first) and voxelizes 1/S of the slow axis — no
//     inter-workgroup communication, bit-identical results;
//   * the augmented form (template AUG) maps every valid pixel / voxel centre / surface point through a
//     per-frame affine transform instead (tsdf_voxelize_aug_hip, re-specified: see include/tsdf.h).
// HBM-bound streaming read + streaming write.  No MFMA (gather/scatter, not a contraction), no
// inter-workgroup communication (XCD placement is irrelevant), no CPU fallback, gfx950 only.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <thread>
#include <type_traits>

#include "../../include/tsdf.h"
#include "tsdf_host.inc"   // the host-only part (also compiled alone, under sanitizers)

namespace {

constexpr int kWG = 1024;               // threads per workgroup (16 wave64; needs <= 128 VGPRs)
// Groups per workgroup: each walks its own frames (see the kernel).  A template parameter of the fused kernel, chosen
// per resolution (groups_for): two 512-thread groups at 32^3 — one's row stream hides behind the other's voxel pass —,
// ONE 1024-thread group from 48^3 upwards, where the voxel pass is 8x the row stream and what counts is that a CU
// writes each channel plane of its volume as one sequential run (paired A/B on 1024 full frames -> 64^3: plain -3.7 %,
// augmented -4.3 %; at 32^3 one group is +8 % on full frames, +13 % on crops).  -DTSDF_GROUPS=g forces one value
// everywhere (experiments).  The names below are the defaults of code outside the fused kernel (stamps; the split
// kernel's LDS layout); inside it they are shadowed by the instantiation's own values.
#ifdef TSDF_GROUPS
constexpr int kGroupsForced = TSDF_GROUPS;
#else
constexpr int kGroupsForced = 0;
#endif
constexpr int kGroups = kGroupsForced ? kGroupsForced : 2;
constexpr int kMaxGroups = kGroups > 2 ? kGroups : 2;
constexpr int kGW = kWG / kGroups;      // threads per group
[[maybe_unused]] constexpr int kGWaves = kGW / 64;  // waves per group
constexpr int groups_for(int R) { return kGroupsForced ? kGroupsForced : (R >= 48 ? 1 : 2); }
constexpr int kMaxR = 128;
constexpr int kTabR = 32;               // projection tables for R <= kTabR (2 + 4 KiB per group)
constexpr int kRedStride = 12;
constexpr int kLdsBytes = 160 * 1024;   // LDS of one gfx950 CU; the workgroup takes all of it
constexpr int kMaxRows = 256;           // rows per frame the row table holds (MSRA: 240)
constexpr int kMaxCapW = 320;           // widest bounding box whose rows are captured (= one row pass)

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
// per-wave stamps: every wave's lane 0 records when it enters (which = 0) and leaves (1) the voxel pass
__device__ unsigned long long g_wstamps[kStampBlocks * kStampFrames * 16 * 2];
#define TSDF_WSTAMP(iter, which)                                                                     \
  do {                                                                                               \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < kStampBlocks && (iter) < kStampFrames)               \
      g_wstamps[((blockIdx.x * kStampFrames + (iter)) * 16 + (threadIdx.x >> 6)) * 2 + (which)] =    \
          __builtin_amdgcn_s_memrealtime();                                                          \
  } while (0)
#else
#define TSDF_STAMP(iter, slot) do { } while (0)
#define TSDF_STAMP_VAL(iter, slot, val) do { } while (0)
#define TSDF_WSTAMP(iter, which) do { } while (0)
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
// min(|a|, b): the absolute value is a source modifier, not an instruction
__device__ __forceinline__ float vminabs(float a, float b) {
  float r;
  asm("v_min_f32 %0, |%1|, %2" : "=v"(r) : "v"(a), "v"(b));
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

// -F / vz for the augmented projection (a true division: the divisor changes from voxel to voxel).
//   FAST = false: the compiler's IEEE division (2 v_div_scale, v_rcp, 4 fma, mul, fma, v_div_fmas, v_div_fixup).
//   FAST = true:  the same sequence without its three scaling / fix-up instructions.  Those are the identity while
//                 neither operand nor the quotient comes near the ends of the exponent range, so the result is the
//                 same bit for bit; the caller proves 2^-600 < |vz| < 2^600 for a lane's whole column of voxels and
//                 2^-100 < F < 2^100 before choosing this form (phase2_aug), anything else takes the division.
// (TSDF_RCP_ONE_STEP=1, a measurement build only: v_rcp_f64 is good to 2^-24.4 on gfx950 (tools/probes/rcp64_probe.hip), so
// ONE Newton step leaves 2^-48.7 and the corrected quotient is off by ~2^-97 before its final rounding — the IEEE
// quotient on every one of 4.2 M probed divisors, but a quotient of two float64 can sit within 2^-107 of a rounding
// boundary, so only the second step makes the result provably the division's.  It stays.)
#ifndef TSDF_RCP_ONE_STEP
#define TSDF_RCP_ONE_STEP 0
#endif
template <bool FAST>
__device__ __forceinline__ double neg_focal_over(double vz, const CamK &k) {
  if constexpr (!FAST) {
    return -k.focal / vz;
  } else {
    const double n = -k.focal;
    double r = __builtin_amdgcn_rcp(vz);
    r = __builtin_fma(r, __builtin_fma(-vz, r, 1.0), r);
#if !TSDF_RCP_ONE_STEP
    r = __builtin_fma(r, __builtin_fma(-vz, r, 1.0), r);
#endif
    const double q0 = n * r;
    return __builtin_fma(__builtin_fma(-vz, q0, n), r, q0);
  }
}
__device__ __forceinline__ bool mid_range(double v, double lo, double hi) {  // false for NaN
  const double a = __builtin_fabs(v);
  return a > lo && a < hi;
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

#ifndef TSDF_ROW_LOAD_NT
#define TSDF_ROW_LOAD_NT 0
#endif
#ifndef TSDF_DMA_AUX
#define TSDF_DMA_AUX 0   // cache-policy bits of the staging copy (gfx940+: 1 = sc0, 2 = nt, 16 = sc1)
#endif
template <int P>
__device__ __forceinline__ PixN<P> load_pix(const float *__restrict__ p) {
  PixN<P> r;
  constexpr int Q = P / 4, T = P % 4;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
#if TSDF_ROW_LOAD_NT
    const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4u *>(p + 4 * q));
#else
    const f4 v = *reinterpret_cast<const f4u *>(p + 4 * q);
#endif
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


// ---- row-span capture (the split kernel; the fused kernel under TSDF_FILL 1) ------------------------
// While phase 1 has a row in registers it copies the row's window of valid pixels — the lane windows from
// the first to the last lane holding a valid pixel, cut at the row end — into an LDS pool and records
// where: one 32-bit entry per bbox row,
//     bits 31..18  offset in the pool, in units of 4 floats
//     bits 17..9   bbox-relative column of the first captured pixel
//     bits  8..0   number of captured pixels (0: the row has no valid pixel)
// A pixel (col, row) is then  pool[4*off4 + col - first]  when  0 <= col - first < cnt, and rejected by
// pre/tsdf_numba.py:40 otherwise (every pixel outside the window is invalid by construction).
// Every wave of the workgroup owns a fixed 1/16 of the pool and fills it with a bump pointer it keeps in a
// scalar register: allocation costs no LDS round trip (an LDS atomic with return was tried first and cost
// 3-7 us per frame: its latency is the LDS queue, which the other group's voxel pass keeps full).  Rows are
// dealt to the waves round-robin, so the waves of a frame fill up evenly.  A row that does not fit makes the
// frame "not captured": its voxel pass gathers from global memory instead (never wrong, only slower).
// (LDS pointers carry their address space in the type: through generic pointers every access would be a
// FLAT instruction, counted in vmcnt together with the row loads.)
typedef __attribute__((address_space(3))) float *LdsF;
typedef __attribute__((address_space(3))) unsigned *LdsU;
typedef __attribute__((address_space(3))) int *LdsI;

struct Capture {
  LdsF pool;           // the pool
  LdsU rowtab;         // this group's row table
  LdsI fail;           // set when a row of this frame did not fit
  int base4;           // this WAVE's private region of the pool: first unit (of 4 floats) ...
  int cap4;            // ... and size in units
  bool on;             // wave-uniform: capture this frame at all
};

constexpr unsigned kRowEmpty = (511u << 9);  // a row inside the bounding box without a valid pixel (cnt = 0);
                                             // 0 is kept for "no such row" (outside the bounding box)
__device__ __forceinline__ unsigned row_pack(int off4, int first, int cnt) {
  return ((unsigned)off4 << 18) | ((unsigned)first << 9) | (unsigned)cnt;
}

// ---- phase 1: extents of all valid back-projected pixels of rows [rbeg, rend) ----------------
// NW waves cooperate (row = rbeg + wave + NW*i); the result is wave-uniform in every thread.
// `wave` is the (scalar) index of this wave among the NW cooperating waves, `sync` their barrier.
// AUG: the extents are those of the affinely mapped cloud p' = A p + b (xf = 12 doubles), which needs
// every valid pixel transformed (the monotone shortcut does not survive a rotation).
template <int NW, bool AUG, bool CAP, typename SYNC>
__device__ __forceinline__ void phase1_extents(const Frame &f, const CamK &k, int rbeg, int rend, float *red,
                                               float (&fin)[kExt], const int wave, SYNC sync, const Capture &cap,
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
  int cap_used = 0;  // units of this wave's pool region taken by the frame so far (scalar)

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
    constexpr int kU = P <= 4 ? 4 : 3;           // rows per register buffer (bytes in flight vs VGPRs; deeper
                                                 // buffers in the split kernel did not shorten its latency)
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
      // ---- pass 1: which lanes of each row hold a pixel with |d| >= eps; ONE pool allocation for the kU rows.
      // (Columns past the row end are not masked here: such a lane holds pixels of the next row, or a reload
      // of this row's last pixels — either can only widen the captured window up to the row end, or make an
      // empty row look occupied; pass 2 applies the exact per-pixel rule.) ----
      unsigned long long vm[kU];       // 0: nothing to do for the row
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int row = row0 + kWaves * u;
        float amax = __builtin_fabsf(v[u].d[0]);
#pragma unroll
        for (int j = 1; j < P; ++j) amax = vmax(amax, __builtin_fabsf(v[u].d[j]));  // NaN operands are dropped
        // most row segments hold no valid pixel at all: they are skipped wave-wide (rows past the band are
        // clamped duplicates of its last row)
        vm[u] = row < rend ? __ballot(amax >= k.eps) : 0ull;                          // pre/tsdf_numba.py:87
      }
      // ---- pass 2: the extents ----
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int row = row0 + kWaves * u;
        if (vm[u]) {
          bool ok[P];
#pragma unroll
          for (int j = 0; j < P; ++j) ok[j] = (__builtin_fabsf(v[u].d[j]) >= k.eps) & mine[j];  // :87 (NaN -> invalid)
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
      // ---- pass 3: copy the lane windows lf..ll of every occupied row into the pool (a handful of LDS stores
      // under one exec mask) and post the rows' entries ----
      if (CAP && cap.on) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int row = row0 + kWaves * u;
          if (row >= rend) continue;  // scalar
          unsigned ent = kRowEmpty;
          if (vm[u]) {
            const int lf = __builtin_ctzll(vm[u]), ll = 63 - __builtin_clzll(vm[u]);  // scalar
            const int need = (P * (ll - lf + 1) + 3) >> 2;
            if (cap_used + need <= cap.cap4) {
              const int off4 = cap.base4 + cap_used;
              cap_used += need;
              const int first = P * lf;
              const int endc = P * (ll + 1) < f.bw ? P * (ll + 1) : f.bw;  // a window crossing the row end is cut
              const LdsF dst = cap.pool + (4 * off4 + P * (lane - lf));
              if ((unsigned)(lane - lf) <= (unsigned)(ll - lf)) {
#pragma unroll
                for (int j = 0; j < P; ++j) dst[j] = v[u].d[j];
              }
              ent = row_pack(off4, first, endc - first);
            } else if (lane == 0) {
              *cap.fail = 1;
            }
          }
          if (lane == 0) cap.rowtab[row] = ent;
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
  // Pixel coordinates are kept relative to a frame of reference (px0, py0): the bounding box when the
  // voxel pass gathers from the LDS pool, the rectangle holding every valid pixel when it gathers from
  // global memory (everything outside that rectangle is rejected by pre/tsdf_numba.py:36 or :40).
  int px0, py0;  // image coordinates of that frame's first pixel
  int dx, dy;    // its extent - 1 (inclusive upper bounds of relative coordinates)
  int stride;    // rows of the gather source: elements per row (the crop's, or the staged rectangle's) ...
  int stride4;   // ... and bytes per row
  int base;      // global gather: index of the frame of reference's first pixel in the crop (the source pointer
                 // handed to the voxel pass already points there)
  double dxc, dyc;  // px0 - cx, py0 - cy: pix - c = relative coordinate + this
};

// Column codes.  The x table (and the on-the-fly projection) hand the voxel pass a column as its BYTE offset in
// the row, 4 * column, or kColBad when the voxel projects outside the frame of reference; a row is its index or
// -1.  With that, row * stride4 + code is the gather's byte offset, and it is negative exactly when the pixel
// does not exist (row -1: code - stride4 < 0; kColBad swamps any row) — one compare and one max instead of an
// or, a compare, a select and a shift per voxel.
constexpr int kColBad = -(1 << 30);
__device__ __forceinline__ int col_code(int rel) { return rel >= 0 ? rel << 2 : kColBad; }

__device__ __forceinline__ void zero_volume(float *__restrict__ out, int R, int tid, int T, int part, int parts) {
  const int n4 = 3 * R * R * R / 4;
  const int per = (n4 + parts - 1) / parts;
  const int beg = part * per, end = beg + per < n4 ? beg + per : n4;
  f4 *o4 = reinterpret_cast<f4 *>(out);
  const f4 z = {0.f, 0.f, 0.f, 0.f};
  for (int i = beg + tid; i < end; i += T) o4[i] = z;
}

// Projection of a voxel coordinate onto a pixel coordinate, pre/tsdf_numba.py:30-32:
//   pix = int(v * q + c)  with q = -F / v_z   (multiply, round, add, round, truncate)
// returned relative to p0, or -1 when outside [p0, p0 + dmax] (then :36 or :40 rejects).
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

// The gather source is either the LDS pool of captured row spans or the frame in global memory.  It is
// passed with its address space in the type: a generic pointer would make every gather a FLAT load,
// which is counted in vmcnt together with the volume stores, so each loop iteration would wait for the
// previous iteration's stores to be acknowledged by memory.  As ds_read the gather only touches lgkmcnt
// and the stores stay in flight.
typedef const __attribute__((address_space(3))) float *LdsSrc;
typedef const __attribute__((address_space(1))) float *GlobalSrc;
typedef __attribute__((address_space(1))) float *GlobalOut;  // the output volume
typedef __attribute__((address_space(1))) int *GlobalPix;    // the diagnostic pixel map

// A staged rectangle in LDS (TSDF_FILL 0): addressed like the crop in global memory (row * stride + column,
// coordinates relative to the rectangle), only in LDS.
struct LdsRect {
  LdsSrc p;
};

// Depth of pixel (column code xc, row): pre/tsdf_numba.py:36-39.  `ent` is the row's pool entry (span capture);
// inb = the pixel exists in the source.  The load is always in bounds.
__device__ __forceinline__ float gather_px(const LdsSrc pool, const VoxK &, int xc, int, unsigned ent, bool &inb) {
  const int rel = (xc >> 2) - (int)((ent >> 9) & 511u);
  inb = (unsigned)rel < (ent & 511u);
  const int idx = inb ? (int)((ent >> 18) << 2) + rel : 0;
  return pool[idx];
}
__device__ __forceinline__ float gather_px(const GlobalSrc src, const VoxK &k, int xc, int ry, unsigned, bool &inb) {
  int off = __mul24(ry, k.stride4) + xc;   // bytes; negative iff the pixel is outside (see kColBad)
  inb = off >= 0;
  off = off > 0 ? off : 0;                  // the load is always in bounds
  return *(GlobalSrc)((const __attribute__((address_space(1))) char *)src + off);
}
__device__ __forceinline__ float gather_px(const LdsRect src, const VoxK &k, int xc, int ry, unsigned, bool &inb) {
  int off = __mul24(ry, k.stride4) + xc;
  inb = off >= 0;
  off = off > 0 ? off : 0;
  return *(LdsSrc)((const __attribute__((address_space(3))) char *)src.p + off);
}

// The same for a pixel given by its coordinates relative to the frame of reference (possibly outside it): the
// on-the-fly projection of the augmented pass, which has no tables whose entries could carry the -1.
struct Tabs;
template <class SrcP>
__device__ __forceinline__ float gather_rel(const SrcP src, const VoxK &k, const Tabs &tb, int relx, int rely, bool &inb);

// Per-voxel value, pre/tsdf_numba.py:36-68, for the 4 voxels of one lane.  Coordinates are pre-scaled
// by it = 1/trunc_dis:  tx = v_x*it - (pix_x-cx)*(pd*kq),  ty likewise,  tz = v_z*it + pd*it (w_z = -pd).
//   ex[j], ry[j], ent[j]  relative pixel of voxel j (ex: column CODE, see kColBad; ry -1: rejected) and its row's entry
//   vxs[j], vys, vzs[j]   pre-scaled voxel centre;   negthr[j] = f32_round_up(-v_z)
// Returns the mask of voxels that passed :36 and :40 (bit j).
template <class SrcP>
__device__ __forceinline__ unsigned voxel_values4(const int (&ex)[4], const int (&ry)[4], const unsigned (&ent)[4],
                                                  const double (&vxs)[4], const double vys, const double (&vzs)[4],
                                                  const float (&negthr)[4], const VoxK &k,
                                                  const SrcP src, f4 &o0, f4 &o1, f4 &o2) {
  float pd[4];
  bool inb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) pd[j] = gather_px(src, k, ex[j], ry[j], ent[j], inb[j]);   // :36-39
  bool ok[4], neg[4];
  double pd64[4], tz[4];
  bool any_near = false;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ok[j] = inb[j] & (__builtin_fabsf(pd[j]) >= k.eps);                     // :40 (NaN -> rejected)
    pd64[j] = (double)pd[j];
    tz[j] = __builtin_fma(pd64[j], k.it, vzs[j]);                           // :46,:49
    neg[j] = pd[j] < negthr[j];                                             // w_z > v_z  :65
    any_near |= ok[j] & (__builtin_fabs(tz[j]) <= 1.0);
  }
  // A voxel that is rejected is 0 in all channels (:33-41), one beyond the truncation distance is (+-1,+-1,+-1)
  // (:54-57), signed by :65-68: that is sv[j], and it is the whole answer unless the wave holds a near voxel.
  float sv[4];
  unsigned okm = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sv[j] = ok[j] ? (neg[j] ? -1.0f : 1.0f) : 0.0f;
    okm |= (unsigned)ok[j] << j;
  }
  o0 = f4{sv[0], sv[1], sv[2], sv[3]};
  o1 = o0;
  o2 = o0;
  if (__any(any_near)) {
    // otherwise every voxel of this wave is rejected or beyond the truncation distance along z
    // alone: dist >= |tz| > 1 -> (1,1,1), and the x/y terms are not needed
    float *p0 = reinterpret_cast<float *>(&o0), *p1 = reinterpret_cast<float *>(&o1),
          *p2 = reinterpret_cast<float *>(&o2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double a = pd64[j] * k.kq;                                      // pd/F/trunc         :43
      const double dxi = __builtin_fma((double)ex[j], 0.25, k.dxc);         // pix_x - cx         :44 (exact)
      const double dyi = (double)ry[j] + k.dyc;                             // pix_y - cy         :45
      const double tx = __builtin_fma(-dxi, a, vxs[j]);                     // (v_x - w_x)/trunc  :47
      const double ty = __builtin_fma(dyi, a, vys);                         // (v_y - w_y)/trunc  :48, w_y = -dyi*q
      const double s = __builtin_fma(tz[j], tz[j], __builtin_fma(ty, ty, tx * tx));  // dist^2 :51-52
      const bool nearv = s <= 1.0;                                          // :54 (sqrt monotone, sqrt(1)=1)
      // |t| clamped to 1 (:58-60), float32 (:70-72); times sv = +-1 or 0: exact, and the sign lands on a zero too
      const float m0 = vminabs((float)tx, 1.0f);
      const float m1 = vminabs((float)ty, 1.0f);
      const float m2 = vminabs((float)tz[j], 1.0f);
      p0[j] = nearv ? __fmul_rn(m0, sv[j]) : sv[j];
      p1[j] = nearv ? __fmul_rn(m1, sv[j]) : sv[j];
      p2[j] = nearv ? __fmul_rn(m2, sv[j]) : sv[j];
    }
  }
  return okm;
}

#ifndef TSDF_NT_STORE
#define TSDF_NT_STORE 1
#endif
#ifndef TSDF_TAIL_HELP
#define TSDF_TAIL_HELP 1
#endif
// One 16-byte store of the output volume.  It is written once and never re-read here, so it goes out
// non-temporal and does not evict the depth rows other workgroups are streaming through L2 / Infinity
// Cache (measured in round 1: 180 -> 155 us per 1024 frames; re-checked in round 3: without nt +16 % at 32^3).
// Round 3: the store also carries DEVICE scope (sc1): it is written through towards memory instead of waiting in
// this XCD's L2 for a write-back — same-buffer paired A/B, 24 blocks each: 1024 full frames -2.2 % +- 0.2, 1024 crops
// -1.5 %, 64^3 augmented -1.3 %, 64^3 crops -1.6 % (gpurun_out/ab_store_policy2.log); "sc0 sc1 nt" (system scope) is
// within noise of it, workgroup scope ("sc0 nt") loses the gain again (+2.7 %), dropping nt costs 16 %.  (Also tried on the
// read side: nt on the row loads is +12.5 % on full frames — the staging copy's re-read then misses L2 / Infinity Cache,
// which shows that it normally hits —, nt on the staging copy itself changes nothing; raising the wave priority of either
// phase with s_setprio costs 3-4 %.)  There is no builtin for the scope bits of a plain store, hence the
// inline assembly; the s_nop covers the "VALU overwrites the data registers of a wide store" hazard the compiler
// can no longer see.  -DTSDF_STORE_ASM='"..."' selects other bits; -DTSDF_STORE_BUILTIN the compiler's nt store.
#ifndef TSDF_STORE_ASM
#define TSDF_STORE_ASM "sc1 nt"
#endif
__device__ __forceinline__ void store_vol4(GlobalOut p, f4 v) {
#if defined(TSDF_STORE_BUILTIN)
  __builtin_nontemporal_store(v, (__attribute__((address_space(1))) f4 *)p);
#elif TSDF_NT_STORE
  asm volatile("global_store_dwordx4 %0, %1, off " TSDF_STORE_ASM "\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#else
  *(__attribute__((address_space(1))) f4 *)p = v;
#endif
}

// Diagnostic pixel map (tsdf_debug_pixmap_hip), one voxel: see include/tsdf.h.  `frame_dc/dr` shift the
// tables' frame of reference back into the bounding box.
struct PixMapK {
  GlobalPix out;  // int32[R][R][R] of this frame, [z][y][x]; null: no map
  int bw, dc, dr;
};
__device__ __forceinline__ int pixmap_value(const PixMapK &pm, int ex, int ry, bool row_in_bbox, bool ok) {
  if (ex < 0 || !row_in_bbox) return -1;                      // :36-37
  const int idx = (ry + pm.dr) * pm.bw + ex + pm.dc;          // :38
  return ok ? idx : -2 - idx;                                 // :40-41
}

// LDS-resident per-frame tables, one set per group.  The pixel a voxel projects to factorises: pix_x
// depends on (x, z) only and pix_y on (y, z) only, so for R <= kTabR both are tabulated once per frame
// (R*R entries each, one pair per thread) instead of 3 float64 operations + a range test per voxel.
//   pxtab[.]  column code of the relative pix_x (4 * column, or kColBad)
//   pytab[.]  LDS gather: the pool entry of row pix_y (0: outside the bounding box)
//             global gather: relative pix_y, or -1
//   pyrow[.]  LDS gather: relative pix_y (0 when outside)              (uint8; rows < kMaxRows)
struct ZEntry {
  double q;      // -F / v_z                     :30
  double vzs;    // v_z / trunc_dis
  float negthr;  // f32_round_up(-v_z)
  float pad;
};

typedef __attribute__((address_space(3))) const ZEntry *LdsZ;
typedef __attribute__((address_space(3))) const int *LdsCI;
typedef __attribute__((address_space(3))) const unsigned *LdsCU;
typedef __attribute__((address_space(3))) const unsigned char *LdsCU8;
typedef __attribute__((address_space(3))) const double *LdsCD;

typedef int i4v __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const i4v *LdsI4;
typedef __attribute__((address_space(3))) const u4v *LdsU4;

__device__ __forceinline__ ZEntry load_z(LdsZ p) {
  ZEntry z;
  z.q = p->q;
  z.vzs = p->vzs;
  z.negthr = p->negthr;
  z.pad = 0.f;
  return z;
}

struct Tabs {
  LdsZ ztab;
  LdsCI pxtab;
  LdsCU pytab;
  LdsCU8 pyrow;
  LdsCU rowtab;
  LdsCD atab;
};

// table index of (fast, slow) coordinates: the 4 entries a lane needs are contiguous
template <int LAYOUT>
__device__ __forceinline__ int tab_index(int x_or_y, int z, int R) {
  return LAYOUT == 0 ? z * R + x_or_y : x_or_y * R + z;
}

template <class SrcP>
struct IsLds {
  static constexpr bool value = std::is_same<SrcP, LdsSrc>::value;
};

// row entry for an on-the-fly projected row (no tables): LDS gather looks the row table up
template <class SrcP>
__device__ __forceinline__ unsigned row_entry(const Tabs &tb, int ry) {
  if constexpr (IsLds<SrcP>::value) {
    return ry >= 0 ? tb.rowtab[ry] : 0u;
  } else {
    return (unsigned)ry;
  }
}

template <>
__device__ __forceinline__ float gather_rel<GlobalSrc>(const GlobalSrc src, const VoxK &k, const Tabs &, int relx, int rely,
                                                       bool &inb) {
  inb = ((unsigned)relx <= (unsigned)k.dx) & ((unsigned)rely <= (unsigned)k.dy);   // :36
  const int idx = inb ? __mul24(rely, k.stride) + relx : 0;
  return src[idx];                                                                  // :38-39
}
template <>
__device__ __forceinline__ float gather_rel<LdsRect>(const LdsRect src, const VoxK &k, const Tabs &, int relx, int rely,
                                                     bool &inb) {
  inb = ((unsigned)relx <= (unsigned)k.dx) & ((unsigned)rely <= (unsigned)k.dy);
  const int idx = inb ? __mul24(rely, k.stride) + relx : 0;
  return src.p[idx];
}
template <>
__device__ __forceinline__ float gather_rel<LdsSrc>(const LdsSrc src, const VoxK &k, const Tabs &tb, int relx, int rely,
                                                    bool &inb) {
  const bool iny = (unsigned)rely <= (unsigned)k.dy;
  const unsigned ent = iny ? tb.rowtab[iny ? rely : 0] : 0u;
  return gather_px(src, k, relx << 2, rely, ent, inb);   // a column outside the row's window (or the box) fails its test
}

// Dynamic units (tile_ctr != null: the one-group-per-CU instantiations, R >= 48).
// With the static split a wave keeps one slab of rows (64/R4 of them: one wave tile per slice) for the whole volume and
// walks the slices.  Two things are wrong with that at 64^3 (round 4):
//  * the ORDER in which the 3 MiB volume is written.  tools/probes/vol_store_probe.hip writes [3][64][64][64] volumes,
//    store-only, in a dozen orders: the static split's order (its waves in step or out of step) is the slowest of all —
//    5.0-5.1 TB/s where the best order reaches 6.1 on the same box (profiles/r04/vol_store_probe*.log; on the pool's fast
//    boxes the spread is 6.5 vs 6.9).  Best: the 16 waves of a CU on consecutive 2-slice pieces of ONE slab, every wave
//    writing its 1 KiB piece in the three channel planes, then the next slab.  That is what units of (slab) x (2 slices)
//    drawn from a counter produce.  Paired A/Bs of the real kernels, same buffer (profiles/r04/ab_plain_dyn.log,
//    ab_aug_chunks.log): plain 64^3 full frames -4.0 %, crops -4.7 %; augmented -3.1 % / -3.2 %.  Units of 1 slice:
//    -3.6 % / -0.3 %; of 4: +0.2 % / -0.7 %; of 8: +1.4 % (augmented).
//  * each SIMD serves its four waves oldest first, so a CU's waves leave a static pass in four steps — after 55, 87, 116
//    and 141 us (profiles/r04/stamps_aug64_per_wave_static.log) — and the youngest wave of each SIMD runs the last
//    quarter of the pass alone.  With units all 16 waves stay busy to the end (..._dyn8.log: 128-131 us each).
// A unit is still "pure" — a wave tile's 64 x 4 voxels are neighbours, so the wave-uniform shortcuts keep their hit
// rate — a wave's stores are still 1 KiB contiguous, and the results are bit-identical (the A/B tool asserts it).  The
// middle slabs, whose tiles most often need the x/y terms, go first.  Per-slab state of the augmented pass is kept
// while a wave stays in its slab (it mostly does: a slab's 32 units are drawn one after the other).
// 32^3 keeps its static two-slices-per-step order: the same probe for 32^3 volumes (vol32_store_probe.hip) finds it
// among the best already.
#ifndef TSDF_DYN_TILES
#define TSDF_DYN_TILES 1
#endif
#ifndef TSDF_DYN_CHUNK
#define TSDF_DYN_CHUNK 2
#endif
constexpr int kDynChunk = TSDF_DYN_CHUNK;

// The unit plan of one pass over slices [sb, se) with n_slab slabs: unit -> (slab, first slice, end slice).
struct DynPlan {
  int n_slab, sb, se;
  int n_chunk, n_unit;      // chunks per slab; units in all
};
__device__ __forceinline__ DynPlan dyn_plan(int n_slab, int sb, int se) {
  DynPlan p;
  p.n_slab = n_slab;
  p.sb = sb;
  p.se = se;
  p.n_chunk = (se - sb + kDynChunk - 1) / kDynChunk;
  p.n_unit = n_slab * p.n_chunk;
  return p;
}
__device__ __forceinline__ int dyn_slab(int rank, int n_slab);
__device__ __forceinline__ void dyn_unit(const DynPlan &p, int unit, int &slab, int &zb, int &ze) {
  const int rank = unit / p.n_chunk, c = unit - rank * p.n_chunk;
  slab = dyn_slab(rank, p.n_slab);
  zb = p.sb + c * kDynChunk;
  ze = zb + kDynChunk < p.se ? zb + kDynChunk : p.se;
}

// unit number -> slab: middle-out (n_slab even: h-1, h, h-2, h+1, ...)
__device__ __forceinline__ int dyn_slab(int rank, int n_slab) {
  const int h = n_slab >> 1, k = rank >> 1;
  const int s = (rank & 1) ? h + k : h - 1 - k;
  return s < 0 ? 0 : (s >= n_slab ? n_slab - 1 : s);
}

// The voxel pass over slow-axis slices [sb, se).  T threads take part (tid in [0, T)): T = kGW when a group
// works alone, 2*kGW when the CU's other group helps (its threads come in as kGW + gtid), kWG in the split
// kernel.
template <int LAYOUT, int T, bool DBG, class SrcP>
__device__ __forceinline__ void phase2(const Grid &g, const CamK &cam, const VoxK &vk, int R, const Tabs &tb,
                                       const bool use_tab, const SrcP src, const GlobalOut out, const int tid,
                                       const int sb, const int se, const PixMapK &pm, int *tile_ctr = nullptr) {
  if (se <= sb) return;
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
  // dynamic units (see above): (slab of 64/R4 rows) x (kDynChunk slices) drawn from a counter in LDS
  const bool dyn = TSDF_DYN_TILES && tile_ctr != nullptr && R4 <= 64 && (64 % R4) == 0 && (R % (64 / R4)) == 0;  // uniform
  const int n_slab = dyn ? R / (64 / R4) : 0;
  const DynPlan plan = dyn_plan(n_slab, sb, se);
  auto draw = [&]() -> int {
    int t = 0;
    if ((tid & 63) == 0) t = __hip_atomic_fetch_add(tile_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __builtin_amdgcn_readfirstlane(t);
  };
  int unit = dyn ? draw() : 0;
  for (int gi = g0;; gi += gstep) {
    int zb = sb + s0, ze = se, zs = sstep;   // this pass's slices (uniform)
    if (dyn) {
      if (unit >= plan.n_unit) break;
      int slab;
      dyn_unit(plan, unit, slab, zb, ze);
      gi = slab * 64 + (tid & 63);
      zs = 1;
      unit = draw();   // (the next unit's number travels while this one is computed)
    } else if (gi >= G) {
      break;
    }
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
      for (int z = zb; z < ze; z += zs) {
        const ZEntry zen = load_z(tb.ztab + z);
        int ex[4], ry[4];
        unsigned ent[4];
        if (use_tab) {
          const i4v e = *(LdsI4)(tb.pxtab + z * R + f4i);
          ex[0] = e.x; ex[1] = e.y; ex[2] = e.z; ex[3] = e.w;
          ent[0] = tb.pytab[z * R + y];
          ry[0] = IsLds<SrcP>::value ? (int)tb.pyrow[z * R + y] : (int)ent[0];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) ex[j] = col_code(project_rel(vx[j], zen.q, cam.cx, vk.px0, vk.dx));  // :31
          ry[0] = project_rel(-vy, zen.q, cam.cy, vk.py0, vk.dy);                                 // :32
          ent[0] = row_entry<SrcP>(tb, ry[0]);
        }
        ry[1] = ry[2] = ry[3] = ry[0];
        ent[1] = ent[2] = ent[3] = ent[0];
        const double vzs[4] = {zen.vzs, zen.vzs, zen.vzs, zen.vzs};
        const float negthr[4] = {zen.negthr, zen.negthr, zen.negthr, zen.negthr};
        f4 o0, o1, o2;
        const unsigned okm = voxel_values4(ex, ry, ent, vxs, vys, vzs, negthr, vk, src, o0, o1, o2);
        const int64_t e = ((int64_t)z * R + y) * R + f4i;                   // o[c][z][y][x] :70-72
        store_vol4(out + e, o0);
        store_vol4(out + R3 + e, o1);
        store_vol4(out + 2 * R3 + e, o2);
        if constexpr (DBG) {
          const bool rin = IsLds<SrcP>::value ? ent[0] != 0u : ry[0] >= 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) pm.out[e + j] = pixmap_value(pm, ex[j] >> 2, ry[0], rin, (okm >> j) & 1u);
        }
      }
    } else {
      // lanes run along z: q, v_z and pix_y fixed per lane, loop over x
      double q[4], vzs[4];
      float negthr[4];
      int ry[4];
      unsigned ent[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const ZEntry zen = load_z(tb.ztab + f4i + j);
        q[j] = zen.q;
        vzs[j] = zen.vzs;
        negthr[j] = zen.negthr;
      }
      if (use_tab) {
        const u4v e = *(LdsU4)(tb.pytab + y * R + f4i);
        ent[0] = e.x; ent[1] = e.y; ent[2] = e.z; ent[3] = e.w;
        if constexpr (IsLds<SrcP>::value) {
          const unsigned rr = *(LdsCU)(tb.pyrow + y * R + f4i);
          ry[0] = rr & 255u; ry[1] = (rr >> 8) & 255u; ry[2] = (rr >> 16) & 255u; ry[3] = rr >> 24;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) ry[j] = (int)ent[j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ry[j] = project_rel(-vy, q[j], cam.cy, vk.py0, vk.dy);
          ent[j] = row_entry<SrcP>(tb, ry[j]);
        }
      }
      for (int x = zb; x < ze; x += zs) {
        const double vx = ox + (double)x * vl;
        const double vx1 = vx * vk.it;
        const double vxs[4] = {vx1, vx1, vx1, vx1};
        int ex[4];
        if (use_tab) {
          const i4v e = *(LdsI4)(tb.pxtab + x * R + f4i);
          ex[0] = e.x; ex[1] = e.y; ex[2] = e.z; ex[3] = e.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) ex[j] = col_code(project_rel(vx, q[j], cam.cx, vk.px0, vk.dx));
        }
        f4 o0, o1, o2;
        const unsigned okm = voxel_values4(ex, ry, ent, vxs, vys, vzs, negthr, vk, src, o0, o1, o2);
        const int64_t e = ((int64_t)x * R + y) * R + f4i;                   // o[c][x][y][z] tsdf_for.py:118-120
        store_vol4(out + e, o0);
        store_vol4(out + R3 + e, o1);
        store_vol4(out + 2 * R3 + e, o2);
        if constexpr (DBG) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool rin = IsLds<SrcP>::value ? ent[j] != 0u : ry[j] >= 0;
            pm.out[((int64_t)(f4i + j) * R + y) * R + x] = pixmap_value(pm, ex[j] >> 2, ry[j], rin, (okm >> j) & 1u);
          }
        }
      }
    }
  }
}

// Phase 2 of the augmented form (oracle/tsdf_oracle.c::tsdf_oracle_voxels_aug): the voxel centre v'
// lives in the augmented frame, v = T^-1(v') is projected and the pixel gathered as in the plain pass; the
// distances are those between v' and T(w), w the pixel's surface point.  The projection no longer factorises,
// so there are no pixel tables and q = -F / v_z is one true division per voxel; what does factorise is the
// inverse map: its three products per row, A_i0*v'_x, A_i1*v'_y, A_i2*v'_z, depend on one grid index each and
// are tabulated per frame (atab), leaving (a + b) + (c + d) — the oracle's exact rounding order.
// The distances never form w or T(w) (round 2 did: ~21 float64 operations per voxel before the first test).
// For an affine T,  v'_i - w'_i = (v'_i - b_i) + pd * c_i  with  c_i = fma(g_i0, dxi, fma(g_i1, dyi, A_i2)),
// g_i0 = -(A_i0 / F), g_i1 = A_i1 / F per frame, dxi = pix_x - cx, dyi = pix_y - cy: three fma for the z
// component that decides everything a far or rejected voxel needs (|u_z| > trunc_dis -> (+-1,+-1,+-1), sign of
// u_z), and the x / y components only in wave tiles that hold a near voxel.  The contract (include/tsdf.h,
// restated operation by operation in the oracle) is written in exactly this form.
// atab layout: [axis][index][4] = { fl(inv[4*row + axis] * (ori_axis + index*voxel_len)) for row 0..2 — the z
// entries + b_row —, (ori_axis + index*voxel_len) - fwd_b[axis] }.
// The table of the axis the LANES index (x for LAYOUT 0, z for LAYOUT 1) is stored lane-major: a lane reads the entries
// of its four voxels 4i..4i+3 — 8 pieces of 16 bytes — and the 16 lanes of a grid row read them together, so piece
// (j, half) of all lanes is one contiguous run: double offset of entry `idx`, value `row` (0..3).  With the plain
// [index][4] layout those reads were 128 bytes apart from lane to lane: ds_read_b128 banks 4-way, and since the voxel pass
// works in (slab x 2-slice) units a wave re-reads them every other unit — SQ_LDS_BANK_CONFLICT went from 20 M to 70 M
// cycles per launch between rounds 3 and 4 although the LDS instruction count fell.
__device__ __forceinline__ int lane_slot(int idx, int row, int R4) {
  return ((2 * (idx & 3) + (row >> 1)) * R4 + (idx >> 2)) * 2 + (row & 1);
}

__device__ __forceinline__ double uniform64(double v) {  // a wave-uniform float64 -> scalar registers
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

template <int LAYOUT, int T, class SrcP>
__device__ __forceinline__ void phase2_aug(const Grid &g, const CamK &cam, const VoxK &vk, int R,
                                           const double *xf, const Tabs &tb, const SrcP src,
                                           const GlobalOut out, const int tid, const int sb, const int se,
                                           int *tile_ctr = nullptr, int stamp_iter = 0) {
  (void)stamp_iter;
  TSDF_WSTAMP(stamp_iter, 0);
  if (se <= sb) return;  // (uniform) nothing to do; the end-slice lookups below assume one slice at least
  const double *fwd = xf;
  const LdsCD tabx = tb.atab, taby = tb.atab + 4 * R, tabz = tb.atab + 8 * R;
  const int R4 = R / 4;
  const int G = R * R4;
  const int64_t R3 = (int64_t)R * R * R;
  // per-frame constants of the distance terms, kept in scalar registers
  double g0[3], g1[3], a2[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    g0[i] = uniform64(-(fwd[4 * i] * cam.inv_focal));
    g1[i] = uniform64(fwd[4 * i + 1] * cam.inv_focal);
    a2[i] = fwd[4 * i + 2];
  }
  // |u_z| beyond this is beyond the truncation distance whatever x and y are: t_z = u_z * it with it = fl(1/trunc)
  // is then > 1 + 2^-31 after both roundings, so dist^2 > 1 (the gate may only err towards "look closer")
  const double tgate = uniform64((double)g.trunc * (1.0 + 0x1p-30));
  int g0i, gstep, s0, sstep;
  if (G <= T && (T % G) == 0) {
    g0i = tid % G;
    gstep = G;
    s0 = tid / G;
    sstep = T / G;
  } else {
    g0i = tid;
    gstep = T;
    s0 = 0;
    sstep = 1;
  }
  // (A wave's 64 groups of 4 voxels are one 64 x 4 tile of a slice at 64^3.  Narrower 32 x 8 tiles — fewer slices hold a
  // near voxel for them when the map rotates, 15 % fewer VALU instructions — measured +-0.2 % in a same-buffer paired A/B:
  // after this round's diet the pass is no longer bound by instruction issue.  Not kept.)
  const bool dyn = TSDF_DYN_TILES && tile_ctr != nullptr && R4 <= 64 && (64 % R4) == 0 && (R % (64 / R4)) == 0;  // uniform
  const int n_slab = dyn ? R / (64 / R4) : 0;                      // slabs of 64/R4 rows: one wave tile per slice
  const DynPlan plan = dyn_plan(n_slab, sb, se);
  auto draw = [&]() -> int {
    int t = 0;
    if ((tid & 63) == 0) t = __hip_atomic_fetch_add(tile_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __builtin_amdgcn_readfirstlane(t);
  };
  int unit = dyn ? draw() : 0;
  // Per-slab state, recomputed only when a wave's unit lies in another slab than its previous one (units of one slab are
  // drawn one after the other, so with small units a wave mostly stays in its slab).
  int gi_have = -1;
  double ty0 = 0, ty1 = 0, ty2 = 0, vby = 0, pre[4][3], vbf[4];
  bool mild = false;
  for (int gi = g0i;; gi += gstep) {
    int zb = sb + s0, ze = se, zs = sstep;   // this pass's slices (uniform)
    if (dyn) {
      if (unit >= plan.n_unit) break;
      int slab;
      dyn_unit(plan, unit, slab, zb, ze);
      gi = slab * 64 + (tid & 63);
      zs = 1;
      unit = draw();   // (the next unit's number travels while this one is computed)
    } else if (gi >= G) {
      break;
    }
    const int f4i = (gi % R4) * 4;
    const int y = gi / R4;
    if (gi != gi_have) {   // (uniform: every lane of a wave changes slab together)
      gi_have = gi;
      ty0 = taby[4 * y];
      ty1 = taby[4 * y + 1];
      ty2 = taby[4 * y + 2];
      vby = taby[4 * y + 3];   // v'_y - b_y
      // The inverse map is (A_i0 x' + A_i1 y') + (A_i2 z' + b_i), every product and sum rounded separately (the
      // oracle's affine3).  Both brackets depend on grid indices only: the z table holds (A_i2 z' + b_i), and the
      // bracket that does not change from slice to slice stays in registers — LAYOUT 0 (x, y fixed per lane) keeps
      // (A_i0 x' + A_i1 y') for its 4 voxels, LAYOUT 1 (y, z fixed) keeps the z bracket — so a voxel costs ONE add
      // per row of the map.  The same goes for v' - b of the distance terms: the fixed axes' live in registers
      // (vbf), the slice's comes from its table entry.
      // (The lanes' own axis is read from the lane-major image of its table — see lane_slot —: 16-byte pieces, the 16
      // lanes of a row side by side, so the eight ds_read_b128 of a slab change are conflict-free.)
      const LdsCD tl = LAYOUT == 0 ? tabx : tabz;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const LdsCD lo = tl + lane_slot(f4i + j, 0, R4), hi = tl + lane_slot(f4i + j, 2, R4);
        if constexpr (LAYOUT == 0) {
          pre[j][0] = lo[0] + ty0;
          pre[j][1] = lo[1] + ty1;
          pre[j][2] = hi[0] + ty2;
        } else {
          pre[j][0] = lo[0];
          pre[j][1] = lo[1];
          pre[j][2] = hi[0];
        }
        vbf[j] = hi[1];                      // v' - b of the lanes' own axis (x for LAYOUT 0, z for LAYOUT 1)
      }
      // Which division (neg_focal_over): v_z of a lane's voxel is fl(pre + slice term), monotone in the slice index
      // (every rounding is), so the pass's two end slices bound it; same sign and mid-range at both ends -> mid-range
      // in every slice of every unit of the slab.
      mild = mid_range(cam.focal, 0x1p-100, 0x1p100);
      const LdsCD tlo = (LAYOUT == 0 ? tabz : tabx) + 4 * sb, thi = (LAYOUT == 0 ? tabz : tabx) + 4 * (se - 1);
      const double s_lo = LAYOUT == 0 ? tlo[2] : tlo[2] + ty2, s_hi = LAYOUT == 0 ? thi[2] : thi[2] + ty2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double z_a = pre[j][2] + s_lo, z_b = pre[j][2] + s_hi;
        mild = mild && mid_range(z_a, 0x1p-600, 0x1p600) && mid_range(z_b, 0x1p-600, 0x1p600) && ((z_a > 0.0) == (z_b > 0.0));
      }
      mild = __all(mild);
    }
    auto slices = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    for (int sl = zb; sl < ze; sl += zs) {
      // ---- project the 4 voxels and gather their depths ----
      int ex[4], ry[4];
      float pd[4];
      bool ok[4];
      double sl0, sl1, sl2, vbs;  // the slice's own terms (wave-uniform)
      if constexpr (LAYOUT == 0) {
        const LdsCD tzp = tabz + 4 * sl;
        sl0 = tzp[0];
        sl1 = tzp[1];
        sl2 = tzp[2];
        vbs = tzp[3];                      // v'_z - b_z
      } else {
        const LdsCD tx = tabx + 4 * sl;
        sl0 = tx[0] + ty0;
        sl1 = tx[1] + ty1;
        sl2 = tx[2] + ty2;
        vbs = tx[3];                       // v'_x - b_x
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double vx = pre[j][0] + sl0;   // v = T^-1(v') = (A_i0 x' + A_i1 y') + (A_i2 z' + b_i)
        const double vy = pre[j][1] + sl1;
        const double vz = pre[j][2] + sl2;
        const double q = neg_focal_over<FAST>(vz, cam);                          // :30  -F / v_z
        ex[j] = trunc_i32(mul_then_add(vx, q, cam.cx)) - vk.px0;                 // :31, relative; may lie outside
        ry[j] = trunc_i32(mul_then_add(-vy, q, cam.cy)) - vk.py0;                // :32
        bool inb;
        pd[j] = gather_rel<SrcP>(src, vk, tb, ex[j], ry[j], inb);                // :36-39
        ok[j] = inb & (__builtin_fabsf(pd[j]) >= vk.eps);                        // :40
      }
      // ---- z component first: a wave whose voxels are all beyond the truncation distance along z' alone
      // needs nothing else (dist >= |t_z| > 1 -> (1,1,1)); one whose voxels are all rejected needs nothing ----
      f4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0, o2 = o0;
      if (__any(ok[0] | ok[1] | ok[2] | ok[3])) {
        double dxi[4], dyi[4], uz[4];
        float sv[4];  // 0 for a rejected voxel, else the sign of :65-68 as +-1: the whole answer unless near
        bool any_near = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dxi[j] = (double)ex[j] + vk.dxc;                                         // pix_x - cx  :44 (exact)
          dyi[j] = (double)ry[j] + vk.dyc;                                         // pix_y - cy  :45
          const double cz = __builtin_fma(g0[2], dxi[j], __builtin_fma(g1[2], dyi[j], a2[2]));
          uz[j] = __builtin_fma((double)pd[j], cz, LAYOUT == 0 ? vbs : vbf[j]);    // v'_z - w'_z  :46,:49
          any_near |= ok[j] & (__builtin_fabs(uz[j]) <= tgate);
          sv[j] = ok[j] ? (uz[j] < 0.0 ? -1.0f : 1.0f) : 0.0f;                     // w'_z > v'_z  :65
        }
        o0 = f4{sv[0], sv[1], sv[2], sv[3]};
        o1 = o0;
        o2 = o0;
        if (__any(any_near)) {
          float *p0 = reinterpret_cast<float *>(&o0), *p1 = reinterpret_cast<float *>(&o1),
                *p2 = reinterpret_cast<float *>(&o2);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const double pd64 = (double)pd[j];
            const double cxx = __builtin_fma(g0[0], dxi[j], __builtin_fma(g1[0], dyi[j], a2[0]));
            const double cyy = __builtin_fma(g0[1], dxi[j], __builtin_fma(g1[1], dyi[j], a2[1]));
            const double ux = __builtin_fma(pd64, cxx, LAYOUT == 0 ? vbf[j] : vbs);  // v'_x - w'_x  :47
            const double uy = __builtin_fma(pd64, cyy, vby);                         // v'_y - w'_y  :48
            const double tx = ux * vk.it, ty = uy * vk.it, tz = uz[j] * vk.it;
            const double s2 = __builtin_fma(tz, tz, __builtin_fma(ty, ty, tx * tx));
            const bool nearv = s2 <= 1.0;                                          // :54
            const float m0 = vminabs((float)tx, 1.0f);
            const float m1 = vminabs((float)ty, 1.0f);
            const float m2 = vminabs((float)tz, 1.0f);
            p0[j] = nearv ? __fmul_rn(m0, sv[j]) : sv[j];                          // exact: sv is +-1 or 0
            p1[j] = nearv ? __fmul_rn(m1, sv[j]) : sv[j];
            p2[j] = nearv ? __fmul_rn(m2, sv[j]) : sv[j];
          }
        }
      }
      const int64_t e = ((int64_t)sl * R + y) * R + f4i;  // o[c][slow][y][fast]
      store_vol4(out + e, o0);
      store_vol4(out + R3 + e, o1);
      store_vol4(out + 2 * R3 + e, o2);
    }
    };
    if (mild) {
      slices(std::true_type{});
    } else {
      slices(std::false_type{});
    }
  }
  TSDF_WSTAMP(stamp_iter, 1);
}


// ---- kernel arguments ---------------------------------------------------------------------------
struct KArgs {
  const float *depth;
  const int64_t *offsets;
  const int32_t *headers;
  int n, R;
  CamK cam;
  float *tsdf, *max_l, *mid_p;
  int32_t *status;
  float *aabb, *grid, *ori;
  int aabb_only;
  const float *grid_in;
  unsigned long long *queue;  // this launch's work-queue word, or null: CU-local queues (see the kernel)
  unsigned int qepoch;        // ... and the launch's number on that word (never 0): see queue_ticket()
  const double *xforms;
  int64_t depth_len;
  const int64_t *index;   // tsdf_voxelize_indexed_hip: batch position -> frame of the resident pack (else null)
  int64_t n_src;          // ... and the number of frames in that pack
  int n_inline;           // tsdf_voxelize_indexed_host_hip: the index travels IN the kernel arguments (n <= kInlineIndex)
  int64_t inline_index[TSDF_INLINE_INDEX_MAX];
  const float *gt;        // labels (optional)
  float *gt_nor, *gt_aug;
  int n_joints, clamp;
  int32_t *pixmap;        // diagnostic pixel map (DBG instantiations only)
  int split, per;         // split kernel: workgroups per frame, slow-axis slices per workgroup
  float *xchg;            // split kernel: this stream's mailboxes for partial extents, or null (see the kernel)
  unsigned int seq;       // ... and the number this launch tags them with
  int polls;              // ... and the bound of the wait for them (kXchgPolls; tests set TSDF_XCHG_POLLS=0 to
                          // force every workgroup onto the fallback)
};

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
  int64_t src;  // where the frame's offsets / header / labels are read: `frame`, or index[frame] (indexed entry)
};

// The frame's header and offsets.  Indexed entry: batch position fr reads pack frame index[fr]; an index outside the
// pack leaves off1 < off0, which frame_from_header turns into TSDF_FRAME_BAD_HEADER (nothing is read through it).
__device__ __forceinline__ void fetch_header(const KArgs &a, const int64_t *__restrict__ in_offsets,
                                             const int32_t *__restrict__ in_headers, int fr, FrameHdr &m) {
  int64_t src = fr;
  bool ok = true;
  if (a.n_inline) {           // a small batch whose index came by value: no memory outside the kernel arguments is read
    src = a.inline_index[fr];
    ok = src >= 0 && src < a.n_src;
    if (!ok) src = 0;
  } else if (a.index) {
    src = a.index[fr];
    ok = src >= 0 && src < a.n_src;
    if (!ok) src = 0;
  }
  const int32_t *h = in_headers + 6 * src;
  m.l = h[2];
  m.t = h[3];
  m.r = h[4];
  m.b = h[5];
  m.off0 = in_offsets[src];
  m.off1 = ok ? in_offsets[src + 1] : m.off0 - 1;
  m.src = src;
}

// Tail help: a group that finds the queue empty does not leave at once.  It raises idle[] and waits; the
// CU's other group, on reaching phase 2 of what is then necessarily its last frame, sees the flag, posts
// the frame's voxel parameters here and both groups split the slow axis of the volume (the pool and the
// tables are in LDS, which the two share).  Nothing has to be handed back: the helper leaves when done.
struct HelpReq {
  Grid g;
  VoxK vk;
  const float *src;  // the frame in global memory (gather source when the frame was not captured)
  float *out;
  int frame, use_tab, mode, owner;  // mode: FillMode
  int pm_bw, pm_dc, pm_dr, pad;     // diagnostic map: see PixMapK
};

struct GroupCtl {
  int bar[kMaxGroups];
  int local_next;         // CU-local work queue (launches without a global queue word)
  int tile_next;          // one-group instantiations: the voxel pass's unit counter (dynamic units)
  int lock;               // TSDF_P2_LOCK builds: one group at a time between the extents barrier and the end of phase 2
  int help_for;           // 0: none; g+1: group g is asked to help with the frame in `help`
  int idle[kMaxGroups];
  int cap_fail[kMaxGroups];  // a row of the group's current frame did not fit the pool
  FrameHdr hdr[kMaxGroups];  // mailbox: the group's first wave fetches the next frame for the others
  HelpReq help;
};

// How the fused kernel gets a frame's pixels into LDS for the voxel pass:
//   TSDF_FILL 0  after phase 1, the rectangle holding every valid pixel is copied into the pool by LDS-DMA
//                (global_load_lds_dwordx4: no VGPR staging, all of a wave's 1 KiB pieces in flight at once); the
//                pool is ONE resource the two groups take turns on through a lock taken on the way into the
//                extents barrier.  The rows come from L2 / Infinity Cache a second time (+10 % fabric traffic).
//   TSDF_FILL 1  row-span capture during phase 1 (see Capture): depth is read exactly once and there is no lock,
//                but the capture's LDS stores sit on the row stream's critical path and a frame whose spans do
//                not fit takes the much slower global gather.  Measured slower on every workload.
// The split kernel always captures (one frame per workgroup: everything fits, nothing to take turns on).
#ifndef TSDF_FILL
#define TSDF_FILL 0
#endif
#ifndef TSDF_P2_LOCK
#define TSDF_P2_LOCK (TSDF_FILL == 0)
#endif
constexpr bool kCaptureFill = TSDF_FILL == 1;

__device__ __forceinline__ int lds_load(const int *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store(int *p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int GWAVES>
__device__ __forceinline__ void group_barrier(int *cnt, int &target) {
  target += GWAVES;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0) {
    __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target)
      __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}

// ---- the workgroup's LDS: everything the CU has --------------------------------------------------
// Per group: projection tables, z table, (AUG) inverse-map products, row table, reduction scratch.
// Shared: control block and the row-span pool, which gets all the rest.
template <int RT, bool AUG, int GROUPS = 2>
struct Lds {
  static constexpr int kZ = RT ? RT : kMaxR;
  static constexpr bool kHasTab = !AUG && (RT == 0 || RT <= kTabR);
  static constexpr int kTabN = kHasTab ? kTabR * kTabR : 16;
  struct PerGroup {
    alignas(16) unsigned pytab[kTabN];
    alignas(16) int pxtab[kTabN];
    alignas(16) unsigned char pyrow[kTabN];
    alignas(16) ZEntry ztab[kZ];
    alignas(16) double atab[AUG ? 12 * kZ : 2];
    alignas(16) unsigned rowtab[kMaxRows];
    alignas(16) float red[16 * kRedStride];  // 16 waves in the split kernel
  };
  static constexpr int kFixed = GROUPS * (int)sizeof(PerGroup) + (int)sizeof(GroupCtl) + 64;
  static constexpr int kPoolFloats = ((kLdsBytes - kFixed) / 16) * 4;
  static constexpr int kPoolUnits = kPoolFloats / 4;
  static_assert(kPoolUnits < 16384, "pool offsets are 14 bits of 4-float units");
  struct Block {
    alignas(16) float pool[kPoolFloats];
    PerGroup pg[GROUPS];
    alignas(16) GroupCtl ctl;
  };
  static_assert(sizeof(Block) <= kLdsBytes, "LDS layout exceeds the CU");
};

template <class PG>
__device__ __forceinline__ Tabs make_tabs(PG &pg) {
  Tabs t;
  t.ztab = (LdsZ)pg.ztab;
  t.pxtab = (LdsCI)pg.pxtab;
  t.pytab = (LdsCU)pg.pytab;
  t.pyrow = (LdsCU8)pg.pyrow;
  t.rowtab = (LdsCU)pg.rowtab;
  t.atab = (LdsCD)pg.atab;
  return t;
}

// ---- per-frame pieces shared by the fused and the split kernel ------------------------------------
// A header that contradicts its payload, or a payload outside the depth buffer, is never read.
__device__ __forceinline__ bool frame_from_header(const FrameHdr &fh, const float *depth, int64_t depth_len, Frame &f) {
  f.l = __builtin_amdgcn_readfirstlane(fh.l);
  f.t = __builtin_amdgcn_readfirstlane(fh.t);
  f.r = __builtin_amdgcn_readfirstlane(fh.r);
  f.b = __builtin_amdgcn_readfirstlane(fh.b);
  const int64_t bw = (int64_t)f.r - f.l, bh = (int64_t)f.b - f.t;  // cannot overflow in 64 bits
  f.bw = (int)bw;
  f.bh = (int)bh;
  f.depth = depth + fh.off0;
  return bw > 0 && bh > 0 && bw <= 0x7fffffff && bh <= 0x7fffffff && bw * bh == fh.off1 - fh.off0 && fh.off0 >= 0 &&
         fh.off1 <= depth_len;
}

__device__ __forceinline__ bool finite32(float v) { return __builtin_fabsf(v) < TSDF_INF; }

// AABB -> grid placement and frame status (degenerate-frame rule of include/tsdf.h).
__device__ __forceinline__ void place_grid(Aabb &ab, int R, const CamK &cam, const float *grid_in, int frame,
                                           Grid &g, int &status) {
  if (!ab.any) {
    status = TSDF_FRAME_DEGENERATE;
    ab.mn[0] = ab.mn[1] = ab.mn[2] = ab.mx[0] = ab.mx[1] = ab.mx[2] = 0.f;
    return;
  }
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
    return;
  }
  const bool mid_ok = finite32(g.mid[0]) && finite32(g.mid[1]) && finite32(g.mid[2]);
  if (!(g.max_l > 0.f) || !(g.max_l < TSDF_INF) || !mid_ok) {
    status = TSDF_FRAME_DEGENERATE;
    g.max_l = g.voxel_len = g.trunc = 0.f;
    if (!mid_ok) g.mid[0] = g.mid[1] = g.mid[2] = 0.f;
  }
}

// one thread per frame writes the scalars
__device__ __forceinline__ void write_frame_outputs(const KArgs &a, int frame, const Grid &g, const Aabb &ab, int status) {
  if (a.max_l) a.max_l[frame] = g.max_l;
  if (a.mid_p) {
    a.mid_p[3 * (int64_t)frame + 0] = g.mid[0];
    a.mid_p[3 * (int64_t)frame + 1] = g.mid[1];
    a.mid_p[3 * (int64_t)frame + 2] = g.mid[2];
  }
  if (a.status) a.status[frame] = status;
  if (a.aabb) {
    float *o = a.aabb + 6 * (int64_t)frame;
    o[0] = ab.mn[0]; o[1] = ab.mn[1]; o[2] = ab.mn[2];
    o[3] = ab.mx[0]; o[4] = ab.mx[1]; o[5] = ab.mx[2];
  }
  if (a.grid) {
    float *q = a.grid + 8 * (int64_t)frame;
    q[0] = g.mid[0]; q[1] = g.mid[1]; q[2] = g.mid[2];
    q[3] = g.max_l; q[4] = g.voxel_len; q[5] = g.trunc; q[6] = 0.f; q[7] = 0.f;
  }
  if (a.ori) {
    float *q = a.ori + 3 * (int64_t)frame;
    q[0] = g.ori[0]; q[1] = g.ori[1]; q[2] = g.ori[2];
  }
}

// Labels: pre/joint_nor.py:8-18 + the clamp of 3D_CNN/train.py:241-242, float32, three separately rounded
// operations; AUG maps the joints with the frame's forward map first (pre/process.py:232-249 does it with the
// cloud's S and R).  Frames that are not OK get 0.5 (see include/tsdf.h).
__device__ __forceinline__ void write_labels(const KArgs &a, int frame, int64_t src, const Grid &g, int status,
                                             const double *xf, int tid, int T) {
  if (!a.gt) return;
  const int nc = 3 * a.n_joints;
  for (int e = tid; e < nc; e += T) {
    const int c = e % 3;
    const float *gj = a.gt + src * nc + (e - c);
    float v = gj[c];
    if (xf) v = (float)affine_row(xf + 4 * c, (double)gj[0], (double)gj[1], (double)gj[2]);
    if (a.gt_aug) a.gt_aug[(int64_t)frame * nc + e] = v;  // the joints in the grid's frame (plain path: a copy)
    float o = 0.5f;
    if (status == TSDF_FRAME_OK) {
      const float m = c == 0 ? g.mid[0] : (c == 1 ? g.mid[1] : g.mid[2]);
      o = __fadd_rn(__fdiv_rn(__fsub_rn(v, m), g.max_l), 0.5f);
      if (a.clamp) {
        o = o < 0.f ? 0.f : o;
        o = o > 1.f ? 1.f : o;
      }
    }
    a.gt_nor[(int64_t)frame * nc + e] = o;
  }
}

// How a frame's pixels reach the voxel pass.
enum FillMode {
  kFillGlobal = 0,  // gather from the crop in global memory (L2)
  kFillSpans = 1,   // row spans captured into the pool during phase 1; per-row entries
  kFillRect = 2     // rectangle of valid pixels staged into the pool by LDS-DMA after phase 1
};

// Voxel-pass constants.  Pixel coordinates are relative to a frame of reference: the bounding box for span
// entries, the rectangle holding every valid pixel otherwise (whole_bbox: the bounding box there too — the
// diagnostic map has to tell "outside the bounding box" from "invalid pixel").
__device__ __forceinline__ VoxK make_voxk(const CamK &cam, const Grid &g, const Frame &f, const Aabb &ab,
                                          int mode, bool whole_bbox, int rect_stride) {
  VoxK vk;
  vk.cx = cam.cx;
  vk.cy = cam.cy;
  vk.it = 1.0 / (double)g.trunc;
  vk.kq = cam.inv_focal * vk.it;
  vk.ncx = -cam.cx;
  vk.eps = cam.eps;
  int c0 = ab.c0, r0 = ab.r0, w = ab.c1 - ab.c0 + 1, h = ab.r1 - ab.r0 + 1;
  if (mode == kFillSpans || whole_bbox) {
    c0 = r0 = 0;
    w = f.bw;
    h = f.bh;
  }
  vk.px0 = f.l + c0;
  vk.py0 = f.t + r0;
  vk.dx = w - 1;
  vk.dy = h - 1;
  vk.stride = mode == kFillRect ? rect_stride : f.bw;
  vk.stride4 = 4 * vk.stride;
  vk.base = mode == kFillGlobal ? r0 * f.bw + c0 : 0;
  vk.dxc = (double)vk.px0 - cam.cx;
  vk.dyc = (double)vk.py0 - cam.cy;
  return vk;
}

// Copy the sh x sw4 pixel rectangle that starts at (row sr0, column sc0) of the frame's crop into the pool by LDS-DMA
// (global_load_lds_dwordx4), NWV waves sharing the rows.  No VGPR staging and no ds_write pass: each wave
// instruction moves up to 64 x 16 B straight into the row-major LDS image (lane i lands at base + 16*i, so lanes are
// laid out as [row][4-pixel group]); all of a wave's pieces are in flight at once.  Sources need only 4-byte
// alignment and EXEC-masked lanes leave their slot untouched (tools/probes/glds_probe.hip).  Completion is counted
// in vmcnt: the caller waits for vmcnt(0) before its barrier.
template <int NWV>
__device__ __forceinline__ void stage_rect_dma(float *stage, const Frame &f, int64_t n_frame, int sc0, int sr0, int sh,
                                               int sw4, int wave, int lane) {
  const int ng = sw4 >> 2;                       // 4-pixel groups per row
  const int64_t base_idx = (int64_t)sr0 * f.bw + sc0;
  if (ng <= 64) {
    const int rows_per = 64 / ng;                // rows one wave instruction covers
    const int rsub = lane / ng, cg = lane - rsub * ng;
    const int nblk = (sh + rows_per - 1) / rows_per;
    for (int blk = wave; blk < nblk; blk += NWV) {
      const int R0 = blk * rows_per;             // scalar
      const int row = R0 + rsub;
      const int64_t gi = base_idx + (int64_t)row * f.bw + 4 * cg;
      const bool act = rsub < rows_per && row < sh;
      float *ldst = stage + R0 * sw4;            // wave-uniform LDS base
      if (act && gi + 3 < n_frame) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(f.depth + gi),
                                         (__attribute__((address_space(3))) void *)ldst, 16, 0, TSDF_DMA_AUX);
      } else if (act) {  // the 16-byte piece would run past the end of the frame: element copies
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (gi + e < n_frame) ldst[(rsub * ng + cg) * 4 + e] = f.depth[gi + e];
      }
    }
  } else {
    for (int row = wave; row < sh; row += NWV) {
      for (int c4 = 0; c4 < ng; c4 += 64) {
        const int cg = c4 + lane;
        const int64_t gi = base_idx + (int64_t)row * f.bw + 4 * cg;
        float *ldst = stage + row * sw4 + 4 * c4;  // wave-uniform
        if (cg < ng && gi + 3 < n_frame) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(f.depth + gi),
                                           (__attribute__((address_space(3))) void *)ldst, 16, 0, TSDF_DMA_AUX);
        } else if (cg < ng) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (gi + e < n_frame) ldst[lane * 4 + e] = f.depth[gi + e];
        }
      }
    }
  }
}

// Per-frame tables (true divisions; (x,z)/(y,z) pairs spread over the T participating threads).
template <int LAYOUT, bool AUG, class PG>
__device__ __forceinline__ void fill_tables(PG &pg, const Grid &g, const CamK &cam, const VoxK &vk, int R,
                                            bool use_tab, bool spans, const double *xf, int vt, int T) {
  const double vl = (double)g.voxel_len;
  const double ox = (double)g.ori[0], oy = (double)g.ori[1], oz = (double)g.ori[2];
  if (vt < R) {
    const double v_z = oz + (double)vt * vl;  // :28
    ZEntry ze;
    ze.q = -cam.focal / v_z;                    // :30
    ze.vzs = v_z * vk.it;
    ze.negthr = f32_round_up(-v_z);             // pd < -v_z  <=>  w_z > v_z  (:65)
    ze.pad = 0.f;
    pg.ztab[vt] = ze;
  }
  if constexpr (AUG) {
    // per (axis, index): the three products of the inverse map and v' - b of the forward map: see phase2_aug
    const double *inv = xf + 12;
    for (int e = vt; e < 12 * R; e += T) {
      const int axis = e / (4 * R), rem = e - axis * 4 * R, i = rem >> 2, row = rem & 3;
      const double o_a = axis == 0 ? ox : (axis == 1 ? oy : oz);
      const double vp = o_a + (double)i * vl;                    // :26-28
      // the axis the lanes index is stored lane-major (lane_slot), the two the slices / slabs index as [index][4]
      const int at = axis == (LAYOUT == 0 ? 0 : 2) ? 4 * R * axis + lane_slot(i, row, R / 4) : e;
      if (row < 3) {
        const double prod = inv[4 * row + axis] * vp;
        pg.atab[at] = axis == 2 ? prod + inv[4 * row + 3] : prod;   // the z entries carry the translation
      } else {
        pg.atab[at] = vp - xf[4 * axis + 3];
      }
    }
  }
  if (use_tab) {
    for (int e = vt; e < R * R; e += T) {
      const int z = e / R, i = e - z * R;
      const double q = -cam.focal / (oz + (double)z * vl);                              // :30
      const double vx = ox + (double)i * vl, vy = oy + (double)i * vl;                  // :26-27
      const int ti = tab_index<LAYOUT>(i, z, R);
      pg.pxtab[ti] = col_code(project_rel(vx, q, cam.cx, vk.px0, vk.dx));               // :31
      const int ry = project_rel(-vy, q, cam.cy, vk.py0, vk.dy);                        // :32
      if (spans) {
        pg.pytab[ti] = ry >= 0 ? pg.rowtab[ry] : 0u;
        pg.pyrow[ti] = (unsigned char)(ry >= 0 ? ry : 0);
      } else {
        pg.pytab[ti] = (unsigned)ry;
      }
    }
  }
}

// Work queue.  Frames beyond the first one per group are handed out dynamically (frame cost varies ~3x with
// the hand's size; a static 4-frames-per-CU split left a 25 % tail).  An eager launch draws tickets from a
// device-global word that belongs to its (device, stream) pair — launches of one stream run in order, so the
// word is never shared (host side: queue_word()).  The word is 64 bits: the launch's EPOCH (its number on that
// word, counted by the host, never 0) in the high half, the ticket counter in the low half.  A drawer that finds
// another epoch in the word — a fresh word, the previous launch's final state, or whatever a launch that died
// mid-flight (or anything else) left there — installs {epoch, 1} by compare-and-swap and takes ticket 0; everybody
// else just adds (two atomics for the first drawers, one for the rest).  In the normal course of things even that does
// not happen: the drawer of a launch's last ticket leaves the NEXT epoch installed.  So no state of the word can make a
// launch skip or repeat a frame (round 3 reset the word from the drawer of ticket n-1: a launch that never got there left
// every later launch of the stream short of frames, silently — tests/test_parity_gpu.py poisons the word).
// A launch without a word (captured into a graph, or more streams than words) shares frames inside each CU only,
// through a counter in LDS.
constexpr int kQueueSlots = 1024;
__device__ unsigned long long g_queue[kQueueSlots];

__device__ __forceinline__ unsigned int queue_next_epoch(unsigned int e) { return e + 1u ? e + 1u : 1u; }  // (host: next_epoch)

__device__ __forceinline__ unsigned int queue_ticket(unsigned long long *q, unsigned int epoch, int n) {
  // Only read-modify-write atomics read the word.  (A first version re-read it with a plain agent-scope load between
  // its attempts: the other XCDs' L2s may keep serving such a load a stale line for tens of microseconds — the
  // compare-and-swap, done at the memory side, then fails against the fresh value again and again.  1024 frames -> 32^3
  // went from 128 to 179 us that way, bimodally; profiles/r04/ab_queue32.log.  The value a failed compare-and-swap
  // returns IS the fresh observation.)
  const unsigned long long mine = (unsigned long long)epoch << 32;
  unsigned int t;
  for (;;) {
    const unsigned long long old = atomicAdd(q, 1ull);
    if ((unsigned int)(old >> 32) == epoch) {   // the common case: one atomic
      t = (unsigned int)old;
      break;
    }
    unsigned long long cur = old + 1;   // the word is not in this launch's epoch; this is what our add left there
    bool installed = false;
    for (;;) {
      const unsigned long long seen = atomicCAS(q, cur, mine | 1ull);
      if (seen == cur) {
        installed = true;                                               // installed here: ticket 0 is ours
        break;
      }
      if ((unsigned int)(seen >> 32) == epoch) break;                   // somebody installed it: draw again
      cur = seen;
    }
    if (installed) {
      t = 0u;
      break;
    }
  }
  // Exactly n tickets are drawn per launch (every group that got a positional frame draws until it fails once), so the
  // drawer of ticket n-1 is the last one to touch the word: it leaves the NEXT launch's epoch installed, and that launch
  // — the stream's next one, by the host's count — pays one atomic per ticket from its first draw on.  (Without this
  // every launch started with the install dance: +2.4 % on 1024 full frames -> 32^3, +5 % on crops.)  A launch that
  // never gets here leaves a foreign epoch behind, which is what the dance is for.
  if (t == (unsigned int)(n - 1))
    __hip_atomic_store(q, (unsigned long long)queue_next_epoch(epoch) << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return t;
}

// Persistent kernel, one 1024-thread workgroup per CU.  Its two 512-thread groups each walk their own
// frames through
//     stream rows (+ capture the valid spans into the LDS pool) -> extents -> glue    memory-bound
//     tables -> voxel pass from the pool                                               VALU/store-bound
// independently of each other, so one group's row streaming overlaps the other group's voxel arithmetic
// and stores on the same CU.
// (The read-only inputs are passed as separate __restrict__ parameters as well as inside KArgs: alias information
// does not survive a by-value struct, and without it the compiler may not use scalar loads for wave-uniform
// reads — the per-frame transform of the augmented form became 700 vector loads in the unrolled row pass.)
template <int RT, int LAYOUT, bool AUG, bool DBG, int GROUPS>
__global__ __launch_bounds__(kWG) void tsdf_fused_kernel(const KArgs a, const float *__restrict__ in_depth,
                                                         const int64_t *__restrict__ in_offsets,
                                                         const int32_t *__restrict__ in_headers,
                                                         const double *__restrict__ in_xforms) {
  constexpr int kGroups = GROUPS, kGW = kWG / GROUPS, kGWaves = kGW / 64;  // (shadow the file-scope defaults)
  using L = Lds<RT, AUG, GROUPS>;
  __shared__ typename L::Block lds;

  const int R = RT ? RT : a.R;
  const CamK &cam = a.cam;
  const int n = a.n;
  const int tid = threadIdx.x, lane = tid & 63, gtid = tid & (kGW - 1);
  const int group = __builtin_amdgcn_readfirstlane(tid / kGW);
  const int gwave = __builtin_amdgcn_readfirstlane((tid >> 6) & (kGWaves - 1));
  auto &pg = lds.pg[group];
  GroupCtl &ctl = lds.ctl;

  if (tid == 0) {
    for (int i = 0; i < kGroups; ++i) ctl.bar[i] = ctl.idle[i] = ctl.cap_fail[i] = 0;
    ctl.lock = 0;
    ctl.help_for = 0;
    ctl.local_next = kGroups;
  }
  __syncthreads();  // the only workgroup-wide barrier
  int bar_target = 0;
  auto gsync = [&]() { group_barrier<kGWaves>(&ctl.bar[group], bar_target); };

  Capture cap;
  cap.pool = (LdsF)lds.pool;
  cap.rowtab = (LdsU)pg.rowtab;
  cap.fail = (LdsI)&ctl.cap_fail[group];
  cap.cap4 = L::kPoolUnits / (kWG / 64);
  cap.base4 = __builtin_amdgcn_readfirstlane(tid >> 6) * cap.cap4;
  cap.on = false;

  const int n_static = gridDim.x * kGroups;  // frames handed out by position (the first one per group)

  int iter = 0;
  (void)iter;
  for (;; ++iter) {
    // ---- the group's first wave fetches the next frame (index + header) and posts it in LDS ----
    if (gwave == 0) {
      int fr;
      if (iter == 0) {
        fr = blockIdx.x + gridDim.x * group;
      } else if (a.queue) {
        unsigned int t = 0;
        if (lane == 0) {
          t = queue_ticket(a.queue, a.qepoch, n);
        }
        fr = n_static + (int)__builtin_amdgcn_readfirstlane(t);
      } else {
        int t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(&ctl.local_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int64_t f64i = (int64_t)blockIdx.x + (int64_t)gridDim.x * __builtin_amdgcn_readfirstlane(t);
        fr = f64i < n ? (int)f64i : n;
      }
      FrameHdr m;
      m.frame = fr < n ? fr : -1;
      m.l = m.t = m.r = m.b = m.pad = 0;
      m.off0 = m.off1 = m.src = 0;
      if (fr < n) fetch_header(a, in_offsets, in_headers, fr, m);
      if (lane == 0) {
        ctl.hdr[group] = m;
        ctl.cap_fail[group] = 0;
      }
    }
    gsync();
    const FrameHdr fh = ctl.hdr[group];
    const int frame = __builtin_amdgcn_readfirstlane(fh.frame);
    if (frame < 0) break;
    TSDF_STAMP(kGroups * iter + group, 0);

    Frame f;
    const bool hdr_ok = frame_from_header(fh, in_depth, a.depth_len, f);
    float *out = a.tsdf ? a.tsdf + (int64_t)frame * 3 * R * R * R : nullptr;
    const bool want_vol = !a.aabb_only && out;

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
    const double *xf = AUG ? in_xforms + 24 * (int64_t)frame : nullptr;

    bool holds_lock = false;  // group-uniform
    cap.on = false;
    if (!hdr_ok) {
      status = TSDF_FRAME_BAD_HEADER;  // group-uniform
    } else {
      float fin[kExt];
      cap.on = kCaptureFill && want_vol && f.bw <= kMaxCapW && f.bh <= kMaxRows;  // group-uniform
      auto sync_ext = [&]() {
#if TSDF_P2_LOCK
        // The group's first wave takes the pool lock on its way into the extents barrier, so the wait for
        // the other group's voxel pass hides behind this group's own slowest wave.
        if (want_vol && gwave == 0 && lane == 0) {
          while (atomicCAS(&ctl.lock, 0, 1) != 0) __builtin_amdgcn_s_sleep(2);
        }
#endif
        gsync();
      };
      phase1_extents<kGWaves, AUG, kCaptureFill>(f, cam, 0, f.bh, pg.red, fin, gwave, sync_ext, cap,
                                                 kGroups * iter + group, xf);
      holds_lock = TSDF_P2_LOCK && want_vol;
      ab = aabb_from_extents(fin);
      TSDF_STAMP(kGroups * iter + group, 4);
      place_grid(ab, R, cam, a.grid_in, frame, g, status);
    }

    if (gtid == 0) write_frame_outputs(a, frame, g, ab, status);
    write_labels(a, frame, fh.src, g, status, xf, gtid, kGW);

    // the frame's spans are in the pool iff capture was on and every row fitted (group-uniform: the flag
    // was written before the extents barrier)
    bool captured = cap.on && lds_load(&ctl.cap_fail[group]) == 0;
    captured = __builtin_amdgcn_readfirstlane(captured);

    if (want_vol) {
      if (status != TSDF_FRAME_OK) {
        zero_volume(out, R, gtid, kGW, 0, 1);
        if constexpr (DBG) {
          if (a.pixmap)
            for (int i = gtid; i < R * R * R; i += kGW) a.pixmap[(int64_t)frame * R * R * R + i] = -1;
        }
      } else {
        // The thread's index for the tables and the voxel pass, hidden from loop-invariant code motion:
        // otherwise every constant derived from it (a dozen float64 conversions of voxel indices) is
        // computed once before the frame loop and then occupies registers, or scratch, all through phase 1.
        int vt = gtid;
        asm volatile("" : "+v"(vt));
        // ---- TSDF_FILL 0: stage the valid pixels' rectangle into the pool by LDS-DMA.  The group holds the pool
        // lock here (taken at the extents barrier), so the whole pool is its own.
        int mode = captured ? kFillSpans : kFillGlobal;
        // the staged image: the rectangle of valid pixels (the whole bounding box in the diagnostic build),
        // row-major from the start of the pool, rows padded to a multiple of 4 pixels (16-byte DMA pieces)
        const int sc0 = DBG ? 0 : ab.c0, sr0 = DBG ? 0 : ab.r0;
        const int sw = DBG ? f.bw : ab.c1 - ab.c0 + 1, sh = DBG ? f.bh : ab.r1 - ab.r0 + 1;
        const int sw4 = (sw + 3) & ~3;
        const bool staged = !kCaptureFill && (int64_t)sw4 * sh <= L::kPoolFloats;  // group-uniform
        if (staged) {
          mode = kFillRect;
          stage_rect_dma<kGWaves>(lds.pool, f, fh.off1 - fh.off0, sc0, sr0, sh, sw4, gwave, lane);
        }
        const VoxK vk = make_voxk(cam, g, f, ab, mode, DBG, sw4);
        const bool use_tab = !AUG && R <= kTabR;  // uniform
        TSDF_STAMP(kGroups * iter + group, 5);
        fill_tables<LAYOUT, AUG>(pg, g, cam, vk, R, use_tab, mode == kFillSpans, xf, vt, kGW);
        TSDF_STAMP(kGroups * iter + group, 6);
        // the copy was issued before the tables were computed (worth 1.1 % of the launch, round 1)
        if (staged) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // LDS-DMA completion is counted in vmcnt
        TSDF_STAMP(kGroups * iter + group, 7);
        if constexpr (kGroups == 2 && TSDF_TAIL_HELP) {
          if (gwave == 0 && lane == 0) ctl.hdr[group].pad = lds_load(&ctl.idle[group ^ 1]);
        }
        if constexpr (kGroups == 1) {
          if (gwave == 0 && lane == 0) lds_store(&ctl.tile_next, 0);   // ordered by the barrier below
        }
        gsync();
        TSDF_STAMP(kGroups * iter + group, 8);
        bool helped = false;  // group-uniform
        if constexpr (kGroups == 2 && TSDF_TAIL_HELP) {
          // the other group is idle (so this is the launch's last frame on this CU): split the volume with it
          if (__builtin_amdgcn_readfirstlane(ctl.hdr[group].pad)) {
            helped = true;
            if (gwave == 0 && lane == 0) {
              HelpReq hq;
              hq.g = g;
              hq.vk = vk;
              hq.src = f.depth + vk.base;
              hq.mode = mode;
              hq.owner = group;
              hq.out = out;
              hq.frame = frame;
              hq.use_tab = use_tab;
              hq.pm_bw = f.bw;
              hq.pm_dc = vk.px0 - f.l;
              hq.pm_dr = vk.py0 - f.t;
              hq.pad = 0;
              ctl.help = hq;
              lds_store(&ctl.help_for, (group ^ 1) + 1);
            }
          }
        }
        TSDF_STAMP_VAL(kGroups * iter + group, 10, helped ? 2 : 1);
        const Tabs tb = make_tabs(pg);
        PixMapK pm;
        pm.out = DBG && a.pixmap ? (GlobalPix)(a.pixmap + (int64_t)frame * R * R * R) : (GlobalPix) nullptr;
        pm.bw = f.bw;
        pm.dc = vk.px0 - f.l;
        pm.dr = vk.py0 - f.t;
        auto run2 = [&](auto src) {
          if constexpr (kGroups == 2 && TSDF_TAIL_HELP) {
            if (__builtin_expect(helped, 0)) {
              if constexpr (AUG) {
                phase2_aug<LAYOUT, 2 * kGW>(g, cam, vk, R, xf, tb, src, (GlobalOut)out, vt, 0, R);
              } else {
                phase2<LAYOUT, 2 * kGW, DBG>(g, cam, vk, R, tb, use_tab, src, (GlobalOut)out, vt, 0, R, pm);
              }
              return;
            }
          }
          if constexpr (AUG) {
            phase2_aug<LAYOUT, kGW>(g, cam, vk, R, xf, tb, src, (GlobalOut)out, vt, 0, R,
                                    kGroups == 1 ? &ctl.tile_next : nullptr, kGroups * iter + group);
          } else {
            phase2<LAYOUT, kGW, DBG>(g, cam, vk, R, tb, use_tab, src, (GlobalOut)out, vt, 0, R, pm,
                                     kGroups == 1 ? &ctl.tile_next : nullptr);
          }
        };
        if (mode == kFillRect) {
          run2(LdsRect{(LdsSrc)lds.pool});
        } else if (mode == kFillSpans) {
          run2((LdsSrc)lds.pool);
        } else {
          run2((GlobalSrc)(f.depth + vk.base));
        }
      }
    }
    TSDF_STAMP(kGroups * iter + group, 9);
    // Close the frame: every wave of the group has left the LDS it shares (pool spans, tables, `red`, the
    // header mailbox) before the next frame rewrites them.
    gsync();
    if (holds_lock && gwave == 0 && lane == 0)
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
      const Tabs tb = make_tabs(lds.pg[group ^ 1]);  // the owner's tables
      PixMapK pm;
      pm.out = DBG && a.pixmap ? (GlobalPix)(a.pixmap + (int64_t)hq.frame * R * R * R) : (GlobalPix) nullptr;
      pm.bw = hq.pm_bw;
      pm.dc = hq.pm_dc;
      pm.dr = hq.pm_dr;
      auto run2 = [&](auto src) {
        if constexpr (AUG) {
          phase2_aug<LAYOUT, 2 * kGW>(hq.g, cam, hq.vk, R, in_xforms + 24 * (int64_t)hq.frame, tb, src, hout,
                                      kGW + gtid, 0, R);
        } else {
          phase2<LAYOUT, 2 * kGW, DBG>(hq.g, cam, hq.vk, R, tb, hq.use_tab != 0, src, hout, kGW + gtid, 0, R, pm);
        }
      };
      if (hq.mode == kFillRect) {
        run2(LdsRect{(LdsSrc)lds.pool});
      } else if (hq.mode == kFillSpans) {
        run2((LdsSrc)lds.pool);
      } else {
        run2(hsrc);
      }
    }
  }
}

// Split kernel for small batches (n <= CUs/2): a.split workgroups per frame.  Every workgroup streams the
// whole frame with its 16 waves (the frame comes from L2 / Infinity Cache for all but the first), captures
// the spans, places the grid, fills the tables — all redundantly, so no workgroup ever waits for another —
// and voxelizes a.per slices of the slow axis.  Results are bit-identical to the fused kernel's: the
// extents are min/max reductions (order-free) and the per-voxel code is the same.
//
// XCHG: the row stream itself is split as well.  Streaming a whole frame through ONE workgroup is a chain of
// dependent cold misses (14.8 of the 17 us such a launch took), so here workgroup p of a frame streams only band p
// of its rows, publishes its 10 partial extents in a mailbox in device memory — tagged with a number unique to the
// launch on its stream — and collects the other bands' (min/max: order-free, so the result is bit-identical).
// The wait is BOUNDED: a workgroup that does not see all mailboxes in time (its siblings are not resident yet,
// e.g. behind another kernel) streams the whole frame itself, exactly like the non-XCHG form — nothing can
// deadlock.  The valid-pixel rectangle then comes into LDS by LDS-DMA (its rows were just read by the siblings:
// L2 hits).  The mailboxes are a per-stream slice of a device global owned by the library (host: xchg_for());
// launches that cannot have one (stream capture, too many streams) use the redundant form.
constexpr int kXchgParts = 16;       // mailboxes per frame (upper bound of a.split)
#ifndef TSDF_XCHG_FRAMES
#define TSDF_XCHG_FRAMES 128
#endif
constexpr int kXchgFrames = TSDF_XCHG_FRAMES;  // frames per split launch at most (n <= CUs/2)
constexpr int kXchgBox = 16;         // floats per mailbox: 10 extents, tag, pad (64 bytes: one mailbox per line)
constexpr int kXchgPolls = 4000;     // bound of the wait (x ~0.1 us)

template <int RT, int LAYOUT, bool AUG, bool XCHG>
__global__ __launch_bounds__(kWG) void tsdf_split_kernel(const KArgs a, const float *__restrict__ in_depth,
                                                         const int64_t *__restrict__ in_offsets,
                                                         const int32_t *__restrict__ in_headers,
                                                         const double *__restrict__ in_xforms) {
  using L = Lds<RT, AUG>;
  __shared__ typename L::Block lds;

  const int R = RT ? RT : a.R;
  const CamK &cam = a.cam;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frame = blockIdx.x / a.split, part = blockIdx.x - frame * a.split;
  auto &pg = lds.pg[0];
  GroupCtl &ctl = lds.ctl;
  if (tid == 0) ctl.cap_fail[0] = 0;
  FrameHdr fh;
  {
    fh.frame = frame;
    fh.pad = 0;
    fetch_header(a, in_offsets, in_headers, frame, fh);  // uniform: scalar loads
  }
  __syncthreads();
  TSDF_STAMP(0, 0);
  Frame f;
  const bool hdr_ok = frame_from_header(fh, in_depth, a.depth_len, f);
  float *out = a.tsdf ? a.tsdf + (int64_t)frame * 3 * R * R * R : nullptr;
  const bool want_vol = !a.aabb_only && out;

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
  const double *xf = AUG ? in_xforms + 24 * (int64_t)frame : nullptr;

  Capture cap;
  cap.pool = (LdsF)lds.pool;
  cap.rowtab = (LdsU)pg.rowtab;
  cap.fail = (LdsI)&ctl.cap_fail[0];
  cap.cap4 = L::kPoolUnits / (kWG / 64);
  cap.base4 = wave * cap.cap4;
  cap.on = false;
  if (!hdr_ok) {
    status = TSDF_FRAME_BAD_HEADER;
  } else {
    float fin[kExt];
    auto sync_all = [&]() { __syncthreads(); };
    bool have = false;  // workgroup-uniform: fin holds the frame's extents
    // (Small crops streamed whole by every workgroup of the frame, skipping the exchange, were tried in round 3: a
    // 16-crop launch takes 13.5 us back to back either way, a 1-crop launch 11.4 instead of 11.9.)
    if constexpr (XCHG) {
      // ---- band `part` of the rows -> partial extents -> mailbox ----
      const int S = a.split;
      const int rb = (int)((int64_t)f.bh * part / S), re = (int)((int64_t)f.bh * (part + 1) / S);
      phase1_extents<kWG / 64, AUG, false>(f, cam, rb, re, pg.red, fin, wave, sync_all, cap, 0, xf);
      float *boxes = a.xchg + (int64_t)frame * (kXchgParts * kXchgBox);
      // Every word goes out as an agent-scope atomic (written through to where the other XCDs' workgroups read it),
      // the tag after the data has been acknowledged: a release without writing the whole L2 back.
      if (tid == 0) {
        unsigned int *box = reinterpret_cast<unsigned int *>(boxes + part * kXchgBox);
#pragma unroll
        for (int i = 0; i < kExt; ++i)
          __hip_atomic_store(box + i, __float_as_uint(fin[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(box + kExt, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      TSDF_STAMP(0, 13);
      // ---- collect the siblings' (wave 0: lane q watches mailbox q), bounded ----
      if (wave == 0) {
        const int lane = tid & 63;
        const bool mine = lane < S;
        const unsigned int *box = reinterpret_cast<const unsigned int *>(boxes + (mine ? lane : 0) * kXchgBox);
        bool ready = !mine || lane == part;
        for (int it = 0; it < a.polls; ++it) {
          if (!ready) ready = __hip_atomic_load(box + kExt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.seq;
          if (__all(ready)) break;
          __builtin_amdgcn_s_sleep(1);
        }
        const bool all = __all(ready);
        if (all) {
          // the tag was written after its data was acknowledged, and these loads bypass this XCD's L2 as well
          const bool use = mine && lane != part;
#pragma unroll
          for (int i = 0; i < 5; ++i) {
            const float v = __uint_as_float(__hip_atomic_load(box + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            fin[i] = vmin(fin[i], row0_min(use ? v : TSDF_INF));
          }
#pragma unroll
          for (int i = 5; i < 10; ++i) {
            const float v = __uint_as_float(__hip_atomic_load(box + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            fin[i] = vmax(fin[i], row0_max(use ? v : -TSDF_INF));
          }
          if (lane == 0) {  // (the row table is free in this form; `red` may still be read by a slow wave)
#pragma unroll
            for (int i = 0; i < kExt; ++i) pg.rowtab[i] = __float_as_uint(fin[i]);
          }
        }
        if (lane == 0) ctl.cap_fail[1] = all ? 1 : 0;   // (a free word of the control block)
      }
      __syncthreads();
      have = lds_load(&ctl.cap_fail[1]) != 0;
      have = __builtin_amdgcn_readfirstlane(have);
      if (have) {
#pragma unroll
        for (int i = 0; i < kExt; ++i) fin[i] = __uint_as_float(pg.rowtab[i]);
      }
    }
    TSDF_STAMP(0, 12);
    if (!have) {
      // the redundant form: this workgroup streams the whole frame (and captures the valid row windows)
      cap.on = !XCHG && want_vol && f.bw <= kMaxCapW && f.bh <= kMaxRows;
      phase1_extents<kWG / 64, AUG, !XCHG>(f, cam, 0, f.bh, pg.red, fin, wave, sync_all, cap, 0, xf);
    }
    ab = aabb_from_extents(fin);
    place_grid(ab, R, cam, a.grid_in, frame, g, status);
  }
  TSDF_STAMP(0, 4);
  if (part == 0) {
    if (tid == 0) write_frame_outputs(a, frame, g, ab, status);
    write_labels(a, frame, fh.src, g, status, xf, tid, kWG);
  }
  if (!want_vol) return;
  if (status != TSDF_FRAME_OK) {
    zero_volume(out, R, tid, kWG, part, a.split);
    return;
  }
  bool captured = cap.on && lds_load(&ctl.cap_fail[0]) == 0;
  captured = __builtin_amdgcn_readfirstlane(captured);
  int mode = captured ? kFillSpans : kFillGlobal;
  const int sw = ab.c1 - ab.c0 + 1, sh = ab.r1 - ab.r0 + 1, sw4 = (sw + 3) & ~3;
  const bool staged = XCHG && (int64_t)sw4 * sh <= L::kPoolFloats;
  if (staged) {
    mode = kFillRect;
    stage_rect_dma<kWG / 64>(lds.pool, f, fh.off1 - fh.off0, ab.c0, ab.r0, sh, sw4, wave, tid & 63);
  }
  const VoxK vk = make_voxk(cam, g, f, ab, mode, false, sw4);
  const bool use_tab = !AUG && R <= kTabR;
  fill_tables<LAYOUT, AUG>(pg, g, cam, vk, R, use_tab, mode == kFillSpans, xf, tid, kWG);
  if (staged) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // LDS-DMA completion is counted in vmcnt
  __syncthreads();
  TSDF_STAMP(0, 6);
  const Tabs tb = make_tabs(pg);
  const int sb = part * a.per, se = sb + a.per < R ? sb + a.per : R;
  PixMapK pm;
  pm.out = (GlobalPix) nullptr;
  pm.bw = f.bw;
  pm.dc = pm.dr = 0;
  auto run2 = [&](auto src) {
    if constexpr (AUG) {
      phase2_aug<LAYOUT, kWG>(g, cam, vk, R, xf, tb, src, (GlobalOut)out, tid, sb, se);
    } else {
      phase2<LAYOUT, kWG, false>(g, cam, vk, R, tb, use_tab, src, (GlobalOut)out, tid, sb, se, pm);
    }
  };
  if (mode == kFillRect) {
    run2(LdsRect{(LdsSrc)lds.pool});
  } else if (mode == kFillSpans) {
    run2((LdsSrc)lds.pool);
  } else {
    run2((GlobalSrc)(f.depth + vk.base));
  }
  TSDF_STAMP(0, 9);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TSDF_STAMP(0, 11);
}

// Label normalisation on its own (pre/joint_nor.py:8-18) and its inverse (3D_CNN/train.py:263-266).
__global__ void tsdf_normalize_kernel(const float *__restrict__ gt, const float *__restrict__ max_l,
                                      const float *__restrict__ mid_p, int64_t total, int nc, int clamp, int inverse,
                                      float *__restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int64_t frame = e / nc;
  const int c = (int)(e - frame * nc) % 3;
  const float ml = max_l[frame], m = mid_p[3 * frame + c], v = gt[e];
  float o;
  if (inverse) {
    o = ml > 0.f ? __fadd_rn(__fmul_rn(__fsub_rn(v, 0.5f), ml), m) : m;
  } else if (ml > 0.f) {
    o = __fadd_rn(__fdiv_rn(__fsub_rn(v, m), ml), 0.5f);
    if (clamp) {
      o = o < 0.f ? 0.f : o;
      o = o > 1.f ? 1.f : o;
    }
  } else {
    o = 0.5f;
  }
  out[e] = o;
}


const tsdf_cam kDefaultCam = {241.42, 160.0, 120.0, 1.0f, 3.0f};

// CU count of the current device (cached per device id; a racing first call computes the same value)
int num_cus() {
  static std::atomic<int> cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int v = cached[dev].load(std::memory_order_relaxed);
  if (v == 0) {
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cached[dev].store(v, std::memory_order_relaxed);
  }
  return v;
}

// The code object holds gfx950 kernels only: any other device is "no usable device", not a launch error.
// (Cached per device id; a racing first call computes the same value.)
int check_device(int *dev_out) {
  static std::atomic<int> arch_state[64];  // 0 unknown, 1 gfx950, -1 something else
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return TSDF_ERR_NO_DEVICE;
  }
  *dev_out = dev;
  if (dev < 0 || dev >= 64) return TSDF_OK;  // beyond the cache: let the launch decide
  int st = arch_state[dev].load(std::memory_order_relaxed);
  if (st == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      (void)hipGetLastError();
      return TSDF_ERR_NO_DEVICE;
    }
    st = strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : -1;
    arch_state[dev].store(st, std::memory_order_relaxed);
  }
  return st == 1 ? TSDF_OK : TSDF_ERR_NO_DEVICE;
}

// ---- work-queue words: one per (device, stream) ------------------------------------------------------
// Launches of one stream execute in order, so a word owned by the stream is never shared by two running
// launches, however many are in flight.  hipStreamPerThread is one handle for a different stream per thread:
// it is keyed by the calling thread as well.  No word (null) means "use CU-local queues": a launch that is
// being captured into a graph (its node may later run anywhere, any number of times), a device beyond the
// table, or more live streams than words.
struct StreamSlots {
  tsdf_host::SlotTable<kQueueSlots> table;   // (stream, thread) -> slot; its own mutex (tsdf_host.inc)
  std::mutex mu;                             // guards the device-side resources below
  unsigned long long *base = nullptr;        // device address of g_queue on this device
  float *xchg = nullptr;                     // split-kernel mailboxes: device address of g_xchg (see xchg_for)
  unsigned int xchg_seq[64] = {0};
};
StreamSlots g_slots[64];

// Index of the (device, stream) pair in the table, or -1 (capturing, no room, beyond the table).  `release` forgets
// the pair instead.  Cost: one hipStreamIsCapturing plus a linear scan of the entries in use under the table's mutex —
// a few tens of nanoseconds with the handful of streams a process normally launches from, O(streams) if hundreds of
// streams are kept alive at once (release the ones that are done: tsdf_stream_release).
int stream_slot(int dev, hipStream_t s, bool release) {
  if (dev < 0 || dev >= 64) return -1;
  StreamSlots &t = g_slots[dev];
  const bool per_thread = s == hipStreamPerThread;
  if (release) {
    t.table.release(static_cast<const void *>(s), per_thread);
    return -1;
  }
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cs) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  if (cs != hipStreamCaptureStatusNone) return -1;
  {
    std::lock_guard<std::mutex> lock(t.mu);
    if (!t.base) {
      void *p = nullptr;
      if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_queue)) != hipSuccess || !p) {
        (void)hipGetLastError();
        return -1;
      }
      t.base = static_cast<unsigned long long *>(p);
    }
  }
  return t.table.acquire(static_cast<const void *>(s), per_thread);
}

// This launch's work-queue word and its epoch on it (see queue_ticket), or null.
unsigned long long *queue_word(int dev, hipStream_t s, unsigned int *epoch) {
  const int i = stream_slot(dev, s, false);
  if (i < 0) return nullptr;
  *epoch = g_slots[dev].table.next_epoch(i);
  return g_slots[dev].base + i;
}

// Mailboxes of the split kernel's XCHG form for the launch being issued on (dev, s), and the tag it must use; null
// when the launch cannot have any (stream capture, a stream beyond the first kXchgSlots).  The workspace — kXchgSlots x
// 128 KiB — is a zero-initialised device global (g_xchg: 8 MiB of the code object's .bss, placed when the library is
// loaded), so no call ever allocates, clears or synchronises anything (rounds 2-3 allocated it with hipMalloc + hipMemset
// on the first small-batch call: the one exception to "allocates nothing" the header had to document).  Launches of
// one stream are ordered, so tags only ever grow inside a slot.
constexpr int kXchgSlots = 64;
constexpr size_t kXchgSlotFloats = (size_t)kXchgFrames * kXchgParts * kXchgBox;
__device__ float g_xchg[kXchgSlots * kXchgSlotFloats];
float *xchg_for(int dev, hipStream_t s, unsigned int *seq) {
  const int i = stream_slot(dev, s, false);
  if (i < 0 || i >= kXchgSlots) return nullptr;
  StreamSlots &t = g_slots[dev];
  std::lock_guard<std::mutex> lock(t.mu);
  if (!t.xchg) {
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_xchg)) != hipSuccess || !p) {
      (void)hipGetLastError();
      return nullptr;
    }
    t.xchg = static_cast<float *>(p);
  }
  unsigned int v = ++t.xchg_seq[i];
  if (v == 0) v = ++t.xchg_seq[i];  // 0 is what fresh mailboxes hold
  *seq = v;
  return t.xchg + (size_t)i * kXchgSlotFloats;
}

// Workgroups per frame and slices per workgroup for the split kernel (0: use the fused kernel).
void split_plan(int n, int R, int cus, int *split, int *per) {
  *split = 0;
  *per = R;
  // (experiments: TSDF_SPLIT_MAXN overrides the batch size up to which frames are split; beyond CUs/2 the fused
  // kernel measured faster: 256 full frames 38 us fused, 46 us split — tools/exp_split_threshold.py)
  static const int max_n = [] {
    const char *e = getenv("TSDF_SPLIT_MAXN");
    return e ? atoi(e) : -1;
  }();
  const int limit = max_n >= 0 ? max_n : cus / 2;
  if (n > limit || n > kXchgFrames) return;
  const int G = R * (R / 4);
  const int sstep = (G <= kWG && kWG % G == 0) ? kWG / G : 1;  // slices one pass of the workgroup covers
  const int rounds = (R + sstep - 1) / sstep;
  int S = cus / n;
  if (S < 2) S = 2;
  if (S > rounds) S = rounds;
  if (S > kXchgParts) S = kXchgParts;
  if (S < 2) return;
  const int p = ((rounds + S - 1) / S) * sstep;  // slices per workgroup
  S = (R + p - 1) / p;
  if (S < 2) return;
  *split = S;
  *per = p;
}

template <int RT, int LAYOUT, bool AUG, bool DBG>
hipError_t launch(hipStream_t s, KArgs &a, int dev) {
  const int cus = num_cus();
  if constexpr (!DBG) {
    int S = 0, per = a.R;
    if (!a.aabb_only && a.tsdf) split_plan(a.n, a.R, cus, &S, &per);
    if (S >= 2) {
      a.split = S;
      a.per = per;
      a.queue = nullptr;
      a.xchg = a.n <= kXchgFrames ? xchg_for(dev, s, &a.seq) : nullptr;
      static const int polls = [] {
        const char *e = getenv("TSDF_XCHG_POLLS");
        return e ? atoi(e) : kXchgPolls;
      }();
      a.polls = polls;
      if (a.xchg) {
        hipLaunchKernelGGL((tsdf_split_kernel<RT, LAYOUT, AUG, true>), dim3(a.n * S), dim3(kWG), 0, s, a, a.depth,
                           a.offsets, a.headers, a.xforms);
      } else {
        hipLaunchKernelGGL((tsdf_split_kernel<RT, LAYOUT, AUG, false>), dim3(a.n * S), dim3(kWG), 0, s, a, a.depth,
                           a.offsets, a.headers, a.xforms);
      }
      return hipGetLastError();
    }
  }
  // persistent: one workgroup per CU; with fewer frames per CU than groups the later groups idle or help
  const int grid = a.n < cus ? a.n : cus;
  a.split = 0;
  a.per = a.R;
  auto fused = [&](auto groups_tag) {
    constexpr int G = decltype(groups_tag)::value;
    a.queue = a.n > grid * G ? queue_word(dev, s, &a.qepoch) : nullptr;  // no dynamic frames: no word needed
    hipLaunchKernelGGL((tsdf_fused_kernel<RT, LAYOUT, AUG, DBG, G>), dim3(grid), dim3(kWG), 0, s, a, a.depth, a.offsets,
                       a.headers, a.xforms);
  };
  if constexpr (RT != 0) {
    fused(std::integral_constant<int, groups_for(RT)>{});
  } else if constexpr (groups_for(32) == groups_for(64)) {
    fused(std::integral_constant<int, groups_for(32)>{});
  } else {   // any other resolution: the instantiation is generic in R, the group count follows the resolution
    if (groups_for(a.R) == groups_for(64)) fused(std::integral_constant<int, groups_for(64)>{});
    else fused(std::integral_constant<int, groups_for(32)>{});
  }
  return hipGetLastError();
}

// -DTSDF_DEV_ONLY64 / -DTSDF_DEV_ONLY32 (experiment builds, never the product): only the 64^3 (32^3) [c,z,y,x]
// instantiations are compiled — 25 s instead of 3 min per variant; every other call returns hipErrorInvalidValue.
template <int LAYOUT, bool AUG>
hipError_t launch_r(hipStream_t s, KArgs &a, int dev) {
#if defined(TSDF_DEV_ONLY64)
  if constexpr (LAYOUT == 0) {
    if (a.R == 64) return launch<64, LAYOUT, AUG, false>(s, a, dev);
  }
  return hipErrorInvalidValue;
#elif defined(TSDF_DEV_ONLY32)
  if constexpr (LAYOUT == 0 && !AUG) {
    if (a.R == 32) return launch<32, LAYOUT, AUG, false>(s, a, dev);
  }
  return hipErrorInvalidValue;
#else
  if (a.R == 32) return launch<32, LAYOUT, AUG, false>(s, a, dev);
  if (a.R == 64) return launch<64, LAYOUT, AUG, false>(s, a, dev);
  return launch<0, LAYOUT, AUG, false>(s, a, dev);
#endif
}

struct RunOpts {
  float *aabb = nullptr, *grid = nullptr, *ori = nullptr;
  int aabb_only = 0;
  const float *grid_in = nullptr;
  const double *xforms = nullptr;
  const tsdf_labels *labels = nullptr;
  int32_t *pixmap = nullptr;
  const int64_t *index = nullptr;  // indexed entry
  int64_t n_src = 0;
  const int64_t *h_index = nullptr;  // indexed entry, index in HOST memory, copied into the kernel arguments
};

int run(const float *d_depth, int64_t depth_len, const int64_t *d_offsets, const int32_t *d_headers, int n, int R,
        const tsdf_cam *cam, int layout, void *hip_stream, float *t, float *ml, float *mp, int32_t *st,
        const RunOpts &o) {
  // (argument checks: tsdf_host.inc, shared with the sanitizer build of the host code)
  const int chk = tsdf_host::check_run_args(d_depth, depth_len, d_offsets, d_headers, n, R, cam, layout, t, o.aabb_only,
                                            o.labels, tsdf_resolution_supported(R));
  if (chk == tsdf_host::kNothingToDo) return TSDF_OK;
  if (chk != TSDF_OK) return chk;
  if (!cam) cam = &kDefaultCam;
  int dev = 0;
  int rc = check_device(&dev);
  if (rc != TSDF_OK) return rc;
  KArgs a;
  memset(&a, 0, sizeof a);
  a.depth = d_depth;
  a.offsets = d_offsets;
  a.headers = d_headers;
  a.n = n;
  a.R = R;
  a.cam.focal = cam->focal;
  a.cam.cx = cam->cx;
  a.cam.cy = cam->cy;
  a.cam.inv_focal = 1.0 / cam->focal;
  a.cam.eps = cam->invalid_eps;
  a.cam.trunc_vox = cam->trunc_voxels;
  a.tsdf = t;
  a.max_l = ml;
  a.mid_p = mp;
  a.status = st;
  a.aabb = o.aabb;
  a.grid = o.grid;
  a.ori = o.ori;
  a.aabb_only = o.aabb_only;
  a.grid_in = o.grid_in;
  a.xforms = o.xforms;
  a.depth_len = depth_len;
  a.index = o.index;
  a.n_src = o.n_src;
  if (o.h_index) {
    if (n > TSDF_INLINE_INDEX_MAX) return TSDF_ERR_INVALID_ARG;
    a.n_inline = n;
    memcpy(a.inline_index, o.h_index, sizeof(int64_t) * (size_t)n);
  }
  if (o.labels) {
    a.gt = o.labels->d_gt;
    a.gt_nor = o.labels->d_out_gt_nor;
    a.gt_aug = o.labels->d_out_gt_aug;
    a.n_joints = o.labels->n_joints;
    a.clamp = o.labels->clamp;
  }
  a.pixmap = o.pixmap;
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  hipError_t e;
#if defined(TSDF_DEV_ONLY64) || defined(TSDF_DEV_ONLY32)
  if (o.pixmap) return TSDF_ERR_INVALID_ARG;
#endif
  if (o.pixmap) {
#if !defined(TSDF_DEV_ONLY64) && !defined(TSDF_DEV_ONLY32)
    if (layout == TSDF_LAYOUT_CZYX)
      e = R == 32 ? launch<32, 0, false, true>(s, a, dev) : launch<0, 0, false, true>(s, a, dev);
    else
      e = R == 32 ? launch<32, 1, false, true>(s, a, dev) : launch<0, 1, false, true>(s, a, dev);
#else
    e = hipErrorInvalidValue;
#endif
  } else if (o.xforms) {
    e = layout == TSDF_LAYOUT_CZYX ? launch_r<0, true>(s, a, dev) : launch_r<1, true>(s, a, dev);
  } else {
    e = layout == TSDF_LAYOUT_CZYX ? launch_r<0, false>(s, a, dev) : launch_r<1, false>(s, a, dev);
  }
  return e == hipSuccess ? TSDF_OK : TSDF_ERR_LAUNCH;
}

int run_normalize(const float *d_in, const float *d_max_l, const float *d_mid_p, int n, int n_joints, int clamp,
                  int inverse, void *hip_stream, float *d_out) {
  if (n < 0 || n_joints < 1 || n_joints > 170) return TSDF_ERR_INVALID_ARG;
  if (n == 0) return TSDF_OK;
  if (!d_in || !d_max_l || !d_mid_p || !d_out) return TSDF_ERR_INVALID_ARG;
  int dev = 0;
  int rc = check_device(&dev);
  if (rc != TSDF_OK) return rc;
  const int nc = 3 * n_joints;
  const int64_t total = (int64_t)n * nc;
  const int64_t blocks = (total + 255) / 256;
  if (blocks > 0x7fffffff) return TSDF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(tsdf_normalize_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                     d_in, d_max_l, d_mid_p, total, nc, clamp, inverse, d_out);
  return hipGetLastError() == hipSuccess ? TSDF_OK : TSDF_ERR_LAUNCH;
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
             d_out_mid_p, d_out_status, RunOpts{});
}

int tsdf_voxelize_labels_hip(const float *d_depth, int64_t depth_len, const int64_t *d_offsets, const int32_t *d_headers,
                             int n, int R, const tsdf_cam *cam, int layout, void *hip_stream, float *d_out_tsdf,
                             float *d_out_max_l, float *d_out_mid_p, int32_t *d_out_status, const tsdf_labels *labels) {
  if (n > 0 && (!d_out_tsdf || !d_out_max_l || !d_out_mid_p)) return TSDF_ERR_INVALID_ARG;
  if (!labels) return TSDF_ERR_INVALID_ARG;
  RunOpts o;
  o.labels = labels;
  return run(d_depth, depth_len, d_offsets, d_headers, n, R, cam, layout, hip_stream, d_out_tsdf, d_out_max_l,
             d_out_mid_p, d_out_status, o);
}

int tsdf_voxelize_indexed_hip(const float *d_depth, int64_t depth_len, const int64_t *d_offsets, const int32_t *d_headers,
                              int64_t n_pack, const int64_t *d_index, int n, int R, const tsdf_cam *cam, int layout,
                              void *hip_stream, float *d_out_tsdf, float *d_out_max_l, float *d_out_mid_p,
                              int32_t *d_out_status, const tsdf_labels *labels) {
  if (n > 0 && (!d_out_tsdf || !d_out_max_l || !d_out_mid_p || !d_index)) return TSDF_ERR_INVALID_ARG;
  if (n_pack < 0 || (n > 0 && n_pack == 0)) return TSDF_ERR_INVALID_ARG;
  RunOpts o;
  o.labels = labels;
  o.index = d_index;
  o.n_src = n_pack;
  return run(d_depth, depth_len, d_offsets, d_headers, n, R, cam, layout, hip_stream, d_out_tsdf, d_out_max_l,
             d_out_mid_p, d_out_status, o);
}

int tsdf_voxelize_indexed_host_hip(const float *d_depth, int64_t depth_len, const int64_t *d_offsets,
                                   const int32_t *d_headers, int64_t n_pack, const int64_t *h_index, int n, int R,
                                   const tsdf_cam *cam, int layout, void *hip_stream, float *d_out_tsdf, float *d_out_max_l,
                                   float *d_out_mid_p, int32_t *d_out_status, const tsdf_labels *labels) {
  if (n > 0 && (!d_out_tsdf || !d_out_max_l || !d_out_mid_p || !h_index)) return TSDF_ERR_INVALID_ARG;
  if (n_pack < 0 || (n > 0 && n_pack == 0) || n > TSDF_INLINE_INDEX_MAX) return TSDF_ERR_INVALID_ARG;
  RunOpts o;
  o.labels = labels;
  o.h_index = h_index;
  o.n_src = n_pack;
  return run(d_depth, depth_len, d_offsets, d_headers, n, R, cam, layout, hip_stream, d_out_tsdf, d_out_max_l,
             d_out_mid_p, d_out_status, o);
}

int tsdf_voxelize_indexed_aug_hip(const float *d_depth, int64_t depth_len, const int64_t *d_offsets,
                                  const int32_t *d_headers, int64_t n_pack, const int64_t *d_index, int n, int R,
                                  const tsdf_cam *cam, int layout, void *hip_stream, const double *d_xforms,
                                  float *d_out_tsdf, float *d_out_max_l, float *d_out_mid_p, int32_t *d_out_status,
                                  const tsdf_labels *labels) {
  if (n > 0 && (!d_out_tsdf || !d_out_max_l || !d_out_mid_p || !d_index || !d_xforms)) return TSDF_ERR_INVALID_ARG;
  if (n_pack < 0 || (n > 0 && n_pack == 0) || (reinterpret_cast<uintptr_t>(d_xforms) & 7)) return TSDF_ERR_INVALID_ARG;
  RunOpts o;
  o.labels = labels;
  o.index = d_index;
  o.n_src = n_pack;
  o.xforms = d_xforms;
  return run(d_depth, depth_len, d_offsets, d_headers, n, R, cam, layout, hip_stream, d_out_tsdf, d_out_max_l,
             d_out_mid_p, d_out_status, o);
}

int tsdf_voxelize_grid_hip(const float *d_depth, int64_t depth_len, const int64_t *d_offsets, const int32_t *d_headers, int n,
                           int R, const tsdf_cam *cam, int layout, void *hip_stream, const float *d_grid,
                           float *d_out_tsdf, int32_t *d_out_status) {
  if (n > 0 && (!d_out_tsdf || !d_grid)) return TSDF_ERR_INVALID_ARG;
  RunOpts o;
  o.grid_in = d_grid;
  return run(d_depth, depth_len, d_offsets, d_headers, n, R, cam, layout, hip_stream, d_out_tsdf, nullptr, nullptr,
             d_out_status, o);
}

int tsdf_voxelize_aug_hip(const float *d_depth, int64_t depth_len, const int64_t *d_offsets, const int32_t *d_headers, int n,
                          int R, const tsdf_cam *cam, int layout, void *hip_stream, const double *d_xforms,
                          float *d_out_tsdf, float *d_out_max_l, float *d_out_mid_p, int32_t *d_out_status) {
  if (n > 0 && (!d_out_tsdf || !d_out_max_l || !d_out_mid_p || !d_xforms)) return TSDF_ERR_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(d_xforms) & 7) return TSDF_ERR_INVALID_ARG;
  RunOpts o;
  o.xforms = d_xforms;
  return run(d_depth, depth_len, d_offsets, d_headers, n, R, cam, layout, hip_stream, d_out_tsdf, d_out_max_l,
             d_out_mid_p, d_out_status, o);
}

int tsdf_voxelize_aug_labels_hip(const float *d_depth, int64_t depth_len, const int64_t *d_offsets,
                                 const int32_t *d_headers, int n, int R, const tsdf_cam *cam, int layout,
                                 void *hip_stream, const double *d_xforms, float *d_out_tsdf, float *d_out_max_l,
                                 float *d_out_mid_p, int32_t *d_out_status, const tsdf_labels *labels) {
  if (n > 0 && (!d_out_tsdf || !d_out_max_l || !d_out_mid_p || !d_xforms)) return TSDF_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(d_xforms) & 7) || !labels) return TSDF_ERR_INVALID_ARG;
  RunOpts o;
  o.xforms = d_xforms;
  o.labels = labels;
  return run(d_depth, depth_len, d_offsets, d_headers, n, R, cam, layout, hip_stream, d_out_tsdf, d_out_max_l,
             d_out_mid_p, d_out_status, o);
}

int tsdf_aabb_hip(const float *d_depth, int64_t depth_len, const int64_t *d_offsets, const int32_t *d_headers, int n, int R,
                  const tsdf_cam *cam, void *hip_stream, float *d_out_aabb, float *d_out_grid,
                  float *d_out_ori, int32_t *d_out_status) {
  RunOpts o;
  o.aabb = d_out_aabb;
  o.grid = d_out_grid;
  o.ori = d_out_ori;
  o.aabb_only = 1;
  return run(d_depth, depth_len, d_offsets, d_headers, n, R, cam, TSDF_LAYOUT_CZYX, hip_stream, nullptr, nullptr,
             nullptr, d_out_status, o);
}

int tsdf_normalize_joints_hip(const float *d_gt, const float *d_max_l, const float *d_mid_p, int n, int n_joints,
                              int clamp, void *hip_stream, float *d_out_gt_nor) {
  return run_normalize(d_gt, d_max_l, d_mid_p, n, n_joints, clamp, 0, hip_stream, d_out_gt_nor);
}

int tsdf_denormalize_joints_hip(const float *d_pred, const float *d_max_l, const float *d_mid_p, int n, int n_joints,
                                void *hip_stream, float *d_out_joints) {
  return run_normalize(d_pred, d_max_l, d_mid_p, n, n_joints, 0, 1, hip_stream, d_out_joints);
}

int tsdf_debug_pixmap_hip(const float *d_depth, int64_t depth_len, const int64_t *d_offsets, const int32_t *d_headers,
                          int n, int R, const tsdf_cam *cam, int layout, void *hip_stream, const float *d_grid,
                          float *d_out_tsdf, int32_t *d_out_pixmap, int32_t *d_out_status) {
  if (n > 0 && (!d_out_tsdf || !d_out_pixmap)) return TSDF_ERR_INVALID_ARG;
  RunOpts o;
  o.grid_in = d_grid;
  o.pixmap = d_out_pixmap;
  return run(d_depth, depth_len, d_offsets, d_headers, n, R, cam, layout, hip_stream, d_out_tsdf, nullptr, nullptr,
             d_out_status, o);
}

int tsdf_stream_release(void *hip_stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return TSDF_OK;
  }
  (void)stream_slot(dev, static_cast<hipStream_t>(hip_stream), true);
  return TSDF_OK;
}

int tsdf_debug_set_queue_word(void *hip_stream, uint64_t value) {
  int dev = 0;
  const int rc = check_device(&dev);
  if (rc != TSDF_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const int i = stream_slot(dev, s, false);
  if (i < 0) return TSDF_ERR_INVALID_ARG;
  if (hipStreamSynchronize(s) != hipSuccess) return TSDF_ERR_LAUNCH;
  const unsigned long long v = value;
  if (hipMemcpy(g_slots[dev].base + i, &v, sizeof v, hipMemcpyHostToDevice) != hipSuccess) return TSDF_ERR_LAUNCH;
  return TSDF_OK;
}

int tsdf_describe_launch(int n, int R, int layout, int aug, char *buf, int buflen) {
  if (!buf || buflen < 1 || n < 1 || !tsdf_resolution_supported(R)) return TSDF_ERR_INVALID_ARG;
  if (layout != TSDF_LAYOUT_CZYX && layout != TSDF_LAYOUT_CXYZ) return TSDF_ERR_INVALID_ARG;
  int dev = 0;
  const int rc = check_device(&dev);
  if (rc != TSDF_OK) return rc;
  const int rt = (R == 32 || R == 64) ? R : 0;   // launch_r's choice of instantiation
  int S = 0, per = R;
  split_plan(n, R, num_cus(), &S, &per);
  if (S >= 2) {
    snprintf(buf, (size_t)buflen, "tsdf_split_kernel<%d, %d, %s, true> x%d", rt, layout, aug ? "true" : "false", S);
  } else {
    snprintf(buf, (size_t)buflen, "tsdf_fused_kernel<%d, %d, %s, false, %d>", rt, layout, aug ? "true" : "false",
             groups_for(R));
  }
  return TSDF_OK;
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
int tsdf_debug_read_wstamps(unsigned long long *host_out, int count) {
  const int total = kStampBlocks * kStampFrames * 16 * 2;
  if (count > total) count = total;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wstamps), sizeof(unsigned long long) * count) != hipSuccess)
    return -1;
  return count;
}
#endif

}  // extern "C"
