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
//   * staging: that rectangle (the only pixels phase 2 can ever use — everything outside it is rejected by
//     pre/tsdf_numba.py:36 or :40) is copied into the LDS pool by LDS-DMA (global_load_lds_dwordx4), so the
//     per-voxel gather is an LDS read: no vector-memory latency, and stores never block loads; a rectangle larger
//     than the pool gathers from global memory (L2);
//   * phase 2: pix_x depends on (x,z) only and pix_y on (y,z) only -> both tabulated per frame in LDS
//     (true division for q = -F/v_z, unfused multiply-add, v_cvt_i32_f64 truncation); the y table holds
//     the row's pool entry directly.  Each lane owns 4 consecutive voxels along the layout's fastest
//     axis, so every wave store is 1 KiB contiguous (global_store_dwordx4 nt: written once, never
//     re-read).  The per-voxel chain uses reciprocals (<= a few ulp64 from the divisions it replaces —
//     10 orders of magnitude inside the 1e-5 parity bound); a wave whose 256 voxels are all rejected or
//     farther than the truncation distance along z skips the x/y terms (the result is then
//     (+-1,+-1,+-1) or 0 by pre/tsdf_numba.py:54-57);
//   * small batches (n <= CUs/2) take the split kernel instead: S workgroups per frame, each streams one
//     band of the frame's rows, the bands' partial extents meet in per-stream mailboxes (bounded wait; a
//     workgroup that does not hear from its siblings streams the whole frame itself), and each workgroup
//     voxelizes 1/S of the slow axis — bit-identical results;
//   * the augmented form (template AUG) maps every valid pixel / voxel centre / surface point through a
//     per-frame affine transform instead (tsdf_voxelize_aug_hip, re-specified: see include/tsdf.h).
// HBM-bound streaming read + streaming write.  No MFMA (gather/scatter, not a contraction), no
// inter-workgroup communication in the fused kernel (XCD placement is irrelevant), no CPU fallback, gfx950 only.
//
// One translation unit; the pieces (all inside the anonymous namespace below, in this order):
//   common.inc   constants, types, VALU / DPP / float64 helpers
//   phase1.inc   row stream -> extents (AABB) -> grid placement
//   phase2.inc   the voxel pass, plain and augmented; gather sources, projection tables, volume stores
//   frame.inc    kernel arguments, LDS layout, group barrier, per-frame helpers, LDS-DMA staging, table fill
//   queue.inc    per-(device, stream) work-queue words and exchange mailboxes, device and host side
//   kernels.inc  tsdf_fused_kernel, tsdf_split_kernel, tsdf_normalize_kernel
//   launch.inc   host side of a call: device check, split plan, instantiation choice, argument marshalling
//   abi.inc      extern "C" — include/tsdf.h (and, under -DTSDF_DEBUG_HOOKS, include/tsdf_debug.h)
//   tsdf_host.inc  host-only helpers, also compiled alone under the CPU sanitizers
// Build-time switches: TSDF_DEBUG_HOOKS (debug build: test hooks), TSDF_STAMPS (diagnostic build: in-kernel timeline),
// TSDF_STORE_ASM (cache-policy bits of the volume store, for A/B builds).  Knobs of earlier rounds whose other setting
// was "measured, not adopted" are parked as a diff: tools/patches/r05_removed_knobs.diff.

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
#ifdef TSDF_DEBUG_HOOKS
#include "../../include/tsdf_debug.h"
#endif
#include "tsdf_host.inc"   // the host-only part (also compiled alone, under sanitizers)


namespace {

#include "common.inc"    // constants, types, VALU / DPP / float64 helpers
#include "phase1.inc"    // row stream -> extents -> grid placement
#include "phase2.inc"    // the voxel pass, plain and augmented
#include "frame.inc"     // kernel arguments, LDS layout, per-frame helpers, staging, tables
#include "queue.inc"     // work-queue words and exchange mailboxes (device + host side)
#include "kernels.inc"   // tsdf_fused_kernel, tsdf_split_kernel, tsdf_normalize_kernel
#include "launch.inc"    // host side of a call

}  // namespace

#include "abi.inc"       // extern "C": include/tsdf.h
