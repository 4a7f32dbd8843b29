/*
 * tsdf.h — C ABI of the MI355X-native projective-TSDF voxelizer (libtsdf_hip.so).
 *
 * This is the drop-in boundary for the reference's voxelization hot path.  The
 * reference (Moon0shang/HandPoseEstimation-with-3D-CNNs) has no FFI of its own:
 * its boundary is the Python call `cal_tsdf_cuda(s)` (pre/tsdf_numba.py:119-161),
 * which runs two numba-CUDA kernels per frame with four synchronous PCIe copies.
 * The entry points below are what a ctypes binding for that call binds instead
 * (INTEGRATION.md shows the stub).  Plain pointers and sizes only; no C++ or
 * torch types cross this boundary.
 *
 * Conventions
 *   - The caller allocates and owns every buffer.  `d_` pointers are DEVICE
 *     pointers (hipMalloc / torch CUDA tensors) valid on the current device.
 *   - Calls enqueue work on `hip_stream` (a hipStream_t, NULL = default stream)
 *     and return WITHOUT synchronising.  They allocate nothing, never print and
 *     never throw; they are re-entrant for distinct streams and may be captured
 *     into a hipGraph.  (Each launch takes one of 1024 device-side work-queue slots,
 *     chosen round-robin on the host and left clean by the launch itself: up to 1024
 *     launches may be in flight at once, and one captured launch must not be
 *     replayed concurrently with itself.)
 *   - Return value: TSDF_OK (0) or a negative tsdf_status.  Per-frame data
 *     conditions (degenerate / malformed frames) never fail the call; they are
 *     reported through `d_out_status` and produce an all-zero volume.
 *   - There is NO CPU fallback in this library.  If no HIP device is usable the
 *     entry points return TSDF_ERR_NO_DEVICE.
 */
#ifndef TSDF_H_
#define TSDF_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSDF_ABI_VERSION 2

/* Output volume layouts.  Both hold float32[n][3][R][R][R]; channel c = x,y,z component. */
enum tsdf_layout {
  TSDF_LAYOUT_CZYX = 0, /* o[c][z][y][x], x fastest — numba kernel, pre/tsdf_numba.py:70-72 */
  TSDF_LAYOUT_CXYZ = 1  /* o[c][x][y][z], z fastest — CPU loop,     pre/tsdf_for.py:118-120  */
};

enum tsdf_status {
  TSDF_OK = 0,
  TSDF_ERR_INVALID_ARG = -1, /* NULL pointer, n < 0, unsupported R or layout                     */
  TSDF_ERR_NO_DEVICE = -2,   /* no usable HIP device / wrong architecture (library is gfx950)    */
  TSDF_ERR_LAUNCH = -3       /* the HIP runtime rejected the launch (see hipGetLastError)        */
};

/* Per-frame status words written to d_out_status (int32). */
enum tsdf_frame_status {
  TSDF_FRAME_OK = 0,
  TSDF_FRAME_DEGENERATE = 1, /* no valid pixel, or AABB of zero extent: zero volume, max_l = 0.
                                The reference prints and returns None here (tsdf_numba.py:162-171). */
  TSDF_FRAME_BAD_HEADER = 2  /* right<=left, bottom<=top, bbox area != offsets[i+1]-offsets[i], or the
                                payload not inside [0, depth_len): zero volume, max_l = 0, mid_p = 0;
                                depth is not read.                                                  */
};

/*
 * Camera / rule constants.  The defaults are the reference's module constants
 * (pre/tsdf_numba.py:8-10): FOCAL = 241.42 is a Python float, i.e. a float64
 * constant, and the principal point is the integer pair (160, 120); they are
 * doubles here so that the arithmetic contract of SURVEY.md Appendix A holds.
 */
typedef struct tsdf_cam {
  double focal;        /* 241.42                                                        */
  double cx;           /* 160                                                           */
  double cy;           /* 120                                                           */
  float invalid_eps;   /* 1.0: a pixel is valid iff |depth| >= invalid_eps (:40, :87)   */
  float trunc_voxels;  /* 3.0: truncation distance in voxel lengths (:146)              */
} tsdf_cam;

/* Fills *cam with the MSRA defaults above.  Replaces pre/tsdf_numba.py:8-12. */
void tsdf_default_cam(tsdf_cam *cam);

/* TSDF_ABI_VERSION of the loaded library. */
int tsdf_version(void);

/* Static, human-readable text for a tsdf_status (never NULL). */
const char *tsdf_strerror(int status);

/* 1 if R is a grid resolution the kernels accept (multiple of 4, 4..128), else 0. */
int tsdf_resolution_supported(int R);

/*
 * Batched voxelization — replaces cal_tsdf_cuda (pre/tsdf_numba.py:119-161), i.e.
 * min_max_kernel (:75-116) + host glue (:140-147) + tsdf_kernel (:15-72), for n
 * frames in ONE fused launch with no host round trip.
 *
 *   d_depth    float32[offsets[n]]  bbox crops packed back to back; frame i is
 *                                   d_depth[offsets[i] .. offsets[i+1]), row-major
 *                                   over its bounding box, millimetres, 0 = no hand
 *                                   (the payload of an MSRA .bin, pre/read_MSRA.py:155-164).
 *   depth_len  number of float32 elements in d_depth.  A frame whose [offsets[i], offsets[i+1]) does
 *                                   not lie inside it is treated as TSDF_FRAME_BAD_HEADER and never read.
 *   d_offsets  int64[n+1]           element offsets into d_depth, non-decreasing.
 *   d_headers  int32[n][6]          W, H, left, top, right, bottom (the .bin header).
 *   n          number of frames (0 is allowed and is a no-op).
 *   R          grid resolution (reference: 32).
 *   cam        constants, or NULL for the MSRA defaults.
 *   layout     enum tsdf_layout.
 *   hip_stream hipStream_t to enqueue on.
 *   d_out_tsdf   float32[n][3][R][R][R], 16-byte aligned (required); 256-byte alignment recommended: a wave
 *                                   stores 1 KiB runs, and a base that splits them across 256-byte lines
 *                                   measured 12 % (full frames) to 39 % (crops) slower (tools/exp_align.py)
 *   d_out_max_l  float32[n]      edge length of the cubic grid (mm)     (tsdf_numba.py:144,161)
 *   d_out_mid_p  float32[n][3]   centre of the grid (camera frame, mm)  (tsdf_numba.py:142,161)
 *   d_out_status int32[n] or NULL: enum tsdf_frame_status per frame.
 *
 * Arithmetic follows the numba kernels' inferred types:
 * float32 parameters, float64 intermediates, unfused multiply-add for the pixel
 * index, float32 store.  Results match that contract to <= 1e-5 absolute.
 */
int tsdf_voxelize_hip(const float *d_depth, int64_t depth_len, const int64_t *d_offsets, const int32_t *d_headers,
                      int n, int R, const tsdf_cam *cam, int layout, void *hip_stream,
                      float *d_out_tsdf, float *d_out_max_l, float *d_out_mid_p,
                      int32_t *d_out_status);

/*
 * Phase 2 with a caller-supplied grid placement — replaces tsdf_cal(data, vox_ori, voxel_len,
 * truncation) (pre/tsdf_for.py:44-122 == DataProcess.tsdf_cal, pre/process.py:122-200) and
 * tsdf_kernel launched on its own (pre/tsdf_numba.py:155-156).  This is what a drop-in for
 * tsdf_f(data, point_cloud) needs: the reference derives the placement from the point cloud it is
 * handed (pre/tsdf_for.py:9-16), not from the depth image.
 *
 *   d_grid  float32[n][8]  vox_ori[3], voxel_len, trunc_dis, then 3 pad words, per frame.
 * Frames without a valid pixel, or with trunc_dis <= 0, give a zero volume (status 1).
 */
int tsdf_voxelize_grid_hip(const float *d_depth, int64_t depth_len, const int64_t *d_offsets, const int32_t *d_headers,
                           int n, int R, const tsdf_cam *cam, int layout, void *hip_stream,
                           const float *d_grid, float *d_out_tsdf, int32_t *d_out_status);

/*
 * Augmented voxelization — BASELINE configs[4] (64^3 grid with 3-D augmentation fused into the TSDF
 * kernel).  Replaces what DataProcess.data_aug + tsdf_f were meant to do (pre/process.py:19-24,202-261)
 * but could not: data_aug raises AxisError on its own input, and where a variant of it runs
 * (cut_version/pre/process.py:69-134,207) the augmented cloud only moves the grid while the TSDF still
 * samples the un-augmented depth image.  This entry is therefore a RE-SPECIFICATION,
 * parity unpinned; its arithmetic contract is oracle/tsdf_oracle.c::tsdf_oracle_voxels_aug:
 *
 *   d_xforms  float64[n][24]  per frame: the forward affine map T(p) = A p + b as three rows
 *                             {A_i0, A_i1, A_i2, b_i}, then its inverse in the same form.
 *   The AABB / grid placement is that of the mapped cloud T(p) over all valid pixels; a voxel centre v'
 *   of that grid is mapped back (T^-1), projected and gathered as in tsdf_voxelize_hip, the surface
 *   point is mapped forward and the truncated distances are taken between v' and T(w).
 *   T is evaluated as fma(A_i0, x, fma(A_i1, y, fma(A_i2, z, b_i))) per row, T^-1 with separately rounded
 *   products summed left to right (float64 both).
 *   With the identity map the result equals tsdf_voxelize_hip.  max_l / mid_p are in the mapped frame.
 */
int tsdf_voxelize_aug_hip(const float *d_depth, int64_t depth_len, const int64_t *d_offsets, const int32_t *d_headers,
                          int n, int R, const tsdf_cam *cam, int layout, void *hip_stream,
                          const double *d_xforms, float *d_out_tsdf, float *d_out_max_l,
                          float *d_out_mid_p, int32_t *d_out_status);

/*
 * Phase 1 + glue only: the per-frame axis-aligned bounding box of all valid
 * back-projected pixels and the grid placement derived from it.  Replaces
 * min_max_kernel + host glue (pre/tsdf_numba.py:75-116,135-147) on their own.
 *
 *   d_out_aabb  float32[n][6]  min x,y,z then max x,y,z (the reference's mm_p row, :110-116)
 *   d_out_grid  float32[n][8]  mid_p[3], max_l, voxel_len, trunc_dis, then 2 pad words
 *   d_out_ori   float32[n][3]  vox_ori (:147)
 * Any of the three may be NULL.
 */
int tsdf_aabb_hip(const float *d_depth, int64_t depth_len, const int64_t *d_offsets, const int32_t *d_headers, int n,
                  int R, const tsdf_cam *cam, void *hip_stream, float *d_out_aabb,
                  float *d_out_grid, float *d_out_ori, int32_t *d_out_status);

#ifdef __cplusplus
}
#endif
#endif /* TSDF_H_ */
