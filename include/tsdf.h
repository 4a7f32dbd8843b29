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
 *     and return WITHOUT synchronising.  They never print and never throw; they are
 *     re-entrant for distinct streams and may be captured into a hipGraph.  They
 *     allocate nothing, clear nothing and never synchronise — the little device state
 *     the library owns (one work-queue word per stream; 8 MiB of mailboxes through which
 *     the workgroups that share a frame of a small batch, n <= CUs/2, exchange partial
 *     extents: 128 KiB per stream for the first 64 streams) is a zero-initialised
 *     device global placed when the library is loaded.  Captured launches and streams
 *     beyond the 64th never use the mailboxes.  Work-queue ownership (the kernels hand frames to their
 *     workgroups dynamically): an eager launch uses a device-side queue word owned by
 *     its (device, stream) pair — launches on one stream execute in order, so the word
 *     is never shared, however many launches are in flight; the library keeps one word
 *     per stream it has seen (tsdf_stream_release returns it).  The word carries the number
 *     of the launch that may draw from it: whatever state an earlier launch left it in — one
 *     that never finished included — a launch re-initialises a word that is not its own, so
 *     no state of the word can cost a frame (tsdf_debug_set_queue_word proves it in the tests).
 *     STREAM OWNERSHIP: the word (and the
 *     mailboxes of small batches) are keyed by the hipStream_t value, so a stream must not be destroyed — and
 *     tsdf_stream_release must not be called for it — while one of its voxelizer launches is still in flight:
 *     hipStreamDestroy returns at once and lets the work drain, and a new stream that is handed the same
 *     value would share the word with the launch still draining (two launches drawing tickets from one word
 *     skip frames silently).  Synchronise the stream first; destroying or releasing a busy stream is
 *     undefined.  A launch issued while
 *     its stream is being captured, on hipStreamPerThread from more threads than there
 *     are words, or on a stream beyond the 1024th distinct one uses no global state at
 *     all: its workgroups then share frames through a counter in their own LDS (static
 *     split across CUs, dynamic inside a CU).  A captured launch is therefore
 *     self-contained: it may be replayed on any stream, next to eager launches, next to
 *     other graphs and next to itself (outputs permitting).
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

#define TSDF_ABI_VERSION 7

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
  TSDF_FRAME_DEGENERATE = 1, /* no valid pixel, or an AABB of zero or non-finite extent / non-finite centre
                                (a +-inf depth is "valid" by the |d| >= eps rule): zero volume, max_l = 0;
                                mid_p as computed when finite (zero extent), else 0.
                                The reference prints and returns None here (tsdf_numba.py:162-171). */
  TSDF_FRAME_BAD_HEADER = 2  /* right<=left, bottom<=top, right-left or bottom-top overflowing int32, bbox area !=
                                offsets[i+1]-offsets[i], or the payload not inside [0, depth_len): zero volume,
                                max_l = 0, mid_p = 0; depth is not read.                            */
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
  float invalid_eps;   /* 1.0: a pixel is valid iff |depth| >= invalid_eps (:40, :87).  A NaN depth is INVALID
                          here (the comparison is false); in the reference abs(NaN) < 1 is false too, which
                          makes NaN "valid" there and poisons the frame — a deliberate deviation.           */
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
 *                                   d_offsets, d_headers and tsdf_labels.d_gt — the per-frame metadata, 32..300 bytes
 *                                   a frame, each word read once — may also be DEVICE-ACCESSIBLE PAGE-LOCKED HOST
 *                                   memory (hipHostMalloc): the kernel then fetches them over the link, which costs
 *                                   a few us per launch and saves the caller three small copies per batch (a
 *                                   streaming loader's depth upload keeps the link's rate only when no small copy
 *                                   sits between two big ones: tools/exp_loader_meta.py).  The memory must stay
 *                                   unchanged until the launch has finished.  Everything else is device memory.
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
 * What the contract is pinned to: the reference's runnable implementation of the same formula, the CPU loop
 * (pre/tsdf_for.py / pre/process.py), evaluated on float64-typed parameters (tests/golden).  Parity with the
 * numba kernel AS COMPILED is unpinned: numba-CUDA goes through NVVM, whose default contracts a*b+c into an
 * fma (which would change pix = int(v*q + c) in rare voxels), it treats NaN depth as valid, and it cannot be
 * run here (no numba, no params.py, no CUDA).
 * Known sub-ulp deviations at the formula's discontinuities (measure ~0, effect O(1) where hit): the three
 * divisions by trunc_dis are multiplications by its float64 reciprocal, and dist > 1 (:54) is tested as
 * dist^2 <= 1 without the square root — these differ from tsdf_numba.py:47-54 only for dist in (1, 1+2^-52].
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
 *   The AABB / grid placement is that of the mapped cloud T(p) over all valid pixels (T evaluated as
 *   fma(A_i0, x, fma(A_i1, y, fma(A_i2, z, b_i))) per row, then rounded to float32); a voxel centre v' of that
 *   grid is mapped back (T^-1 with separately rounded products and sums grouped as (A_i0 x + A_i1 y) +
 *   (A_i2 z + b_i)), projected and gathered as in tsdf_voxelize_hip, and the truncated distances are those
 *   between v' and T(w), w the surface point of the gathered pixel.  ABI v5 states them in the cheapest exact
 *   form an affine T allows (v4 formed w and T(w) explicitly, about twice the float64 work per voxel, when round 2
 *   had measured the kernel bound by instruction issue; since round 4 it is ~9/10 memory system): with dxi = pix_x - cx, dyi = pix_y - cy, iF = 1/F, it = 1/trunc_dis, float64, one rounding
 *   per operation,
 *       g_i0 = -(A_i0 * iF),  g_i1 = A_i1 * iF                          per frame
 *       c_i  = fma(g_i0, dxi, fma(g_i1, dyi, A_i2))
 *       u_i  = fma(pd, c_i, v'_i - b_i)                                 = v'_i - T(w)_i in mm
 *       t_i  = u_i * it ;  near iff fma(t_z, t_z, fma(t_y, t_y, t_x * t_x)) <= 1
 *       value_i = near ? min(|t_i|, 1) : 1, negated iff u_z < 0         (pre/tsdf_numba.py:47-68)
 *   With the identity map the grid, the pixel every voxel gathers, the zero mask, the sign and the z component
 *   equal tsdf_voxelize_hip bit for bit and x / y agree to the float32 rounding (<= 1e-5 by a wide margin).
 *   max_l / mid_p are in the mapped frame.
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

/* ---- ABI v3 additions ---------------------------------------------------------------------------- */

/*
 * Label normalisation fused into the voxelizer — replaces pre/joint_nor.py:8-18 and the per-sample Python
 * loop of 3D_CNN/train.py:236-244: joint_nor = (joint - mid_p) / max_l + 0.5 (float32, three separately
 * rounded operations), then, with clamp != 0, joint_nor < 0 -> 0 and > 1 -> 1 (train.py:241-242; NaN stays).
 * The lanes that hold the frame's mid_p / max_l write it, so the labels cost no extra launch.
 * A frame whose status is not TSDF_FRAME_OK (max_l == 0; the reference returns None for it) gets 0.5 for
 * every coordinate — the centre of the cube — instead of a division by zero.
 */
typedef struct tsdf_labels {
  const float *d_gt;   /* float32[n][3*n_joints]: x,y,z per joint, camera frame, mm (rows of joint.txt,
                          pre/read_MSRA.py:143-152)                                                         */
  int n_joints;        /* 21 for MSRA; 1..170                                                               */
  int clamp;           /* 1: clamp to [0,1] (3D_CNN/train.py:241-242); 0: pre/joint_nor.py as written       */
  float *d_out_gt_nor; /* float32[n][3*n_joints]                                                            */
  float *d_out_gt_aug; /* may be NULL: T(joint) in mm — the joints in the frame of the grid.  With
                          tsdf_voxelize_aug_labels_hip they are mapped with the frame's forward map (as
                          pre/process.py:232-249 maps them with the cloud's S and R) and the normalised labels
                          are those of T(joint) in the augmented grid; with tsdf_voxelize_labels_hip T is the
                          identity, i.e. this is a device copy of d_gt (for callers whose d_gt is host memory) */
} tsdf_labels;

/* tsdf_voxelize_hip + labels.  `labels` is a HOST struct of device pointers, read during the call. */
int tsdf_voxelize_labels_hip(const float *d_depth, int64_t depth_len, const int64_t *d_offsets,
                             const int32_t *d_headers, int n, int R, const tsdf_cam *cam, int layout,
                             void *hip_stream, float *d_out_tsdf, float *d_out_max_l, float *d_out_mid_p,
                             int32_t *d_out_status, const tsdf_labels *labels);

/* tsdf_voxelize_aug_hip + labels (joints are mapped with the frame's forward map first). */
int tsdf_voxelize_aug_labels_hip(const float *d_depth, int64_t depth_len, const int64_t *d_offsets,
                                 const int32_t *d_headers, int n, int R, const tsdf_cam *cam, int layout,
                                 void *hip_stream, const double *d_xforms, float *d_out_tsdf, float *d_out_max_l,
                                 float *d_out_mid_p, int32_t *d_out_status, const tsdf_labels *labels);

/* ---- ABI v4 addition ----------------------------------------------------------------------------- */

/*
 * Batches drawn from a pack that is RESIDENT on the device — what replaces "load every npz into RAM" of
 * 3D_CNN/dataset.py:35,99-117 when the whole dataset fits the GPU (all of MSRA: 76.5 k crops = 4.8 GB of 288 GB):
 * d_depth / d_offsets[n_pack+1] / d_headers[n_pack][6] (and labels->d_gt[n_pack][3*n_joints]) describe the pack, uploaded
 * once; frame i of the batch is pack frame d_index[i] (any order, repeats allowed: a shuffled minibatch), so a training step
 * uploads n indices instead of n crops (d_index may be page-locked host memory like the other metadata).  Outputs are
 * indexed by batch position.  An index outside [0, n_pack) gives that frame TSDF_FRAME_BAD_HEADER.  labels may be NULL.
 * Same kernels, same arithmetic: the result equals tsdf_voxelize_labels_hip on the gathered frames bit for bit.
 */
int tsdf_voxelize_indexed_hip(const float *d_depth, int64_t depth_len, const int64_t *d_offsets,
                              const int32_t *d_headers, int64_t n_pack, const int64_t *d_index, int n, int R,
                              const tsdf_cam *cam, int layout, void *hip_stream, float *d_out_tsdf, float *d_out_max_l,
                              float *d_out_mid_p, int32_t *d_out_status, const tsdf_labels *labels);

/* The same with the 3-D augmentation of tsdf_voxelize_aug_hip: d_xforms float64[n][24] holds one map per BATCH position
 * (device memory, or — like the index — page-locked host memory: a frame's 24 words are read a few times per workgroup, so a
 * loader that draws its batches by index can keep the maps next to the indices and upload nothing); labels (may be NULL)
 * are those of the pack, mapped and normalised as by tsdf_voxelize_aug_labels_hip. */
int tsdf_voxelize_indexed_aug_hip(const float *d_depth, int64_t depth_len, const int64_t *d_offsets,
                                  const int32_t *d_headers, int64_t n_pack, const int64_t *d_index, int n, int R,
                                  const tsdf_cam *cam, int layout, void *hip_stream, const double *d_xforms,
                                  float *d_out_tsdf, float *d_out_max_l, float *d_out_mid_p, int32_t *d_out_status,
                                  const tsdf_labels *labels);

/* Host-side helper for loaders that cannot keep the pack on the device (no GPU involved, callable without one):
 * frames index[0..n) of a packed HOST buffer (src, src_offsets[n_src+1]) copied back to back into dst — e.g. a page-locked
 * staging buffer of dst_capacity elements — by `threads` workers; dst_offsets[n+1] receives the new offsets.  Replaces
 * the per-file reads a shuffled batch costs the reference's loader (3D_CNN/dataset.py:99-117). */
int tsdf_host_gather_frames(const float *src, const int64_t *src_offsets, int64_t n_src, const int64_t *index, int64_t n,
                            float *dst, int64_t dst_capacity, int64_t *dst_offsets, int threads);

/* ---- ABI v5 additions ---------------------------------------------------------------------------- */

/* tsdf_voxelize_indexed_hip for SMALL batches whose index is in ordinary HOST memory: h_index[0..n), n <=
 * TSDF_INLINE_INDEX_MAX, is read DURING the call and travels to the GPU inside the kernel arguments — no upload, no
 * page-locked buffer whose lifetime the caller has to manage, and no read over the link on the launch's critical path (a
 * kernel that fetches its 16 indices from page-locked host memory starts ~2 us later: at the reference's batch size, 16 —
 * 3D_CNN/train.py:36 — that is a tenth of the launch).  What a DataLoader-driven training step calls once per batch.
 * Everything else as tsdf_voxelize_indexed_hip. */
#define TSDF_INLINE_INDEX_MAX 32
int tsdf_voxelize_indexed_host_hip(const float *d_depth, int64_t depth_len, const int64_t *d_offsets,
                                   const int32_t *d_headers, int64_t n_pack, const int64_t *h_index, int n, int R,
                                   const tsdf_cam *cam, int layout, void *hip_stream, float *d_out_tsdf, float *d_out_max_l,
                                   float *d_out_mid_p, int32_t *d_out_status, const tsdf_labels *labels);

/* tsdf_host_gather_frames with the LENGTH of the source buffer (src_len elements): a frame whose
 * [src_offsets[f], src_offsets[f+1]) does not lie inside [0, src_len) — a damaged pack file — makes the call return
 * TSDF_ERR_INVALID_ARG before anything is copied, instead of reading outside the mapping.  The v4 entry above trusts
 * src_offsets (kept for v4 callers; it now applies the same checks with src_len = "unknown": negative offsets and
 * decreasing pairs are still refused).  Neither ever throws: if worker threads cannot be started the copy runs on the
 * calling thread. */
int tsdf_host_gather_frames_n(const float *src, int64_t src_len, const int64_t *src_offsets, int64_t n_src,
                              const int64_t *index, int64_t n, float *dst, int64_t dst_capacity, int64_t *dst_offsets,
                              int threads);

/* The normalisation on its own, from max_l / mid_p already on the device (pre/joint_nor.py:8-18), and its
 * inverse for predictions, (pred - 0.5) * max_l + mid_p (3D_CNN/train.py:263-266).  Frames with max_l == 0:
 * 0.5 / mid_p respectively. */
int tsdf_normalize_joints_hip(const float *d_gt, const float *d_max_l, const float *d_mid_p, int n, int n_joints,
                              int clamp, void *hip_stream, float *d_out_gt_nor);
int tsdf_denormalize_joints_hip(const float *d_pred, const float *d_max_l, const float *d_mid_p, int n,
                                int n_joints, void *hip_stream, float *d_out_joints);

/* Forget the work-queue word kept for `hip_stream` on the current device (call it when destroying a stream that
 * has no voxelizer launch in flight; optional — an unknown stream is not an error).  Returns TSDF_OK. */
int tsdf_stream_release(void *hip_stream);

/* ---- ABI v6 (adds only) ----
 * Which kernel a call would launch on the current device: writes the instantiation's name — e.g.
 * "tsdf_fused_kernel<32, 0, false, false, 2>" or "tsdf_split_kernel<32, 0, false, x8" (v7) — into buf (at most
 * buflen bytes, NUL-terminated) for a batch of n frames at resolution R in the given layout, plain (aug = 0) or
 * augmented (aug != 0).  Launches nothing.  bench.py names the kernel of its roofline with it.  Returns TSDF_OK,
 * TSDF_ERR_INVALID_ARG or TSDF_ERR_NO_DEVICE. */
int tsdf_describe_launch(int n, int R, int layout, int aug, char *buf, int buflen);

/* ---- ABI v7 ----
 * Removes tsdf_debug_pixmap_hip and tsdf_debug_set_queue_word from the product library: test hooks do not belong in a
 * shipping ABI (one of them synchronised a stream, against this header's own "entries never synchronise").  They are
 * declared in include/tsdf_debug.h and exist only in the debug build of the same sources (make -C .../csrc debug ->
 * build/libtsdf_hip_debug.so, -DTSDF_DEBUG_HOOKS), which the test-suite loads for the tests that need them.  Nothing else
 * changed: a v6 caller that used neither hook runs unmodified.
 * tsdf_describe_launch names a split launch by family only ("tsdf_split_kernel<32, 0, false, x8"): whether the exchange
 * or the redundant form runs depends on the stream the launch is issued on. */

#ifdef __cplusplus
}
#endif
#endif /* TSDF_H_ */
