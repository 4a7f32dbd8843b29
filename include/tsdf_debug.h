/*
 * tsdf_debug.h — test hooks of the TSDF voxelizer.  NOT part of the product ABI (include/tsdf.h): these two entries
 * exist only in the debug build of the library (make -C handposeestimation-with-3d-cnns_amd/csrc debug ->
 * build/libtsdf_hip_debug.so, the same sources with -DTSDF_DEBUG_HOOKS), which exports everything tsdf.h declares as
 * well.  The test-suite loads it for the pixel-map, queue-word and exchange-fallback tests (the debug build also reads
 * TSDF_XCHG_POLLS from the environment; the product reads no environment variable).
 */
#ifndef TSDF_DEBUG_H_
#define TSDF_DEBUG_H_

#include "tsdf.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Diagnostic: the voxelizer with its pixel map.  Same kernel code path as tsdf_voxelize_hip /
 * tsdf_voxelize_grid_hip (d_grid NULL / non-NULL) — projection tables, LDS-DMA staging, gather — with one
 * extra store per voxel:  d_out_pixmap int32[n][R][R][R], indexed [z][y][x] whatever the layout, holds the
 * gathered element index (pix_y - top) * b_w + pix_x - left (pre/tsdf_numba.py:38), -1 when the voxel
 * projects outside the bounding box (:36-37), -2 - index when the pixel there is invalid (:40-41).
 * Tests compare it exactly with the oracle's map.  Slower than the production entry (the staged image is the
 * whole bounding box instead of the rectangle of valid pixels, so more frames gather from global memory).
 */
int tsdf_debug_pixmap_hip(const float *d_depth, int64_t depth_len, const int64_t *d_offsets, const int32_t *d_headers,
                          int n, int R, const tsdf_cam *cam, int layout, void *hip_stream, const float *d_grid,
                          float *d_out_tsdf, int32_t *d_out_pixmap, int32_t *d_out_status);

/* Diagnostic: overwrite the work-queue word kept for `hip_stream` on the current device with `value` (synchronous;
 * the stream must be idle).  A launch re-initialises a word that does not carry its own epoch (csrc/queue.inc:
 * queue_ticket), so no word left behind by an EARLIER launch — finished, or dead mid-flight — can cost a frame, and this
 * is how tests/test_parity_gpu.py proves it: it poisons the word and checks that the next launch still voxelizes every
 * frame (the reference's loop processes every frame of a gesture, pre/read_MSRA.py:98-106).  Known limit, recorded by the
 * same test: a word that already carries the NEXT launch's epoch (host-predictable: 1, 2, 3, ... per stream) with a
 * non-zero count k makes that launch skip k queue frames — nothing but this hook can write such a word.  Returns TSDF_OK, or
 * TSDF_ERR_INVALID_ARG when the stream has no word (more than 1024 live streams, stream capture). */
int tsdf_debug_set_queue_word(void *hip_stream, uint64_t value);

#ifdef __cplusplus
}
#endif
#endif /* TSDF_DEBUG_H_ */
