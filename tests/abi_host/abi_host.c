/*
 * abi_host.c — the C ABI of include/tsdf.h driven from plain C: hipMalloc'd buffers, a HIP stream, no Python and
 * no torch anywhere in the process.  Test program (tests/test_parity_gpu.py::test_plain_c_host_program builds and
 * runs it on the GPU box); it links the oracle as the checker, which only tests may do.
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ abi_host.c -I/opt/rocm/include -I../../include -L<pkg> -ltsdf_hip \
 *       -L../../oracle -ltsdf_oracle -L/opt/rocm/lib -lamdhip64 -lm -o abi_host
 *
 * Frames: synthetic "hand" blobs generated here (seeded LCG), three crops of different sizes.
 * Checks: return codes, per-frame status, max_l / mid_p bit-exact, volume <= 1e-5 against the oracle, both
 * layouts, tsdf_aabb_hip, argument validation, n = 0; ABI v3: labels fused into the launch (tsdf_labels), the
 * stand-alone normalisation and its inverse, the pixel-map diagnostic (exact), a launch on hipStreamPerThread,
 * tsdf_stream_release.  Prints one line per check; exit code 0 = all passed.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tsdf.h"
#ifdef ABI_HOST_DEBUG_LIB /* linked against build/libtsdf_hip_debug.so: the hooks of include/tsdf_debug.h as well */
#include "tsdf_debug.h"
#endif

/* oracle/tsdf_oracle.c (test infrastructure) */
int tsdf_oracle_voxelize(const float *depth, const int64_t *offsets, const int32_t *headers, int n, int R,
                         const tsdf_cam *cam, int layout, int n_threads, float *out_tsdf, float *out_max_l,
                         float *out_mid_p, int32_t *out_status, float *out_aabb, float *out_grid, float *out_ori);

void tsdf_oracle_normalize_joints(const float *gt, const float *max_l, const float *mid_p, int n, int J, int clamp,
                                  float *out);
void tsdf_oracle_voxels(const float *depth, const int32_t *header, const float *ori, float voxel_len, float trunc_dis,
                        int R, const tsdf_cam *cam, int layout, float *out, int32_t *pixmap);

#define HIP(x)                                                                       \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      printf("FAIL %s: %s\n", #x, hipGetErrorString(e_));                            \
      return 2;                                                                      \
    }                                                                                \
  } while (0)

static uint32_t lcg_state = 12345u;
static float frand(void) { /* [0,1) */
  lcg_state = lcg_state * 1664525u + 1013904223u;
  return (float)(lcg_state >> 8) / 16777216.0f;
}

static int failures = 0;
static void check(int ok, const char *what) {
  printf("%s %s\n", ok ? "ok  " : "FAIL", what);
  if (!ok) ++failures;
}

int main(void) {
  enum { N = 3, R = 32 };
  const int bw[N] = {97, 160, 320}, bh[N] = {120, 131, 240};
  const int left[N] = {100, 40, 0}, top[N] = {60, 30, 0};
  int32_t headers[N][6];
  int64_t offsets[N + 1];
  offsets[0] = 0;
  for (int i = 0; i < N; ++i) {
    const int32_t h[6] = {320, 240, left[i], top[i], left[i] + bw[i], top[i] + bh[i]};
    memcpy(headers[i], h, sizeof h);
    offsets[i + 1] = offsets[i] + (int64_t)bw[i] * bh[i];
  }
  const int64_t total = offsets[N];
  float *depth = (float *)calloc((size_t)total, sizeof(float));
  for (int i = 0; i < N; ++i) {
    const float cx = bw[i] * (0.3f + 0.4f * frand()), cy = bh[i] * (0.3f + 0.4f * frand());
    const float rad = 0.3f * (float)(bw[i] < bh[i] ? bw[i] : bh[i]), base = 350.f + 200.f * frand();
    for (int y = 0; y < bh[i]; ++y)
      for (int x = 0; x < bw[i]; ++x) {
        const float dx = (x - cx) / rad, dy = (y - cy) / rad, rr = dx * dx + dy * dy;
        if (rr <= 1.f && frand() > 0.01f)
          depth[offsets[i] + (int64_t)y * bw[i] + x] = base - 50.f * sqrtf(1.f - rr) + frand();
      }
  }

  const size_t vol = (size_t)3 * R * R * R;
  float *ref_t = (float *)malloc(N * vol * sizeof(float)), *got_t = (float *)malloc(N * vol * sizeof(float));
  float ref_l[N], ref_m[N][3], got_l[N], got_m[N][3], ref_ab[N][6], got_ab[N][6];
  int32_t ref_s[N], got_s[N];

  check(tsdf_version() == TSDF_ABI_VERSION, "tsdf_version() == TSDF_ABI_VERSION");
  check(tsdf_resolution_supported(32) && !tsdf_resolution_supported(30), "tsdf_resolution_supported");
  tsdf_cam cam;
  tsdf_default_cam(&cam);
  check(cam.focal == 241.42 && cam.cx == 160 && cam.cy == 120, "tsdf_default_cam");

  float *d_depth, *d_t, *d_l, *d_m, *d_ab;
  int64_t *d_off;
  int32_t *d_hdr, *d_s;
  hipStream_t stream;
  HIP(hipStreamCreate(&stream));
  HIP(hipMalloc((void **)&d_depth, (size_t)total * 4));
  HIP(hipMalloc((void **)&d_off, sizeof offsets));
  HIP(hipMalloc((void **)&d_hdr, sizeof headers));
  HIP(hipMalloc((void **)&d_t, N * vol * 4));
  HIP(hipMalloc((void **)&d_l, N * 4));
  HIP(hipMalloc((void **)&d_m, N * 12));
  HIP(hipMalloc((void **)&d_s, N * 4));
  HIP(hipMalloc((void **)&d_ab, N * 24));
  HIP(hipMemcpyAsync(d_depth, depth, (size_t)total * 4, hipMemcpyHostToDevice, stream));
  HIP(hipMemcpyAsync(d_off, offsets, sizeof offsets, hipMemcpyHostToDevice, stream));
  HIP(hipMemcpyAsync(d_hdr, headers, sizeof headers, hipMemcpyHostToDevice, stream));

  for (int layout = 0; layout < 2; ++layout) {
    tsdf_oracle_voxelize(depth, offsets, &headers[0][0], N, R, NULL, layout, 1, ref_t, ref_l, &ref_m[0][0], ref_s,
                         &ref_ab[0][0], NULL, NULL);
    const int rc = tsdf_voxelize_hip(d_depth, total, d_off, d_hdr, N, R, NULL, layout, stream, d_t, d_l, d_m, d_s);
    check(rc == TSDF_OK, layout ? "tsdf_voxelize_hip [c,x,y,z] returns TSDF_OK" : "tsdf_voxelize_hip [c,z,y,x] returns TSDF_OK");
    HIP(hipMemcpyAsync(got_t, d_t, N * vol * 4, hipMemcpyDeviceToHost, stream));
    HIP(hipMemcpyAsync(got_l, d_l, N * 4, hipMemcpyDeviceToHost, stream));
    HIP(hipMemcpyAsync(got_m, d_m, N * 12, hipMemcpyDeviceToHost, stream));
    HIP(hipMemcpyAsync(got_s, d_s, N * 4, hipMemcpyDeviceToHost, stream));
    HIP(hipStreamSynchronize(stream));
    check(memcmp(got_s, ref_s, sizeof got_s) == 0 && got_s[0] == TSDF_FRAME_OK, "  per-frame status");
    check(memcmp(got_l, ref_l, sizeof got_l) == 0, "  max_l bit exact");
    check(memcmp(got_m, ref_m, sizeof got_m) == 0, "  mid_p bit exact");
    double worst = 0;
    for (size_t k = 0; k < N * vol; ++k) {
      const double e = fabs((double)got_t[k] - (double)ref_t[k]);
      if (e > worst) worst = e;
    }
    printf("     max |hip - oracle| = %.3g\n", worst);
    check(worst <= 1e-5, "  volume within 1e-5 of the oracle");
  }

  check(tsdf_aabb_hip(d_depth, total, d_off, d_hdr, N, R, NULL, stream, d_ab, NULL, NULL, d_s) == TSDF_OK,
        "tsdf_aabb_hip returns TSDF_OK");
  HIP(hipMemcpyAsync(got_ab, d_ab, N * 24, hipMemcpyDeviceToHost, stream));
  HIP(hipStreamSynchronize(stream));
  check(memcmp(got_ab, ref_ab, sizeof got_ab) == 0, "  AABB bit exact");

  /* ---- ABI v3 ---- */
  {
    enum { J = 21 };
    static float gt[N][3 * J], ref_nor[N][3 * J], got_nor[N][3 * J], got_back[N][3 * J];
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < 3 * J; ++j) gt[i][j] = ref_m[i][j % 3] + (frand() - 0.5f) * 1.3f * ref_l[i];
    tsdf_oracle_normalize_joints(&gt[0][0], ref_l, &ref_m[0][0], N, J, 1, &ref_nor[0][0]);
    float *d_gt, *d_nor, *d_back;
    HIP(hipMalloc((void **)&d_gt, sizeof gt));
    HIP(hipMalloc((void **)&d_nor, sizeof gt));
    HIP(hipMalloc((void **)&d_back, sizeof gt));
    HIP(hipMemcpyAsync(d_gt, gt, sizeof gt, hipMemcpyHostToDevice, stream));
    tsdf_labels lab = {d_gt, J, 1, d_nor, NULL};
    check(tsdf_voxelize_labels_hip(d_depth, total, d_off, d_hdr, N, R, NULL, 0, stream, d_t, d_l, d_m, d_s, &lab) == TSDF_OK,
          "tsdf_voxelize_labels_hip returns TSDF_OK");
    HIP(hipMemcpyAsync(got_nor, d_nor, sizeof gt, hipMemcpyDeviceToHost, stream));
    HIP(hipMemcpyAsync(got_l, d_l, sizeof got_l, hipMemcpyDeviceToHost, stream));
    HIP(hipStreamSynchronize(stream));
    check(memcmp(got_nor, ref_nor, sizeof gt) == 0, "  labels (gt - mid_p)/max_l + 0.5, clamped: bit exact");
    check(memcmp(got_l, ref_l, sizeof got_l) == 0, "  max_l unchanged by the label path");
    int clamped = 0;
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < 3 * J; ++j) clamped += got_nor[i][j] == 0.0f || got_nor[i][j] == 1.0f;
    check(clamped > 0, "  the clamp was exercised");
    check(tsdf_normalize_joints_hip(d_gt, d_l, d_m, N, J, 0, stream, d_nor) == TSDF_OK &&
              tsdf_denormalize_joints_hip(d_nor, d_l, d_m, N, J, stream, d_back) == TSDF_OK,
          "tsdf_normalize_joints_hip / tsdf_denormalize_joints_hip return TSDF_OK");
    HIP(hipMemcpyAsync(got_back, d_back, sizeof gt, hipMemcpyDeviceToHost, stream));
    HIP(hipStreamSynchronize(stream));
    double werr = 0;
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < 3 * J; ++j) werr = fmax(werr, fabs((double)got_back[i][j] - gt[i][j]));
    check(werr < 2e-3, "  denormalize(normalize(gt)) == gt to float32 rounding");
    /* offsets / headers / gt in page-locked host memory (hipHostMalloc), read by the kernel over the link; the plain
     * label entry's d_out_gt_aug is then the joints' device copy */
    {
      int64_t *h_off;
      int32_t *h_hdr;
      float *h_gt, *d_copy;
      static float got_copy[N][3 * J], got_nor2[N][3 * J];
      HIP(hipHostMalloc((void **)&h_off, sizeof offsets, hipHostMallocDefault));
      HIP(hipHostMalloc((void **)&h_hdr, sizeof headers, hipHostMallocDefault));
      HIP(hipHostMalloc((void **)&h_gt, sizeof gt, hipHostMallocDefault));
      HIP(hipMalloc((void **)&d_copy, sizeof gt));
      memcpy(h_off, offsets, sizeof offsets);
      memcpy(h_hdr, headers, sizeof headers);
      memcpy(h_gt, gt, sizeof gt);
      HIP(hipMemsetAsync(d_nor, 0, sizeof gt, stream));
      tsdf_labels hl = {h_gt, J, 1, d_nor, d_copy};
      check(tsdf_voxelize_labels_hip(d_depth, total, h_off, h_hdr, N, R, NULL, 0, stream, d_t, d_l, d_m, d_s, &hl) == TSDF_OK,
            "metadata in page-locked host memory: TSDF_OK");
      HIP(hipMemcpyAsync(got_nor2, d_nor, sizeof gt, hipMemcpyDeviceToHost, stream));
      HIP(hipMemcpyAsync(got_copy, d_copy, sizeof gt, hipMemcpyDeviceToHost, stream));
      HIP(hipMemcpyAsync(got_l, d_l, sizeof got_l, hipMemcpyDeviceToHost, stream));
      HIP(hipStreamSynchronize(stream));
      check(memcmp(got_nor2, ref_nor, sizeof gt) == 0 && memcmp(got_l, ref_l, sizeof got_l) == 0,
            "  labels and max_l bit exact");
      check(memcmp(got_copy, gt, sizeof gt) == 0, "  d_out_gt_aug of the plain entry: the joints' device copy");
      HIP(hipHostFree(h_off));
      HIP(hipHostFree(h_hdr));
      HIP(hipHostFree(h_gt));
      HIP(hipFree(d_copy));
    }
    /* ABI v4: a batch drawn by index from the resident pack (here: frames 2, 0, 2 and one index outside the pack) */
    {
      const int64_t idx[4] = {2, 0, 2, N};
      int64_t *d_idx;
      float *d_t4, *d_l4, *d_m4, *d_nor4;
      int32_t *d_s4;
      static float got_l4[4], got_nor4[4][3 * J];
      int32_t got_s4[4];
      const size_t vol = (size_t)3 * R * R * R;
      float *v4 = (float *)malloc(sizeof(float) * 4 * vol), *v1 = (float *)malloc(sizeof(float) * N * vol);
      HIP(hipMalloc((void **)&d_idx, sizeof idx));
      HIP(hipMalloc((void **)&d_t4, sizeof(float) * 4 * vol));
      HIP(hipMalloc((void **)&d_l4, sizeof(float) * 4));
      HIP(hipMalloc((void **)&d_m4, sizeof(float) * 12));
      HIP(hipMalloc((void **)&d_s4, sizeof(int32_t) * 4));
      HIP(hipMalloc((void **)&d_nor4, sizeof(float) * 4 * 3 * J));
      HIP(hipMemcpyAsync(d_idx, idx, sizeof idx, hipMemcpyHostToDevice, stream));
      tsdf_labels l4 = {d_gt, J, 1, d_nor4, NULL};
      check(tsdf_voxelize_indexed_hip(d_depth, total, d_off, d_hdr, N, d_idx, 4, R, NULL, 0, stream, d_t4, d_l4, d_m4, d_s4,
                                      &l4) == TSDF_OK, "tsdf_voxelize_indexed_hip returns TSDF_OK");
      check(tsdf_voxelize_hip(d_depth, total, d_off, d_hdr, N, R, NULL, 0, stream, d_t, d_l, d_m, d_s) == TSDF_OK,
            "  (the whole pack, for comparison)");
      HIP(hipMemcpyAsync(v4, d_t4, sizeof(float) * 4 * vol, hipMemcpyDeviceToHost, stream));
      HIP(hipMemcpyAsync(v1, d_t, sizeof(float) * N * vol, hipMemcpyDeviceToHost, stream));
      HIP(hipMemcpyAsync(got_l4, d_l4, sizeof got_l4, hipMemcpyDeviceToHost, stream));
      HIP(hipMemcpyAsync(got_s4, d_s4, sizeof got_s4, hipMemcpyDeviceToHost, stream));
      HIP(hipMemcpyAsync(got_nor4, d_nor4, sizeof got_nor4, hipMemcpyDeviceToHost, stream));
      HIP(hipStreamSynchronize(stream));
      int same = 1;
      for (int k = 0; k < 3; ++k) {
        same &= memcmp(v4 + k * vol, v1 + idx[k] * vol, sizeof(float) * vol) == 0;
        same &= got_l4[k] == ref_l[idx[k]] && got_s4[k] == TSDF_FRAME_OK;
        same &= memcmp(got_nor4[k], ref_nor[idx[k]], sizeof(float) * 3 * J) == 0;
      }
      check(same, "  frames 2, 0, 2 of the pack: volumes, max_l, labels bit-identical");
      check(got_s4[3] == TSDF_FRAME_BAD_HEADER, "  an index outside the pack -> TSDF_FRAME_BAD_HEADER");
      check(tsdf_voxelize_indexed_hip(d_depth, total, d_off, d_hdr, N, NULL, 4, R, NULL, 0, stream, d_t4, d_l4, d_m4, d_s4,
                                      NULL) == TSDF_ERR_INVALID_ARG, "  NULL index -> TSDF_ERR_INVALID_ARG");
      free(v4);
      free(v1);
      HIP(hipFree(d_idx));
      HIP(hipFree(d_t4));
      HIP(hipFree(d_l4));
      HIP(hipFree(d_m4));
      HIP(hipFree(d_s4));
      HIP(hipFree(d_nor4));
    }
    lab.n_joints = 0;
    check(tsdf_voxelize_labels_hip(d_depth, total, d_off, d_hdr, N, R, NULL, 0, stream, d_t, d_l, d_m, d_s, &lab) ==
              TSDF_ERR_INVALID_ARG, "n_joints = 0 -> TSDF_ERR_INVALID_ARG");
#ifdef ABI_HOST_DEBUG_LIB
    /* pixel-map diagnostic against the oracle's map, frame by frame, on the oracle's own grid */
    static float ref_grid[N][8], ref_ori[N][3];
    tsdf_oracle_voxelize(depth, offsets, &headers[0][0], N, R, NULL, 0, 1, NULL, NULL, NULL, NULL, NULL, &ref_grid[0][0],
                         &ref_ori[0][0]);
    int32_t *d_pm, *got_pm = (int32_t *)malloc(sizeof(int32_t) * N * R * R * R), *ref_pm = (int32_t *)malloc(sizeof(int32_t) * R * R * R);
    float *scratch = (float *)malloc(sizeof(float) * 3 * R * R * R);
    HIP(hipMalloc((void **)&d_pm, sizeof(int32_t) * N * R * R * R));
    check(tsdf_debug_pixmap_hip(d_depth, total, d_off, d_hdr, N, R, NULL, 0, stream, NULL, d_t, d_pm, d_s) == TSDF_OK,
          "tsdf_debug_pixmap_hip returns TSDF_OK");
    HIP(hipMemcpyAsync(got_pm, d_pm, sizeof(int32_t) * N * R * R * R, hipMemcpyDeviceToHost, stream));
    HIP(hipStreamSynchronize(stream));
    int pm_ok = 1;
    for (int i = 0; i < N; ++i) {
      tsdf_oracle_voxels(depth + offsets[i], headers[i], ref_ori[i], ref_grid[i][4], ref_grid[i][5], R, NULL, 0, scratch, ref_pm);
      pm_ok &= memcmp(ref_pm, got_pm + (size_t)i * R * R * R, sizeof(int32_t) * R * R * R) == 0;
    }
    check(pm_ok, "  pixel maps equal the oracle's exactly");
#endif
    /* the per-thread default stream is a legal stream argument too */
    check(tsdf_voxelize_hip(d_depth, total, d_off, d_hdr, N, R, NULL, 0, hipStreamPerThread, d_t, d_l, d_m, d_s) == TSDF_OK,
          "launch on hipStreamPerThread");
    HIP(hipStreamSynchronize(hipStreamPerThread));
    HIP(hipMemcpy(got_m, d_m, sizeof got_m, hipMemcpyDeviceToHost));
    check(memcmp(got_m, ref_m, sizeof got_m) == 0, "  mid_p bit exact");
    check(tsdf_stream_release(stream) == TSDF_OK && tsdf_stream_release(NULL) == TSDF_OK, "tsdf_stream_release");
  }

  { /* ABI v5: the host-side gather of a shuffled batch, with the length of the source buffer (no GPU involved) */
    const int64_t pick[3] = {2, 0, 2};
    int64_t off2[4];
    float *buf = (float *)malloc((size_t)(2 * (offsets[3] - offsets[2]) + offsets[1]) * sizeof(float));
    const int64_t cap = 2 * (offsets[3] - offsets[2]) + offsets[1];
    check(tsdf_host_gather_frames_n(depth, total, offsets, N, pick, 3, buf, cap, off2, 4) == TSDF_OK &&
              off2[3] == cap && memcmp(buf, depth + offsets[2], (size_t)(offsets[3] - offsets[2]) * 4) == 0 &&
              memcmp(buf + off2[1], depth, (size_t)offsets[1] * 4) == 0,
          "tsdf_host_gather_frames_n: frames 2, 0, 2 back to back");
    check(tsdf_host_gather_frames_n(depth, total - 1, offsets, N, pick, 3, buf, cap, off2, 4) == TSDF_ERR_INVALID_ARG,
          "  a frame that leaves the source buffer -> TSDF_ERR_INVALID_ARG, nothing copied");
    check(tsdf_host_gather_frames_n(depth, total, offsets, N, pick, 3, buf, cap - 1, off2, 4) == TSDF_ERR_INVALID_ARG,
          "  a destination that is too small -> TSDF_ERR_INVALID_ARG");
    free(buf);
  }

  check(tsdf_voxelize_hip(d_depth, total, d_off, d_hdr, 0, R, NULL, 0, stream, d_t, d_l, d_m, d_s) == TSDF_OK,
        "n = 0 is a no-op");
  check(tsdf_voxelize_hip(NULL, total, d_off, d_hdr, N, R, NULL, 0, stream, d_t, d_l, d_m, d_s) == TSDF_ERR_INVALID_ARG,
        "NULL depth -> TSDF_ERR_INVALID_ARG");
  check(tsdf_voxelize_hip(d_depth, total, d_off, d_hdr, N, 30, NULL, 0, stream, d_t, d_l, d_m, d_s) == TSDF_ERR_INVALID_ARG,
        "unsupported R -> TSDF_ERR_INVALID_ARG");
  check(tsdf_voxelize_hip(d_depth, total, d_off, d_hdr, N, R, NULL, 7, stream, d_t, d_l, d_m, d_s) == TSDF_ERR_INVALID_ARG,
        "unknown layout -> TSDF_ERR_INVALID_ARG");
  check(strlen(tsdf_strerror(TSDF_ERR_INVALID_ARG)) > 0 && strlen(tsdf_strerror(-99)) > 0, "tsdf_strerror");
  HIP(hipStreamSynchronize(stream));
  HIP(hipStreamDestroy(stream));
  printf("%s (%d failure%s)\n", failures ? "FAILED" : "PASSED", failures, failures == 1 ? "" : "s");
  return failures ? 1 : 0;
}
