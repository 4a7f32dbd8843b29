// The store order of a [3][32][32][32] float volume (393,216 B), written by 512-thread groups, two per 1024-thread
// workgroup, one workgroup per CU — store-only upper bounds for the 32^3 voxel pass (tools/probes/vol_store_probe.hip is
// the 64^3 version, where the order is worth up to 20 % on some boxes).  1 KiB per wave instruction, sc1 nt.
//   A  contiguous sweep of the volume by the group (8 KiB per step)
//   B  the voxel pass today: per step two slices (waves 0-3 slice z, waves 4-7 slice z+1), the three planes back to back
//   C  units (slab of 8 rows) x (2 slices): the 8 waves on 8 consecutive slice pairs of ONE slab
//   D  units (slab) x (4 slices): 8 waves on the 8 four-slice chunks of one slab
//   E  units (slab) x (1 slice): 8 waves on 8 consecutive slices of one slab
//   F  C with the plane loop outside the slice loop
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/vol32_store_probe.bin tools/probes/vol32_store_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ void st(f4 *p, f4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
template <int MODE>
__global__ __launch_bounds__(1024) void k_vol(f4 *__restrict__ out, int n_frames) {
  const int group = threadIdx.x >> 9, t = threadIdx.x & 511, w = t >> 6, l = t & 63;
  const f4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  constexpr int PL = 8192, SL = 256;   // f4 per plane (128 KiB) / per slice (4 KiB); a wave tile = 64 f4 = 8 rows
  for (int fr = blockIdx.x * 2 + group; fr < n_frames; fr += gridDim.x * 2) {
    f4 *o = out + (size_t)fr * 3 * PL;
    if (MODE == 0) {
      for (int i = 0; i < 3 * PL; i += 512) st(o + i + t, v);
    } else if (MODE == 1) {
      for (int z = (w >> 2); z < 32; z += 2)
        for (int c = 0; c < 3; ++c) st(o + c * PL + z * SL + (w & 3) * 64 + l, v);
    } else if (MODE == 2) {
      for (int k = 0; k < 8; ++k) {          // step k: slab k & 3, slices 16 * (k >> 2) + 2w, +1
        for (int dz = 0; dz < 2; ++dz)
          for (int c = 0; c < 3; ++c) st(o + c * PL + (16 * (k >> 2) + 2 * w + dz) * SL + (k & 3) * 64 + l, v);
      }
    } else if (MODE == 3) {
      for (int k = 0; k < 4; ++k)
        for (int dz = 0; dz < 4; ++dz)
          for (int c = 0; c < 3; ++c) st(o + c * PL + (4 * w + dz) * SL + k * 64 + l, v);
    } else if (MODE == 4) {
      for (int k = 0; k < 16; ++k)
        for (int c = 0; c < 3; ++c) st(o + c * PL + (8 * (k >> 2) + w) * SL + (k & 3) * 64 + l, v);
    } else if (MODE == 5) {
      for (int k = 0; k < 8; ++k)
        for (int c = 0; c < 3; ++c)
          for (int dz = 0; dz < 2; ++dz) st(o + c * PL + (16 * (k >> 2) + 2 * w + dz) * SL + (k & 3) * 64 + l, v);
    }
  }
}
int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1024;
  f4 *buf;
  CK(hipMalloc(&buf, (size_t)393216 * n));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const char *names[6] = {"A contiguous", "B today (z pairs)", "C slab x 2 slices", "D slab x 4 slices", "E slab x 1 slice", "F C, planes outer"};
  for (int round = 0; round < 2; ++round)
    for (int mode = 0; mode < 6; ++mode) {
      float best = 1e30f, sum = 0;
      for (int rep = 0; rep < 21; ++rep) {
        CK(hipEventRecord(a));
        switch (mode) {
          case 0: hipLaunchKernelGGL(k_vol<0>, dim3(256), dim3(1024), 0, 0, buf, n); break;
          case 1: hipLaunchKernelGGL(k_vol<1>, dim3(256), dim3(1024), 0, 0, buf, n); break;
          case 2: hipLaunchKernelGGL(k_vol<2>, dim3(256), dim3(1024), 0, 0, buf, n); break;
          case 3: hipLaunchKernelGGL(k_vol<3>, dim3(256), dim3(1024), 0, 0, buf, n); break;
          case 4: hipLaunchKernelGGL(k_vol<4>, dim3(256), dim3(1024), 0, 0, buf, n); break;
          case 5: hipLaunchKernelGGL(k_vol<5>, dim3(256), dim3(1024), 0, 0, buf, n); break;
        }
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep) { sum += ms; if (ms < best) best = ms; }
      }
      const double bytes = 393216.0 * n;
      printf("%-20s %d volumes: mean %7.1f us (min %7.1f)  %7.1f GB/s\n", names[mode], n, sum / 20 * 1e3, best * 1e3,
             bytes / (sum / 20 * 1e-3) / 1e9);
    }
  return 0;
}
