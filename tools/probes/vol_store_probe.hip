// Which ORDER of writing a [3][64][64][64] float volume does the memory system like?  256 persistent 1024-thread
// workgroups (one per CU) each write `pieces` private 3 MiB volumes with global_store_dwordx4 sc1 nt, 1 KiB per wave
// instruction, in different orders.  Store-only (no reads, no arithmetic): an upper bound for the 64^3 voxel pass.
//   A  sequential: the workgroup sweeps its volume front to back, 16 KiB per step
//   B  slice-major: per slice z the 16 waves write the slice's 16 KiB in each of the 3 channel planes (what the voxel
//      pass does when its waves stay in step)
//   C  B with the waves out of step: wave w starts at slice 4w (what oldest-first issue does to the static split)
//   D  dynamic units: at any time the 16 waves work on the 16 four-slice chunks of ONE 4-row slab
//   E  grid-stride fill over the whole buffer (tools/probes/hbm_probe.hip's write_only_nt)
//   F  B, one plane after the other (all of channel 0, then 1, then 2)
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/vol_store_probe.bin tools/probes/vol_store_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void st(f4 *p, f4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(1024) void k_vol(f4 *__restrict__ out, int pieces, int total_pieces) {
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  const f4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  constexpr int PL = 65536, SL = 1024;  // f4 per plane / per slice
  if (MODE == 4) {
    const size_t n4 = (size_t)total_pieces * 3 * PL, stride = (size_t)gridDim.x * 1024;
    for (size_t i = (size_t)blockIdx.x * 1024 + t; i < n4; i += stride) st(out + i, v);
    return;
  }
  for (int p = 0; p < pieces; ++p) {
    f4 *o = out + ((size_t)blockIdx.x * pieces + p) * 3 * PL;
    if (MODE == 0) {
      for (int i = 0; i < 3 * PL; i += 1024) st(o + i + t, v);
    } else if (MODE == 1) {
      for (int z = 0; z < 64; ++z)
        for (int c = 0; c < 3; ++c) st(o + c * PL + z * SL + t, v);
    } else if (MODE == 2) {
      for (int zz = 0; zz < 64; ++zz) {
        const int z = (zz + 4 * w) & 63;
        for (int c = 0; c < 3; ++c) st(o + c * PL + z * SL + t, v);
      }
    } else if (MODE == 3) {
      for (int k = 0; k < 16; ++k) {          // step k: slab k, wave w takes chunk w (slices 4w..4w+3)
        for (int dz = 0; dz < 4; ++dz)
          for (int c = 0; c < 3; ++c) st(o + c * PL + (4 * w + dz) * SL + k * 64 + l, v);
      }
    } else if (MODE == 5) {
      for (int c = 0; c < 3; ++c)
        for (int z = 0; z < 64; ++z) st(o + c * PL + z * SL + t, v);
    } else if (MODE == 6) {   // D, but every wave on a different slab as well: slab (k + w) & 15, chunk w
      for (int k = 0; k < 16; ++k)
        for (int dz = 0; dz < 4; ++dz)
          for (int c = 0; c < 3; ++c) st(o + c * PL + (4 * w + dz) * SL + ((k + w) & 15) * 64 + l, v);
    } else if (MODE == 7) {   // D with the channel loop outside the slice loop
      for (int k = 0; k < 16; ++k)
        for (int c = 0; c < 3; ++c)
          for (int dz = 0; dz < 4; ++dz) st(o + c * PL + (4 * w + dz) * SL + k * 64 + l, v);
    } else if (MODE == 8) {   // one-slice units: step k2 (0..63): wave w writes slab (k2 >> 2), slice 16 * (k2 & 3) + w
      for (int k2 = 0; k2 < 64; ++k2)
        for (int c = 0; c < 3; ++c) st(o + c * PL + (16 * (k2 & 3) + w) * SL + (k2 >> 2) * 64 + l, v);
    } else if (MODE == 9) {   // eight-slice units: waves 0-7 on slab 2k, waves 8-15 on slab 2k+1
      for (int k = 0; k < 8; ++k)
        for (int dz = 0; dz < 8; ++dz)
          for (int c = 0; c < 3; ++c) st(o + c * PL + (8 * (w & 7) + dz) * SL + (2 * k + (w >> 3)) * 64 + l, v);
    } else if (MODE == 10) {  // static split, waves in step (B) but each wave keeps its slab and walks z: = B (control)
      for (int z = 0; z < 64; ++z)
        for (int c = 0; c < 3; ++c) st(o + c * PL + z * SL + w * 64 + l, v);
    } else if (MODE == 11) {  // static split with a per-wave z rotation of 4w AND channel rotation
      for (int zz = 0; zz < 64; ++zz) {
        const int z = (zz + 4 * w) & 63;
        for (int cc = 0; cc < 3; ++cc) { const int c = (cc + w) % 3; st(o + c * PL + z * SL + w * 64 + l, v); }
      }
    } else if (MODE == 12) {  // D with 16-slice chunks: 4 slabs in flight, 4 waves each
      for (int k = 0; k < 4; ++k)
        for (int dz = 0; dz < 16; ++dz)
          for (int c = 0; c < 3; ++c) st(o + c * PL + (16 * (w & 3) + dz) * SL + (4 * k + (w >> 2)) * 64 + l, v);
    } else if (MODE == 13) {  // two-slice units
      for (int k2 = 0; k2 < 32; ++k2)
        for (int dz = 0; dz < 2; ++dz)
          for (int c = 0; c < 3; ++c) st(o + c * PL + (32 * (k2 & 1) + 2 * w + dz) * SL + (k2 >> 1) * 64 + l, v);
    }
  }
}

int main(int argc, char **argv) {
  const int pieces = argc > 1 ? atoi(argv[1]) : 4;
  const int G = 256;
  f4 *buf;
  CK(hipMalloc(&buf, (size_t)(3u << 20) * pieces * G));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const char *names[14] = {"A sequential", "B slice-major", "C B out of step", "D dyn units(4)", "E grid-stride", "F plane by plane", "G D+slab spread", "H D chan outer", "I 1-slice units", "J 8-slice units", "K static in step", "L static rot z,c", "M 16-slice units", "N 2-slice units"};
  for (int round = 0; round < 2; ++round)
    for (int mode = 0; mode < 14; ++mode) {
      float best = 1e30f, sum = 0;
      for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(a));
        switch (mode) {
          case 0: hipLaunchKernelGGL(k_vol<0>, dim3(G), dim3(1024), 0, 0, buf, pieces, pieces * G); break;
          case 1: hipLaunchKernelGGL(k_vol<1>, dim3(G), dim3(1024), 0, 0, buf, pieces, pieces * G); break;
          case 2: hipLaunchKernelGGL(k_vol<2>, dim3(G), dim3(1024), 0, 0, buf, pieces, pieces * G); break;
          case 3: hipLaunchKernelGGL(k_vol<3>, dim3(G), dim3(1024), 0, 0, buf, pieces, pieces * G); break;
          case 4: hipLaunchKernelGGL(k_vol<4>, dim3(G), dim3(1024), 0, 0, buf, pieces, pieces * G); break;
          case 5: hipLaunchKernelGGL(k_vol<5>, dim3(G), dim3(1024), 0, 0, buf, pieces, pieces * G); break;
          case 6: hipLaunchKernelGGL(k_vol<6>, dim3(G), dim3(1024), 0, 0, buf, pieces, pieces * G); break;
          case 7: hipLaunchKernelGGL(k_vol<7>, dim3(G), dim3(1024), 0, 0, buf, pieces, pieces * G); break;
          case 8: hipLaunchKernelGGL(k_vol<8>, dim3(G), dim3(1024), 0, 0, buf, pieces, pieces * G); break;
          case 9: hipLaunchKernelGGL(k_vol<9>, dim3(G), dim3(1024), 0, 0, buf, pieces, pieces * G); break;
          case 10: hipLaunchKernelGGL(k_vol<10>, dim3(G), dim3(1024), 0, 0, buf, pieces, pieces * G); break;
          case 11: hipLaunchKernelGGL(k_vol<11>, dim3(G), dim3(1024), 0, 0, buf, pieces, pieces * G); break;
          case 12: hipLaunchKernelGGL(k_vol<12>, dim3(G), dim3(1024), 0, 0, buf, pieces, pieces * G); break;
          case 13: hipLaunchKernelGGL(k_vol<13>, dim3(G), dim3(1024), 0, 0, buf, pieces, pieces * G); break;
        }
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep) { sum += ms; if (ms < best) best = ms; }
      }
      const double bytes = (double)(3u << 20) * pieces * G;
      printf("%-17s %d x 256 volumes: mean %7.1f us (min %7.1f)  %7.1f GB/s\n", names[mode], pieces, sum / 5 * 1e3, best * 1e3,
             bytes / (sum / 5 * 1e-3) / 1e9);
    }
  return 0;
}
