// How fast can ONE CU store?  G persistent 1024-thread workgroups (one per CU) each write their own contiguous
// 3 MiB pieces (a 64^3 volume's worth) with the voxel pass's instruction, global_store_dwordx4 sc1 nt, 1 KiB per wave
// instruction.  If a CU's rate does not rise when fewer CUs write, the 64^3 kernels' store stream is bound per CU
// (outstanding writes x latency), not by the chip's fill rate.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/cu_store_probe.bin tools/probes/cu_store_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void k_store(f4 *__restrict__ out, size_t piece4, int pieces_per_wg, int waves_active) {
  if ((int)(threadIdx.x >> 6) >= waves_active) return;
  const int nthr = waves_active * 64;
  const f4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  for (int p = 0; p < pieces_per_wg; ++p) {
    f4 *o = out + ((size_t)blockIdx.x * pieces_per_wg + p) * piece4;
    for (size_t i = threadIdx.x; i < piece4; i += nthr) {
      if (MODE == 0) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(o + i), "v"(v) : "memory");
      } else if (MODE == 1) {
        __builtin_nontemporal_store(v, o + i);
      } else {
        o[i] = v;
      }
    }
  }
}

int main(int argc, char **argv) {
  const size_t piece = 3u << 20;  // bytes per piece
  const int pieces = argc > 1 ? atoi(argv[1]) : 16;
  f4 *buf;
  CK(hipMalloc(&buf, piece * pieces * 256));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const char *names[3] = {"sc1 nt (asm)", "nt (builtin)", "plain"};
  for (int mode = 0; mode < 3; ++mode) {
    for (int waves : {16, 8, 4}) {
      for (int G : {256, 128, 64, 32, 8, 1}) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
          CK(hipEventRecord(a));
          if (mode == 0) hipLaunchKernelGGL(k_store<0>, dim3(G), dim3(1024), 0, 0, buf, piece / 16, pieces, waves);
          if (mode == 1) hipLaunchKernelGGL(k_store<1>, dim3(G), dim3(1024), 0, 0, buf, piece / 16, pieces, waves);
          if (mode == 2) hipLaunchKernelGGL(k_store<2>, dim3(G), dim3(1024), 0, 0, buf, piece / 16, pieces, waves);
          CK(hipEventRecord(b));
          CK(hipEventSynchronize(b));
          float ms;
          CK(hipEventElapsedTime(&ms, a, b));
          if (rep && ms < best) best = ms;
        }
        const double bytes = (double)piece * pieces * G;
        printf("%-13s waves/CU %2d  CUs %3d: %8.1f us  per CU %6.1f GB/s  total %7.1f GB/s\n", names[mode], waves, G,
               best * 1e3, bytes / G / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e9);
      }
    }
  }
  return 0;
}
