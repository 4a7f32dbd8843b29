// Do the chip's XCDs serve a store stream equally?  256 persistent 1024-thread workgroups (one per CU; workgroup b runs on
// XCD b % 8) each write 3 MiB volumes in the voxel pass's adopted order (2-slice units of one slab, tools/probes/
// vol_store_probe.hip "N"), store-only, with four cache policies; every workgroup stamps its end (s_memrealtime).
// The 64^3 kernels' stamps show odd XCDs ending ~10 % after even ones (profiles/r04/xcd_parity.log): is that the memory
// system or the kernels?
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/xcd_store_probe.bin tools/probes/xcd_store_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int POL>
__device__ __forceinline__ void st(f4 *p, f4 v) {
  if (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
template <int POL>
__global__ __launch_bounds__(1024) void k_vol(f4 *__restrict__ out, int pieces_even, int pieces_odd, unsigned long long *stamps) {
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  const f4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  constexpr int PL = 65536, SL = 1024;
  if (t == 0) stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  const int pieces = (blockIdx.x & 1) ? pieces_odd : pieces_even, pmax = pieces_even > pieces_odd ? pieces_even : pieces_odd;
  for (int p = 0; p < pieces; ++p) {
    f4 *o = out + ((size_t)blockIdx.x * pmax + p) * 3 * PL;
    for (int k2 = 0; k2 < 32; ++k2)
      for (int dz = 0; dz < 2; ++dz)
        for (int c = 0; c < 3; ++c) st<POL>(o + c * PL + (32 * (k2 & 1) + 2 * w + dz) * SL + (k2 >> 1) * 64 + l, v);
  }
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (t == 0) stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
}
int main(int argc, char **argv) {
  const int pe = argc > 1 ? atoi(argv[1]) : 4, po = argc > 2 ? atoi(argv[2]) : pe, pieces = pe > po ? pe : po;
  const int G = 256;
  f4 *buf;
  unsigned long long *d_st;
  CK(hipMalloc(&buf, (size_t)(3u << 20) * pieces * G));
  CK(hipMalloc(&d_st, sizeof(unsigned long long) * 2 * G));
  std::vector<unsigned long long> h(2 * G);
  const char *names[4] = {"sc1 nt", "nt", "plain", "sc0 sc1 nt"};
  for (int round = 0; round < 2; ++round)
    for (int pol = 0; pol < 4; pol += 3) {
      for (int rep = 0; rep < 3; ++rep) {
        switch (pol) {
          case 0: hipLaunchKernelGGL(k_vol<0>, dim3(G), dim3(1024), 0, 0, buf, pe, po, d_st); break;
          case 1: hipLaunchKernelGGL(k_vol<1>, dim3(G), dim3(1024), 0, 0, buf, pe, po, d_st); break;
          case 2: hipLaunchKernelGGL(k_vol<2>, dim3(G), dim3(1024), 0, 0, buf, pe, po, d_st); break;
          case 3: hipLaunchKernelGGL(k_vol<3>, dim3(G), dim3(1024), 0, 0, buf, pe, po, d_st); break;
        }
        CK(hipDeviceSynchronize());
      }
      CK(hipMemcpy(h.data(), d_st, sizeof(unsigned long long) * 2 * G, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull, t1 = 0;
      for (int b = 0; b < G; ++b) { t0 = std::min(t0, h[2 * b]); t1 = std::max(t1, h[2 * b + 1]); }
      double per[8] = {0}, beg[8] = {0};
      for (int b = 0; b < G; ++b) {
        per[b & 7] += (double)(h[2 * b + 1] - t0) / 100.0 / (G / 8);
        beg[b & 7] += (double)(h[2 * b] - t0) / 100.0 / (G / 8);
      }
      printf("%-11s %d/%d volumes per even/odd CU (%.2f GB): span %7.1f us = %6.1f GB/s; mean end per XCD:", names[pol], pe, po,
             (double)(3u << 20) * (pe + po) * (G / 2) / 1e9, (double)(t1 - t0) / 100.0,
             (double)(3u << 20) * (pe + po) * (G / 2) / ((double)(t1 - t0) / 100.0 * 1e-6) / 1e9);
      for (int x = 0; x < 8; ++x) printf(" %6.1f", per[x]);
      printf("; mean start:");
      for (int x = 0; x < 8; ++x) printf(" %4.1f", beg[x]);
      printf("\n");
    }
  return 0;
}
