// Probe (GPU box): does global_load_lds_dwordx4 accept 4-byte-aligned sources, and do EXEC-masked
// lanes leave their 16-byte LDS slot untouched?   hipcc --offload-arch=gfx950 -O2 glds_probe.hip -o glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* __restrict__ src, float* __restrict__ dst, int shift, int nactive) {
  __shared__ __attribute__((aligned(16))) float stage[256];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 256; i += 64) stage[i] = -1.0f;
  __syncthreads();
  const float* g = src + 4 * lane + shift;
  if (lane < nactive)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)stage, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) dst[i] = stage[i];
}
int main() {
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, 4096); hipMalloc(&o, 1024);
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  int bad = 0;
  for (int shift = 0; shift < 4; ++shift)
    for (int nact : {64, 40, 1}) {
      k<<<1, 64>>>(d, o, shift, nact);
      std::vector<float> r(256);
      hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
      int b = 0;
      for (int i = 0; i < 256; ++i) {
        float want = (i / 4 < nact) ? (float)(i + shift) : -1.0f;
        if (r[i] != want) ++b;
      }
      printf("shift %d nactive %d: %d mismatches (r[0..4]= %g %g %g %g %g, r[252..255]= %g %g %g %g)\n", shift, nact, b,
             r[0], r[1], r[2], r[3], r[4], r[252], r[253], r[254], r[255]);
      bad += b;
    }
  printf(bad ? "PROBE FAIL\n" : "PROBE OK\n");
  return bad != 0;
}
