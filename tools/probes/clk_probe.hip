// Shader clock of the GPU as a kernel sees it: a one-wave kernel reads s_memtime (shader clock cycles) and s_memrealtime
// (100 MHz) before and after a short spin and stores both differences.  Launched between other kernels it tells at what
// engine clock they ran (tools/exp_warmup.py --clock: is the augmented kernel's start-up transient the clock?).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/probes/libclk_probe.so tools/probes/clk_probe.hip
#include <hip/hip_runtime.h>
__global__ void k_clk(unsigned long long *out, int spin) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float x = (float)threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[0] = c1 - c0;
    out[1] = r1 - r0;
    out[2] = (unsigned long long)x;
  }
}
extern "C" int clk_probe_launch(void *stream, unsigned long long *d_out, int spin) {
  hipLaunchKernelGGL(k_clk, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), d_out, spin);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
