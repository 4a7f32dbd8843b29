// rcp64_probe.hip — how accurate is v_rcp_f64 on gfx950, and how many Newton steps does a correctly rounded -F/v need?
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k(const double *x, double *r0, double *q1, double *q2, double F, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i], nn = -F;
  double r = __builtin_amdgcn_rcp(v);
  r0[i] = r;
  double ra = __builtin_fma(r, __builtin_fma(-v, r, 1.0), r);          // one Newton step
  double qa = nn * ra;
  q1[i] = __builtin_fma(__builtin_fma(-v, qa, nn), ra, qa);            // + Markstein correction
  double rb = __builtin_fma(ra, __builtin_fma(-v, ra, 1.0), ra);       // two Newton steps (what the kernel does)
  double qb = nn * rb;
  q2[i] = __builtin_fma(__builtin_fma(-v, qb, nn), rb, qb);
}
int main() {
  const int n = 1 << 22;
  double *hx = (double *)malloc(n * 8), *h0 = (double *)malloc(n * 8), *h1 = (double *)malloc(n * 8), *h2 = (double *)malloc(n * 8);
  srand48(7);
  for (int i = 0; i < n; ++i) hx[i] = -(50.0 + 2950.0 * drand48()) * (1.0 + 1e-9 * drand48());   // v_z: -50 .. -3000 mm
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, 241.42, n);
  hipMemcpy(h0, d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, n * 8, hipMemcpyDeviceToHost);
  double worst = 0; long bad1 = 0, bad2 = 0;
  for (int i = 0; i < n; ++i) {
    const long double ex = 1.0L / (long double)hx[i];
    const double rel = fabs((double)(((long double)h0[i] - ex) / ex));
    if (rel > worst) worst = rel;
    const double q = -241.42 / hx[i];   // correctly rounded by the host's divider
    bad1 += h1[i] != q; bad2 += h2[i] != q;
  }
  printf("v_rcp_f64: worst relative error %.3g (2^%.1f) over %d values\n", worst, log2(worst), n);
  printf("-F/v with ONE Newton step + correction: %ld of %d differ from the IEEE quotient; with TWO: %ld\n", bad1, n, bad2);
  return 0;
}
