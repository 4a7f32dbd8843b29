// valu_rate_probe.hip — issue cost of the float64 / conversion instructions the augmented voxel pass is made of
// (gfx950).  Not product code: a measuring stick for DESIGN.md's instruction budget of phase2_aug.
// Each kernel runs ITER x 8 independent instructions of one kind per wave, 16 waves per CU (4 per SIMD, like the
// voxelizer), one workgroup per CU; reported: cycles per instruction per SIMD at the measured clock, relative to v_add_f32.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITER = 4096;

#define KERNEL(NAME, DECL, BODY, SINK)                                                     \
  __global__ __launch_bounds__(1024) void NAME(double *out, double seed) {                  \
    DECL;                                                                                  \
    for (int i = 0; i < ITER; ++i) { BODY; }                                               \
    SINK;                                                                                  \
  }

// 8 independent accumulators so that dependent-issue latency never limits the stream
#define D8 double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
           double b = seed * 0.5 + threadIdx.x, c = seed * 0.25
#define SINKD if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[threadIdx.x] = a0
#define OP3D(op) asm volatile(op " %0, %0, %8, %9\n\t" op " %1, %1, %8, %9\n\t" op " %2, %2, %8, %9\n\t" op " %3, %3, %8, %9\n\t" \
                              op " %4, %4, %8, %9\n\t" op " %5, %5, %8, %9\n\t" op " %6, %6, %8, %9\n\t" op " %7, %7, %8, %9"      \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c))
#define OP2D(op) asm volatile(op " %0, %0, %8\n\t" op " %1, %1, %8\n\t" op " %2, %2, %8\n\t" op " %3, %3, %8\n\t" \
                              op " %4, %4, %8\n\t" op " %5, %5, %8\n\t" op " %6, %6, %8\n\t" op " %7, %7, %8"      \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b))
#define OP1D(op) asm volatile(op " %0, %0\n\t" op " %1, %1\n\t" op " %2, %2\n\t" op " %3, %3\n\t" \
                              op " %4, %4\n\t" op " %5, %5\n\t" op " %6, %6\n\t" op " %7, %7"      \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))

KERNEL(k_fma_f64, D8, OP3D("v_fma_f64"), SINKD)
KERNEL(k_add_f64, D8, OP2D("v_add_f64"), SINKD)
KERNEL(k_mul_f64, D8, OP2D("v_mul_f64"), SINKD)
KERNEL(k_rcp_f64, D8, OP1D("v_rcp_f64"), SINKD)
KERNEL(k_rndne_f64, D8, OP1D("v_rndne_f64"), SINKD)

// float32 accumulators
#define F8 float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
           float b = seed * 0.5f + threadIdx.x, c = seed * 0.25f
#define SINKF if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[threadIdx.x] = a0
KERNEL(k_add_f32, F8, OP2D("v_add_f32"), SINKF)
KERNEL(k_fma_f32, F8, OP3D("v_fma_f32"), SINKF)
KERNEL(k_rcp_f32, F8, OP1D("v_rcp_f32"), SINKF)

// conversions: source and destination of different width -> separate register sets
#define CVT(op, TD, TS)                                                                                       \
  TS s0 = (TS)(seed + threadIdx.x), s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;                                    \
  TD d0, d1, d2, d3, d4, d5, d6, d7;                                                                          \
  TD acc = 0;                                                                                                 \
  for (int i = 0; i < ITER; ++i) {                                                                            \
    asm volatile(op " %0, %8\n\t" op " %1, %9\n\t" op " %2, %10\n\t" op " %3, %11\n\t"                        \
                 op " %4, %8\n\t" op " %5, %9\n\t" op " %6, %10\n\t" op " %7, %11"                            \
                 : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3), "=v"(d4), "=v"(d5), "=v"(d6), "=v"(d7)              \
                 : "v"(s0), "v"(s1), "v"(s2), "v"(s3));                                                       \
  }                                                                                                           \
  acc = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;                                                                \
  if ((double)acc == 12345.678) out[threadIdx.x] = (double)acc;

__global__ __launch_bounds__(1024) void k_cvt_f64_i32(double *out, double seed) { CVT("v_cvt_f64_i32", double, int) }
__global__ __launch_bounds__(1024) void k_cvt_i32_f64(double *out, double seed) { CVT("v_cvt_i32_f64", int, double) }
__global__ __launch_bounds__(1024) void k_cvt_f64_f32(double *out, double seed) { CVT("v_cvt_f64_f32", double, float) }
__global__ __launch_bounds__(1024) void k_cvt_f32_f64(double *out, double seed) { CVT("v_cvt_f32_f64", float, double) }
__global__ __launch_bounds__(1024) void k_cvt_f32_i32(double *out, double seed) { CVT("v_cvt_f32_i32", float, int) }

// compares write VCC: 8 per iteration
__global__ __launch_bounds__(1024) void k_cmp_f64(double *out, double seed) {
  double a = seed + threadIdx.x, b = seed * 0.5;
  unsigned long long m = 0;
  for (int i = 0; i < ITER; ++i) {
    asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\tv_cmp_lt_f64 vcc, %2, %1\n\tv_cmp_lt_f64 vcc, %1, %2\n\tv_cmp_lt_f64 vcc, %2, %1\n\t"
                 "v_cmp_lt_f64 vcc, %1, %2\n\tv_cmp_lt_f64 vcc, %2, %1\n\tv_cmp_lt_f64 vcc, %1, %2\n\tv_cmp_lt_f64 %0, %2, %1"
                 : "=s"(m) : "v"(a), "v"(b) : "vcc");
  }
  if (m == 12345ull) out[threadIdx.x] = a;
}
__global__ __launch_bounds__(1024) void k_cmp_f32(double *out, double seed) {
  float a = seed + threadIdx.x, b = seed * 0.5;
  unsigned long long m = 0;
  for (int i = 0; i < ITER; ++i) {
    asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cmp_lt_f32 vcc, %2, %1\n\tv_cmp_lt_f32 vcc, %1, %2\n\tv_cmp_lt_f32 vcc, %2, %1\n\t"
                 "v_cmp_lt_f32 vcc, %1, %2\n\tv_cmp_lt_f32 vcc, %2, %1\n\tv_cmp_lt_f32 vcc, %1, %2\n\tv_cmp_lt_f32 %0, %2, %1"
                 : "=s"(m) : "v"(a), "v"(b) : "vcc");
  }
  if (m == 12345ull) out[threadIdx.x] = a;
}

template <class K>
double time_us(K kern, double *out, int grid) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), 0, 0, out, 1.0);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), 0, 0, out, 1.0);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best * 1e3;
}

int main() {
  int cus = 0;
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  double *out;
  CHECK(hipMalloc(&out, 1024 * sizeof(double)));
  struct { const char *name; double us; } r[] = {
    {"v_add_f32", time_us(k_add_f32, out, cus)}, {"v_fma_f32", time_us(k_fma_f32, out, cus)},
    {"v_rcp_f32", time_us(k_rcp_f32, out, cus)}, {"v_cmp_lt_f32", time_us(k_cmp_f32, out, cus)},
    {"v_add_f64", time_us(k_add_f64, out, cus)}, {"v_mul_f64", time_us(k_mul_f64, out, cus)},
    {"v_fma_f64", time_us(k_fma_f64, out, cus)}, {"v_rcp_f64", time_us(k_rcp_f64, out, cus)},
    {"v_rndne_f64", time_us(k_rndne_f64, out, cus)}, {"v_cmp_lt_f64", time_us(k_cmp_f64, out, cus)},
    {"v_cvt_f64_i32", time_us(k_cvt_f64_i32, out, cus)}, {"v_cvt_i32_f64", time_us(k_cvt_i32_f64, out, cus)},
    {"v_cvt_f64_f32", time_us(k_cvt_f64_f32, out, cus)}, {"v_cvt_f32_f64", time_us(k_cvt_f32_f64, out, cus)},
    {"v_cvt_f32_i32", time_us(k_cvt_f32_i32, out, cus)},
  };
  const double base = r[0].us;
  // per SIMD: 4 waves x ITER x 8 instructions in `us`
  printf("%d CUs, 16 waves per CU, %d x 8 instructions per wave\n", cus, ITER);
  for (auto &x : r)
    printf("%-16s %9.1f us   %.2f x v_add_f32   (%.2f ns per wave-instruction per SIMD)\n", x.name, x.us, x.us / base,
           x.us * 1e3 / (4.0 * ITER * 8));
  return 0;
}
