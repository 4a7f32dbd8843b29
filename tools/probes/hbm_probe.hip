// Practical HBM ceilings of the box the bench runs on: read-only, write-only (plain and non-temporal)
// and copy streams, 16 B per lane, persistent grid.  Context for roofline.frac: the spec peak
// is 8 TB/s, what a stream reaches is lower and differs between reads and writes.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/hbm_probe.bin tools/probes/hbm_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(1024) void k_read(const f4 *__restrict__ a, size_t n4, float *sink) {
  f4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) acc += a[i];
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = 1.f;
}
template <bool NT>
__global__ __launch_bounds__(1024) void k_write(f4 *__restrict__ b, size_t n4) {
  const f4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    if (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v;
  }
}
template <bool NT>
__global__ __launch_bounds__(1024) void k_copy(const f4 *__restrict__ a, f4 *__restrict__ b, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f4 v = a[i];
    if (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v;
  }
}
// the voxelizer's mix: 307,200 B read per 393,216 B written
template <bool NT>
__global__ __launch_bounds__(1024) void k_mix(const f4 *__restrict__ a, f4 *__restrict__ b, size_t nr4, size_t nw4, float *sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  f4 acc = {0, 0, 0, 0};
  for (size_t i = t; i < nr4; i += stride) acc += a[i];
  for (size_t i = t; i < nw4; i += stride) { if (NT) __builtin_nontemporal_store(acc, b + i); else b[i] = acc; }
  if (acc.x == 12345.678f) *sink = 1.f;
}

// Volume-shaped writes: 512-thread groups each write whole 32^3 x 3 float volumes (393,216 B) either as one
// contiguous stream or the way the voxel pass does (thread -> 4 x-voxels, z pairs, the 3 channel planes 128 KiB
// apart written back to back).  Tells whether the [c][z][y][x] plane stride costs anything at the HBM.
template <bool NT, bool PLANES, int ROT = 0>
__global__ __launch_bounds__(1024) void k_volumes(f4 *__restrict__ out, int n_frames) {
  const int group = threadIdx.x >> 9, t = threadIdx.x & 511;
  const f4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  for (int fr = blockIdx.x * 2 + group; fr < n_frames; fr += gridDim.x * 2) {
    f4 *o = out + (size_t)fr * (3 * 32768 / 4);
    if (PLANES) {
      const int gi = t & 255, s0 = t >> 8;           // (y, x4) and the z parity
      for (int zz = s0; zz < 32; zz += 2) {
        const int z = ROT ? (zz + fr * ROT) & 31 : zz;   // ROT: every frame starts at a different slice
        const int e4 = (z * 32 * 32 + gi * 4) / 4;   // float4 index inside a plane
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if (NT) __builtin_nontemporal_store(v, o + c * 8192 + e4); else o[c * 8192 + e4] = v;
        }
      }
    } else {
      for (int i = t; i < 3 * 8192; i += 512) {
        if (NT) __builtin_nontemporal_store(v, o + i); else o[i] = v;
      }
    }
  }
}

// The same volumes written by T threads each (T = 256, 512 or 1024: 4, 2 or 1 volume streams per workgroup), contiguous:
// does the NUMBER of concurrent streams matter?
template <bool NT, int T>
__global__ __launch_bounds__(1024) void k_volumes_t(f4 *__restrict__ out, int n_frames) {
  constexpr int G = 1024 / T;
  const int group = threadIdx.x / T, t = threadIdx.x % T;
  const f4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  for (int fr = blockIdx.x * G + group; fr < n_frames; fr += gridDim.x * G) {
    f4 *o = out + (size_t)fr * (3 * 32768 / 4);
    for (int i = t; i < 3 * 8192; i += T) {
      if (NT) __builtin_nontemporal_store(v, o + i); else o[i] = v;
    }
  }
}
// ... and volumes written cooperatively by W consecutive workgroups (W = 2, 4, 8): fewer, faster streams
template <bool NT, int W>
__global__ __launch_bounds__(1024) void k_volumes_w(f4 *__restrict__ out, int n_frames) {
  const int team = blockIdx.x / W, part = blockIdx.x % W, teams = gridDim.x / W;
  const f4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  for (int fr = team; fr < n_frames; fr += teams) {
    f4 *o = out + (size_t)fr * (3 * 32768 / 4);
    for (int i = part * 1024 + threadIdx.x; i < 3 * 8192; i += 1024 * W) {
      if (NT) __builtin_nontemporal_store(v, o + i); else o[i] = v;
    }
  }
}

int main(int argc, char **argv) {
  const size_t bytes = (argc > 1 ? atoll(argv[1]) : 1024) << 20;
  const size_t n4 = bytes / 16;
  f4 *a, *b; float *sink;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int K = 20;
  for (int mult : {1, 2, 4, 8}) {
    const int grid = p.multiProcessorCount * mult;
    auto run = [&](const char *name, auto launch, double moved) {
      for (int i = 0; i < 3; ++i) launch();
      (void)hipEventRecord(e0, 0);
      for (int i = 0; i < K; ++i) launch();
      (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("grid %4d x1024  %-22s %8.1f us  %7.1f GB/s\n", grid, name, ms / K * 1e3, moved / (ms / K * 1e-3) / 1e9);
    };
    run("read", [&] { hipLaunchKernelGGL(k_read, grid, 1024, 0, 0, a, n4, sink); }, (double)bytes);
    run("write", [&] { hipLaunchKernelGGL(k_write<false>, grid, 1024, 0, 0, b, n4); }, (double)bytes);
    run("write nt", [&] { hipLaunchKernelGGL(k_write<true>, grid, 1024, 0, 0, b, n4); }, (double)bytes);
    run("copy", [&] { hipLaunchKernelGGL(k_copy<false>, grid, 1024, 0, 0, a, b, n4); }, 2.0 * bytes);
    run("copy nt", [&] { hipLaunchKernelGGL(k_copy<true>, grid, 1024, 0, 0, a, b, n4); }, 2.0 * bytes);
    const size_t nr4 = n4 * 307200 / 393216;
    run("read then write 307:393", [&] { hipLaunchKernelGGL(k_mix<false>, grid, 1024, 0, 0, a, b, nr4, n4, sink); }, 16.0 * (nr4 + n4));
    run("same, nt stores", [&] { hipLaunchKernelGGL(k_mix<true>, grid, 1024, 0, 0, a, b, nr4, n4, sink); }, 16.0 * (nr4 + n4));
  }
  {
    const int nf = (int)(bytes / 393216);
    auto runv = [&](const char *name, auto launch) {
      for (int i = 0; i < 3; ++i) launch();
      (void)hipEventRecord(e0, 0);
      for (int i = 0; i < K; ++i) launch();
      (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("volumes x%d  %-34s %8.1f us  %7.1f GB/s\n", nf, name, ms / K * 1e3, (double)nf * 393216 / (ms / K * 1e-3) / 1e9);
    };
    const int grid = p.multiProcessorCount;
    runv("contiguous", [&] { hipLaunchKernelGGL((k_volumes<false, false>), grid, 1024, 0, 0, b, nf); });
    runv("contiguous nt", [&] { hipLaunchKernelGGL((k_volumes<true, false>), grid, 1024, 0, 0, b, nf); });
    runv("3 planes 128 KiB apart", [&] { hipLaunchKernelGGL((k_volumes<false, true>), grid, 1024, 0, 0, b, nf); });
    runv("3 planes 128 KiB apart nt", [&] { hipLaunchKernelGGL((k_volumes<true, true>), grid, 1024, 0, 0, b, nf); });
    runv("planes nt, start slice rotated x5", [&] { hipLaunchKernelGGL((k_volumes<true, true, 5>), grid, 1024, 0, 0, b, nf); });
    runv("planes nt, start slice rotated x2", [&] { hipLaunchKernelGGL((k_volumes<true, true, 2>), grid, 1024, 0, 0, b, nf); });
    runv("planes, start slice rotated x5", [&] { hipLaunchKernelGGL((k_volumes<false, true, 5>), grid, 1024, 0, 0, b, nf); });
    runv("contiguous nt, 256 thr/volume", [&] { hipLaunchKernelGGL((k_volumes_t<true, 256>), grid, 1024, 0, 0, b, nf); });
    runv("contiguous nt, 512 thr/volume", [&] { hipLaunchKernelGGL((k_volumes_t<true, 512>), grid, 1024, 0, 0, b, nf); });
    runv("contiguous nt, 1024 thr/volume", [&] { hipLaunchKernelGGL((k_volumes_t<true, 1024>), grid, 1024, 0, 0, b, nf); });
    runv("contiguous, 1024 thr/volume", [&] { hipLaunchKernelGGL((k_volumes_t<false, 1024>), grid, 1024, 0, 0, b, nf); });
    runv("nt, 2 workgroups/volume", [&] { hipLaunchKernelGGL((k_volumes_w<true, 2>), grid, 1024, 0, 0, b, nf); });
    runv("nt, 4 workgroups/volume", [&] { hipLaunchKernelGGL((k_volumes_w<true, 4>), grid, 1024, 0, 0, b, nf); });
    runv("nt, 8 workgroups/volume", [&] { hipLaunchKernelGGL((k_volumes_w<true, 8>), grid, 1024, 0, 0, b, nf); });
    runv("nt, 32 workgroups/volume", [&] { hipLaunchKernelGGL((k_volumes_w<true, 32>), grid, 1024, 0, 0, b, nf); });
    runv("nt, 256 workgroups/volume", [&] { hipLaunchKernelGGL((k_volumes_w<true, 256>), grid, 1024, 0, 0, b, nf); });
  }
  CK(hipDeviceSynchronize());
  printf("device: %s, %d CUs\n", p.name, p.multiProcessorCount);
  return 0;
}
