// Micro-benchmark: how fast can N workgroups each stream their OWN contiguous region (the access
// pattern of hr_merge_kernel) compared with a plain grid-stride read of the same bytes?
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_streams tools/ubench_streams.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(1024) void grid_stride(const uint4 *p, size_t n, unsigned *sink) {
  unsigned acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = p[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}

// region r = [r * cap, r * cap + n); SPLIT workgroups share a region (each takes every SPLIT-th batch)
extern __shared__ unsigned char dynLds[];
template <int BATCH, int THREADS>
__global__ __launch_bounds__(THREADS) void per_region(const uint4 *p, size_t cap, size_t n, int split, unsigned *sink) {
  if (n == 1) dynLds[threadIdx.x] = 1;  // keeps the dynamic LDS allocation alive
  const int r = blockIdx.x / split, part = blockIdx.x % split;
  const uint4 *rec = p + (size_t)r * cap;
  unsigned acc = 0;
  const size_t stride = (size_t)BATCH * THREADS;
  uint4 a[BATCH], b[BATCH];
  auto load = [&](uint4 (&x)[BATCH], size_t base) {
#pragma unroll
    for (int k = 0; k < BATCH; k++) {
      const size_t i = base + (size_t)k * THREADS + threadIdx.x;
      x[k] = i < n ? rec[i] : make_uint4(0, 0, 0, 0);
    }
  };
  auto use = [&](const uint4 (&x)[BATCH]) {
#pragma unroll
    for (int k = 0; k < BATCH; k++) acc ^= x[k].x ^ x[k].y ^ x[k].z ^ x[k].w;
  };
  const size_t first = (size_t)part * stride, step = (size_t)split * stride;
  load(a, first);
  for (size_t base = first; base < n; base += 2 * step) {
    if (base + step < n) load(b, base + step);
    use(a);
    if (base + 2 * step < n) load(a, base + 2 * step);
    if (base + step < n) use(b);
  }
  if (acc == 0x12345678u) *sink = acc;
}

int main(int argc, char **argv) {
  const int regions = argc > 1 ? atoi(argv[1]) : 512;
  const size_t n = argc > 2 ? atol(argv[2]) : 117000;   // records per region
  const size_t cap = ((2 * n + 2 * 8192) | 63) + 18;
  const size_t total = cap * regions;
  uint4 *buf; unsigned *sink;
  CK(hipMalloc(&buf, total * 16)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 1, total * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char *name, double bytes, auto launch) {
    float best = 1e9;
    for (int it = 0; it < 5; it++) {
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-44s %8.3f ms  %7.2f TB/s\n", name, best, bytes / best / 1e9);
  };
  const double used = (double)n * regions * 16;
  timeit("grid-stride read of the used bytes (contig)", used, [&] { grid_stride<<<256 * 8, 1024>>>(buf, (size_t)n * regions, sink); });
  timeit("grid-stride, 2048 WG x 256", used, [&] { grid_stride<<<2048, 256>>>(buf, (size_t)n * regions, sink); });
  timeit("per-region 1024 thr, batch 4", used, [&] { per_region<4, 1024><<<regions, 1024>>>(buf, cap, n, 1, sink); });
  timeit("per-region 1024 thr, batch 8", used, [&] { per_region<8, 1024><<<regions, 1024>>>(buf, cap, n, 1, sink); });
  timeit("per-region 1024 thr, batch 2", used, [&] { per_region<2, 1024><<<regions, 1024>>>(buf, cap, n, 1, sink); });
  timeit("per-region 256 thr, batch 4, 4 WG/region", used, [&] { per_region<4, 256><<<regions * 4, 256>>>(buf, cap, n, 4, sink); });
  timeit("per-region 256 thr, batch 4, 1 WG/region", used, [&] { per_region<4, 256><<<regions, 256>>>(buf, cap, n, 1, sink); });
  timeit("per-region 512 thr, batch 4, 2 WG/region", used, [&] { per_region<4, 512><<<regions * 2, 512>>>(buf, cap, n, 2, sink); });
  timeit("per-region 1024 thr, batch 4, 2 WG/region", used, [&] { per_region<4, 1024><<<regions * 2, 1024>>>(buf, cap, n, 2, sink); });
  // the merge kernel owns 128 KiB of LDS: one 1024-lane workgroup per CU
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(per_region<4, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(per_region<8, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(per_region<2, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  timeit("1 WG/CU (128K LDS) 1024 thr, batch 4", used, [&] { per_region<4, 1024><<<regions, 1024, 128 * 1024>>>(buf, cap, n, 1, sink); });
  timeit("1 WG/CU (128K LDS) 1024 thr, batch 8", used, [&] { per_region<8, 1024><<<regions, 1024, 128 * 1024>>>(buf, cap, n, 1, sink); });
  timeit("1 WG/CU (128K LDS) 1024 thr, batch 2", used, [&] { per_region<2, 1024><<<regions, 1024, 128 * 1024>>>(buf, cap, n, 1, sink); });
  timeit("2 WG/CU (64K LDS) 1024 thr, batch 4", used, [&] { per_region<4, 1024><<<regions, 1024, 64 * 1024>>>(buf, cap, n, 1, sink); });
  return 0;
}
