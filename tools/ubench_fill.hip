// How fast does hipMemsetAsync clear memory on this GPU, against a plain 16-byte-store kernel?
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_fill tools/ubench_fill.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void __launch_bounds__(256) fill16(uint4 *p, size_t n) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * 256)
    p[i] = make_uint4(0u, 0u, 0u, 0u);
}
__global__ void __launch_bounds__(256) fill16nt(uint4 *p, size_t n) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * 256)
  {
    typedef unsigned int v4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(v4{0u, 0u, 0u, 0u}, reinterpret_cast<v4 *>(p) + i);
  }
}
int main() {
  for (size_t bytes : {size_t(64) << 20, size_t(256) << 20, size_t(2) << 30}) {
    void *p; CHECK(hipMalloc(&p, bytes));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int mode = 0; mode < 4; mode++) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(a));
        if (mode == 0) CHECK(hipMemsetAsync(p, 0, bytes, 0));
        else if (mode == 1) hipLaunchKernelGGL(fill16, dim3(256 * 8), dim3(256), 0, 0, static_cast<uint4 *>(p), bytes / 16);
        else if (mode == 2) hipLaunchKernelGGL(fill16, dim3(256 * 32), dim3(256), 0, 0, static_cast<uint4 *>(p), bytes / 16);
        else hipLaunchKernelGGL(fill16nt, dim3(256 * 8), dim3(256), 0, 0, static_cast<uint4 *>(p), bytes / 16);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
      }
      const char *names[] = {"hipMemsetAsync", "fill16 x2048 wg", "fill16 x8192 wg", "fill16 nontemporal"};
      printf("%5zu MB  %-20s %.3f ms  %.0f GB/s\n", bytes >> 20, names[mode], best, bytes / (best * 1e-3) / 1e9);
    }
    CHECK(hipFree(p));
  }
  return 0;
}
