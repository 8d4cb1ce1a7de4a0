// How many 256-thread workgroups of a given static LDS size does a CU hold, and how fast are short workgroups dispatched?
// Each workgroup touches its LDS and spins for `us` microseconds (wall clock); grid = 131072 workgroups.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_residency tools/ubench_residency.hip && tools/bin/ubench_residency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int KB>
__global__ __launch_bounds__(256) void spin_kernel(unsigned long long *stamps, int ticks) {
  __shared__ unsigned int lds[KB * 256];
  const unsigned long long t0 = wall_clock64();
  for (int i = threadIdx.x; i < KB * 256; i += 256) lds[i] = i;
  __syncthreads();
  unsigned int acc = lds[(threadIdx.x * 7) % (KB * 256)];
  while (wall_clock64() - t0 < static_cast<unsigned long long>(ticks)) acc += lds[acc % (KB * 256)];
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = wall_clock64() + (acc == 0xFFFFFFFFu);
  }
}
template <int KB>
void run(int us, int grid) {
  unsigned long long *d;
  hipMalloc(&d, sizeof(unsigned long long) * 2 * grid);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  spin_kernel<KB><<<grid, 256>>>(d, us * 100);
  hipDeviceSynchronize();
  hipEventRecord(a);
  spin_kernel<KB><<<grid, 256>>>(d, us * 100);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h(2 * grid);
  hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
  double resident = 0;
  unsigned long long first = ~0ull, last = 0;
  for (int i = 0; i < grid; i++) {
    resident += static_cast<double>(h[2 * i + 1] - h[2 * i]);
    if (h[2 * i] < first) first = h[2 * i];
    if (h[2 * i + 1] > last) last = h[2 * i + 1];
  }
  printf("LDS %3d KB, %2d us per workgroup, %d workgroups: %.3f ms, %.0f resident on average (%.2f per CU), %.1f workgroups/us\n", KB, us, grid, ms,
         resident / static_cast<double>(last - first), resident / static_cast<double>(last - first) / 256.0, grid / (ms * 1e3));
  hipFree(d);
}
int main() {
  const int grid = 131072;
  for (int us : {2, 9, 30}) {
    run<1>(us, grid);
    run<16>(us, grid);
    run<37>(us, grid);
    run<52>(us, grid);
    run<64>(us, grid);
    run<80>(us, grid);
  }
  return 0;
}
