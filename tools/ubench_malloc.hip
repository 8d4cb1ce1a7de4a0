// What does a fresh device allocation cost the calling thread?  (cold-start diagnostics, profiles/r4_experiments.md)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_malloc.hip -o tools/bin/ubench_malloc
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void spin_kernel(unsigned long long *out, long long cycles) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) *out = 1;
}

int main() {
  hipSetDevice(0);
  void *warm = nullptr;
  double t = now_ms();
  hipMalloc(&warm, 1 << 20);
  printf("first hipMalloc (1 MB, runtime start-up): %.1f ms\n", now_ms() - t);
  const size_t sizes[] = {64ull << 20, 256ull << 20, 1ull << 30, 2ull << 30, 4ull << 30, 1ull << 30, 256ull << 20};
  std::vector<void *> held;
  for (size_t s : sizes) {
    void *p = nullptr;
    t = now_ms();
    hipError_t e = hipMalloc(&p, s);
    const double a = now_ms() - t;
    printf("hipMalloc %7.0f MB: %8.2f ms  (%.1f GB/s)%s\n", s / 1048576.0, a, s / a / 1e6, e == hipSuccess ? "" : "  FAILED");
    held.push_back(p);
  }
  for (size_t i = 0; i < held.size(); i++) {
    t = now_ms();
    hipFree(held[i]);
    printf("hipFree   %7.0f MB: %8.2f ms\n", sizes[i] / 1048576.0, now_ms() - t);
  }
  {  // again, after the frees: does the runtime / driver keep anything?
    void *p = nullptr;
    t = now_ms();
    hipMalloc(&p, 2ull << 30);
    printf("hipMalloc 2048 MB after the frees: %.2f ms\n", now_ms() - t);
    hipFree(p);
  }
  {  // the stream-ordered allocator
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    void *p = nullptr;
    t = now_ms();
    hipMallocAsync(&p, 2ull << 30, s);
    hipStreamSynchronize(s);
    printf("hipMallocAsync 2048 MB (first): %.2f ms\n", now_ms() - t);
    hipFreeAsync(p, s);
    t = now_ms();
    hipMallocAsync(&p, 2ull << 30, s);
    hipStreamSynchronize(s);
    printf("hipMallocAsync 2048 MB (second, pool warm): %.2f ms\n", now_ms() - t);
    hipFreeAsync(p, s);
    hipStreamSynchronize(s);
  }
  {  // does an allocation on another thread hold up launches on this one?
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned long long *flag = nullptr;
    hipMalloc(reinterpret_cast<void **>(&flag), 8);
    spin_kernel<<<1, 64, 0, s>>>(flag, 1000);
    hipStreamSynchronize(s);
    double allocMs = 0;
    std::thread bg([&] {
      hipSetDevice(0);
      void *p = nullptr;
      const double t0 = now_ms();
      hipMalloc(&p, 4ull << 30);
      allocMs = now_ms() - t0;
      hipFree(p);
    });
    double worst = 0, total = 0;
    int launches = 0;
    const double t0 = now_ms();
    while (now_ms() - t0 < 300) {
      const double l0 = now_ms();
      spin_kernel<<<1, 64, 0, s>>>(flag, 1000);
      hipStreamSynchronize(s);
      const double d = now_ms() - l0;
      worst = d > worst ? d : worst;
      total += d;
      launches++;
    }
    bg.join();
    printf("4096 MB hipMalloc on a second thread: %.1f ms; %d launch+sync pairs meanwhile, mean %.3f ms, worst %.2f ms\n", allocMs, launches,
           total / launches, worst);
  }
  return 0;
}
