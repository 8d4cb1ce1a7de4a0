// Host cost of getting one small result back from a kernel (what every count-returning ABI call does):
//   A  launch + hipStreamSynchronize (the result lands in mapped pinned memory, written by the kernel)
//   B  launch, then the host POLLS a sequence word in mapped pinned memory the kernel's last lane writes (system scope)
//   C  launch + 16-byte hipMemcpyAsync D2H + hipStreamSynchronize
//   D  launch + hipEventRecord + hipEventSynchronize
// each for a ~2 us and a ~50 us kernel.   hipcc --offload-arch=gfx950 -O2 -o tools/bin/ubench_sync tools/ubench_sync.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <immintrin.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void work(volatile unsigned *host, unsigned seq, int spin, unsigned *dev) {
  unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    dev[0] = seq;
    host[1] = seq * 3u;                      // the result
    __threadfence_system();
    __hip_atomic_store(const_cast<unsigned *>(host), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);  // the sequence word
  }
}
int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned *pinned, *dev; CK(hipHostMalloc((void **)&pinned, 64, hipHostMallocMapped)); CK(hipMalloc((void **)&dev, 64));
  pinned[0] = 0;
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  unsigned seq = 0, hostcopy[4];
  for (int spin : {200, 5000}) {  // 100 MHz clock ticks: 2 us, 50 us
    for (int mode = 0; mode < 4; mode++) {
      const int reps = 2000;
      double total = 0;
      for (int r = -50; r < reps; r++) {
        auto t0 = std::chrono::steady_clock::now();
        seq++;
        hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s, pinned, seq, spin, dev);
        if (mode == 0) CK(hipStreamSynchronize(s));
        else if (mode == 1) { while (__atomic_load_n(pinned, __ATOMIC_ACQUIRE) != seq) _mm_pause(); }
        else if (mode == 2) { CK(hipMemcpyAsync(hostcopy, dev, 16, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); }
        else { CK(hipEventRecord(ev, s)); CK(hipEventSynchronize(ev)); }
        auto t1 = std::chrono::steady_clock::now();
        if (pinned[1] != seq * 3u) { printf("result not visible\n"); return 2; }
        if (r >= 0) total += std::chrono::duration<double, std::micro>(t1 - t0).count();
      }
      if (mode == 1) CK(hipStreamSynchronize(s));
      const char *names[] = {"A launch + hipStreamSynchronize", "B launch + host polls pinned word", "C launch + 16 B D2H copy + sync", "D launch + event record + event sync"};
      printf("kernel %3d us: %-40s %7.2f us per round trip\n", spin / 100, names[mode], total / reps);
    }
  }
  return 0;
}
