// BootstrapDevice (query/utils.cu:63-85).
#include "common.hpp"

using namespace ares;

extern "C" {

// The reference uploads its calendar table to constant memory on every device
// (query/utils.cu:63-85).  Here the table is an immediate inside the kernels; bootstrapping
// only warms the runtime up (context + stream-ordered pool) so the first query does not pay it.
CGoCallResHandle BootstrapDevice(void) {
  CGoCallResHandle resHandle = {nullptr, nullptr};
  try {
    int n = 0;
    hip_check(hipGetDeviceCount(&n), "hipGetDeviceCount");
    for (int d = 0; d < n; d++) {
      hip_check(hipSetDevice(d), "hipSetDevice");
      hipMemPool_t pool;
      hip_check(hipDeviceGetDefaultMemPool(&pool, d), "hipDeviceGetDefaultMemPool");
      uint64_t keep = UINT64_MAX;
      hip_check(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep), "hipMemPoolSetAttribute");
      // The kernel driver charges for device memory it hands out for the first time since the box came up (~28 ms per GB: the
      // first query of the first process on a fresh box took 270-290 ms, of every later process 30 ms — 7 GB of result vectors
      // and workspace).  A server pays that here, once, not inside its first query: a slab is allocated, touched and given
      // back (the reference's pooled allocator reserves its pool before main() runs: cgoutils/memory/rmm_alloc.cu:50-83).
      // ARES_BOOTSTRAP_WARM_MB: megabytes per device (default 8192, capped at a quarter of the device; 0: off).
      static const size_t warmMB = [] {
        const char *e = getenv("ARES_BOOTSTRAP_WARM_MB");
        return static_cast<size_t>(e ? atoll(e) : 8192);
      }();
      size_t freeB = 0, totalB = 0;
      if (warmMB && hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
        size_t bytes = warmMB << 20;
        if (bytes > totalB / 4) bytes = totalB / 4;
        if (bytes > freeB / 2) bytes = freeB / 2;
        void *slab = nullptr;
        if (bytes && hipMalloc(&slab, bytes) == hipSuccess) {
          (void)hipMemset(slab, 0, bytes);
          (void)hipFree(slab);
        } else {
          (void)hipGetLastError();
        }
      }
    }
    if (n > 0) hip_check(hipSetDevice(0), "hipSetDevice");
  } catch (std::exception &e) {
    fprintf(stderr, "Exception happened when bootstrapping device: %s\n", e.what());
    resHandle.pStrErr = strdup(e.what());
  }
  return resHandle;
}

}  // extern "C"
