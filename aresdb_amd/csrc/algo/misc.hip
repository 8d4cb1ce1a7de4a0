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
    }
    if (n > 0) hip_check(hipSetDevice(0), "hipSetDevice");
  } catch (std::exception &e) {
    fprintf(stderr, "Exception happened when bootstrapping device: %s\n", e.what());
    resHandle.pStrErr = strdup(e.what());
  }
  return resHandle;
}

}  // extern "C"
