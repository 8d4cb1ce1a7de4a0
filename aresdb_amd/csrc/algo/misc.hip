// Entry points of the ABI that are outside the hot-path scope of this build (SURVEY.md 8:
// geo intersection) plus BootstrapDevice.  They are exported so
// that the library is link-compatible with the Go host (query/time_series_aggregate.go binds all
// 14 symbols); calling one returns a clean error through the cgo convention instead of crashing.
#include "common.hpp"

using namespace ares;

extern "C" {

CGoCallResHandle GeoBatchIntersects(GeoShapeBatch, InputVector, uint32_t *, int, uint32_t, RecordID **, int,
                                    uint32_t *, bool, void *, int device) {
  ARES_ABI_BEGIN(device)
  throw AlgorithmError("GeoBatchIntersects is outside the scope of the MI355X library (SURVEY.md 8)");
  ARES_ABI_END("GeoBatchIntersects")
}

CGoCallResHandle WriteGeoShapeDim(int, DimensionOutputVector, int, uint32_t *, void *, int device) {
  ARES_ABI_BEGIN(device)
  throw AlgorithmError("WriteGeoShapeDim is outside the scope of the MI355X library (SURVEY.md 8)");
  ARES_ABI_END("WriteGeoShapeDim")
}

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
