// libmem.so for MI355X — allocator / stream / copy library behind include/ares_memory.h.
//
// Replaces the reference's three libmem flavours (cgoutils/memory/malloc.c, cuda_malloc.cu,
// rmm_alloc.cu) with one HIP implementation designed for a 288 GB HBM3E device:
//   * host memory is pinned + portable (hipHostMalloc) and zero-filled, so AsyncCopyHostToDevice
//     is a true asynchronous DMA (reference cuda_malloc.cu:44-52);
//   * device memory comes from the device's stream-ordered pool (hipMallocAsync) with the release
//     threshold lifted, so the per-batch, per-column alloc/free pattern of the Go host
//     (query/aql_processor.go:726-804, :1345-1431) never reaches the driver after warm-up; the
//     block is zeroed before DeviceAllocate returns (reference cuda_malloc.cu:97-104 contract);
//   * ARES_MEM_POOL=0 switches to plain hipMalloc/hipFree (debugging aid, same semantics).
// Every Go-facing entry point selects the device itself; the lower-case ones assume the caller
// (libalgorithm.so) already did (reference cgoutils/memory.h:88-98).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "ares_memory.h"

#pragma clang diagnostic ignored "-Wdeprecated-declarations"  // hipProfilerStart/Stop: the ABI asks for them

namespace {

constexpr int kMaxErrorLen = 160;
constexpr int kMaxDevices = 64;

char *format_error(const char *what, hipError_t err) {
  char *buf = static_cast<char *>(malloc(kMaxErrorLen));
  snprintf(buf, kMaxErrorLen, "ERROR when calling HIP functions: %s: %s\n", what,
           hipGetErrorString(err));
  return buf;
}

char *format_message(const char *fn, const char *msg) {
  char *buf = static_cast<char *>(malloc(kMaxErrorLen));
  snprintf(buf, kMaxErrorLen, "ERROR when making C function %s: %s\n", fn, msg);
  return buf;
}

inline CGoCallResHandle ok(void *res = nullptr) { return CGoCallResHandle{res, nullptr}; }
inline CGoCallResHandle fail(const char *what, hipError_t err) {
  (void)hipGetLastError();  // clear the sticky error like cudaGetLastError does
  return CGoCallResHandle{nullptr, format_error(what, err)};
}

#define MEM_TRY(expr, what)                        \
  do {                                             \
    hipError_t e_ = (expr);                        \
    if (e_ != hipSuccess) return fail(what, e_);   \
  } while (0)

bool use_pool() {
  static const bool v = [] {
    const char *e = getenv("ARES_MEM_POOL");
    return !(e && e[0] == '0');
  }();
  return v;
}

// One allocation stream per device: DeviceAllocate = hipMallocAsync + hipMemsetAsync + sync on
// this stream, so the zeroed block is ready for any other stream when the call returns.
struct DeviceState {
  std::once_flag once;
  hipStream_t allocStream = nullptr;
  hipError_t initError = hipSuccess;
};
DeviceState g_devices[kMaxDevices];

hipError_t device_state(int device, DeviceState **out) {
  if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
  DeviceState &st = g_devices[device];
  std::call_once(st.once, [&] {
    hipError_t e = hipStreamCreateWithFlags(&st.allocStream, hipStreamNonBlocking);
    if (e == hipSuccess && use_pool()) {
      hipMemPool_t pool;
      e = hipDeviceGetDefaultMemPool(&pool, device);
      if (e == hipSuccess) {
        uint64_t keep = UINT64_MAX;  // never trim: batches re-use the same sizes every ~ms
        e = hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
      }
    }
    st.initError = e;
  });
  *out = &st;
  return st.initError;
}

hipError_t current_device_state(DeviceState **out) {
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) return e;
  return device_state(device, out);
}

hipError_t pool_alloc(DeviceState *st, void **p, size_t bytes, bool zero) {
  if (bytes == 0) bytes = 1;
  hipError_t e;
  if (use_pool()) {
    e = hipMallocAsync(p, bytes, st->allocStream);
    if (e != hipSuccess) return e;
    if (zero) {
      e = hipMemsetAsync(*p, 0, bytes, st->allocStream);
      if (e != hipSuccess) return e;
    }
    return hipStreamSynchronize(st->allocStream);
  }
  e = hipMalloc(p, bytes);
  if (e != hipSuccess) return e;
  return zero ? hipMemset(*p, 0, bytes) : hipSuccess;
}

hipError_t pool_free(DeviceState *st, void *p) {
  if (p == nullptr) return hipSuccess;
  if (use_pool()) return hipFreeAsync(p, st->allocStream);
  return hipFree(p);
}

}  // namespace

extern "C" {

DeviceMemoryFlags GetFlags(void) {
  // reference cuda_malloc.cu:36-42 / rmm_alloc.cu:84-91
  DeviceMemoryFlags f = DEVICE_MEMORY_IMPLEMENTATION_FLAG | HASH_REDUCTION_SUPPORT;
  if (use_pool()) f |= POOLED_MEMORY_FLAG;
  return f;
}

CGoCallResHandle HostAlloc(size_t bytes) {
  void *p = nullptr;
  MEM_TRY(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable), "Allocate");
  memset(p, 0, bytes);
  return ok(p);
}

CGoCallResHandle HostFree(void *p) {
  if (p) MEM_TRY(hipHostFree(p), "Free");
  return ok();
}

CGoCallResHandle HostMemCpy(void *dst, const void *src, size_t bytes) {
  if (memcpy(dst, src, bytes) != dst)
    return CGoCallResHandle{nullptr,
                            format_message("HostMemCpy", "Returned pointer does not match destination")};
  return ok();
}

CGoCallResHandle CreateCudaStream(int device) {
  MEM_TRY(hipSetDevice(device), "CreateCudaStream");
  hipStream_t s = nullptr;
  MEM_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "CreateCudaStream");
  return ok(reinterpret_cast<void *>(s));
}

CGoCallResHandle WaitForCudaStream(void *s, int device) {
  MEM_TRY(hipSetDevice(device), "WaitForCudaStream");
  MEM_TRY(hipStreamSynchronize(reinterpret_cast<hipStream_t>(s)), "WaitForCudaStream");
  return ok();
}

CGoCallResHandle DestroyCudaStream(void *s, int device) {
  MEM_TRY(hipSetDevice(device), "DestroyCudaStream");
  if (s) MEM_TRY(hipStreamDestroy(reinterpret_cast<hipStream_t>(s)), "DestroyCudaStream");
  return ok();
}

CGoCallResHandle DeviceAllocate(size_t bytes, int device) {
  MEM_TRY(hipSetDevice(device), "DeviceAllocate");
  DeviceState *st;
  MEM_TRY(device_state(device, &st), "DeviceAllocate");
  void *p = nullptr;
  MEM_TRY(pool_alloc(st, &p, bytes, /*zero=*/true), "DeviceAllocate");
  return ok(p);
}

CGoCallResHandle DeviceFree(void *p, int device) {
  MEM_TRY(hipSetDevice(device), "DeviceFree");
  DeviceState *st;
  MEM_TRY(device_state(device, &st), "DeviceFree");
  MEM_TRY(pool_free(st, p), "DeviceFree");
  return ok();
}

CGoCallResHandle AsyncCopyHostToDevice(void *dst, void *src, size_t bytes, void *stream, int device) {
  MEM_TRY(hipSetDevice(device), "AsyncCopyHostToDevice");
  if (bytes)
    MEM_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, reinterpret_cast<hipStream_t>(stream)),
            "AsyncCopyHostToDevice");
  return ok();
}

CGoCallResHandle AsyncCopyDeviceToDevice(void *dst, void *src, size_t bytes, void *stream, int device) {
  MEM_TRY(hipSetDevice(device), "AsyncCopyDeviceToDevice");
  if (bytes)
    MEM_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream)),
            "AsyncCopyDeviceToDevice");
  return ok();
}

CGoCallResHandle AsyncCopyDeviceToHost(void *dst, void *src, size_t bytes, void *stream, int device) {
  MEM_TRY(hipSetDevice(device), "AsyncCopyDeviceToHost");
  if (bytes)
    MEM_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, reinterpret_cast<hipStream_t>(stream)),
            "AsyncCopyDeviceToHost");
  return ok();
}

CGoCallResHandle GetDeviceCount(void) {
  int n = 0;
  MEM_TRY(hipGetDeviceCount(&n), "GetDeviceCount");
  return ok(reinterpret_cast<void *>(static_cast<intptr_t>(n)));
}

CGoCallResHandle GetDeviceGlobalMemoryInMB(int device) {
  hipDeviceProp_t prop;
  MEM_TRY(hipGetDeviceProperties(&prop, device), "GetDeviceGlobalMemoryInMB");
  return ok(reinterpret_cast<void *>(static_cast<intptr_t>(prop.totalGlobalMem / (1024 * 1024))));
}

// The Go host brackets one stage of one batch with these when `profiling=<stage>` is requested
// (query/aql_processor.go:808-822); under rocprofv3 they delimit the collection window.
CGoCallResHandle CudaProfilerStart(void) {
  (void)hipProfilerStart();
  (void)hipGetLastError();
  return ok();
}

CGoCallResHandle CudaProfilerStop(void) {
  MEM_TRY(hipDeviceSynchronize(), "cudaProfilerStop");
  (void)hipProfilerStop();
  (void)hipGetLastError();
  return ok();
}

CGoCallResHandle GetDeviceMemoryInfo(size_t *freeSize, size_t *totalSize, int device) {
  // pooled flavour of the reference (rmm_alloc.cu:239-246): real numbers instead of "Not supported"
  MEM_TRY(hipSetDevice(device), "GetDeviceMemoryInfo");
  MEM_TRY(hipMemGetInfo(freeSize, totalSize), "GetDeviceMemoryInfo");
  return ok();
}

CGoCallResHandle deviceMalloc(void **devPtr, size_t size) {
  DeviceState *st;
  MEM_TRY(current_device_state(&st), "deviceMalloc");
  MEM_TRY(pool_alloc(st, devPtr, size, /*zero=*/false), "deviceMalloc");
  return ok();
}

CGoCallResHandle deviceFree(void *devPtr) {
  DeviceState *st;
  MEM_TRY(current_device_state(&st), "deviceFree");
  MEM_TRY(pool_free(st, devPtr), "deviceFree");
  return ok();
}

CGoCallResHandle deviceMemset(void *devPtr, int value, size_t count) {
  MEM_TRY(hipMemset(devPtr, value, count), "deviceMemset");
  return ok();
}

CGoCallResHandle asyncCopyHostToDevice(void *dst, const void *src, size_t count, void *stream) {
  if (count)
    MEM_TRY(hipMemcpyAsync(dst, src, count, hipMemcpyHostToDevice, reinterpret_cast<hipStream_t>(stream)),
            "asyncCopyHostToDevice");
  return ok();
}

CGoCallResHandle asyncCopyDeviceToHost(void *dst, const void *src, size_t count, void *stream) {
  if (count)
    MEM_TRY(hipMemcpyAsync(dst, src, count, hipMemcpyDeviceToHost, reinterpret_cast<hipStream_t>(stream)),
            "asyncCopyDeviceToHost");
  return ok();
}

CGoCallResHandle waitForCudaStream(void *stream) {
  MEM_TRY(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)), "waitForCudaStream");
  return ok();
}

}  // extern "C"
