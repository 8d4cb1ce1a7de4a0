/* ares_memory.h — C ABI of libmem.so, the allocator / stream / copy library.
 *
 * Drop-in for the reference's cgoutils/memory.h:51-99 (bound from Go by
 * cgoutils/memory.go:17-19 with `-lmem`).  Symbol names, argument order and result
 * conventions are identical; the implementation behind them (aresdb_amd/csrc/mem) is HIP:
 * pinned host allocations, a stream-ordered device pool, hipMemcpyAsync, roctx/hip profiler
 * hooks.  Unlike the reference header (memory.h:34-42) nothing is *defined* here, so it can be
 * included from any number of translation units.
 */
#ifndef ARES_MEMORY_H_
#define ARES_MEMORY_H_

#include <stddef.h>
#include <stdint.h>
#include "ares_cgo.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Capability bits returned by GetFlags() — reference memory.h:28-32, read by
 * cgoutils/memory.go:27-41 (IsDeviceMemoryImplementation / IsPooledMemory / SupportHashReduction). */
enum {
  DEVICE_MEMORY_IMPLEMENTATION_FLAG = 1,
  POOLED_MEMORY_FLAG = 1 << 1,
  HASH_REDUCTION_SUPPORT = 1 << 2,
};

typedef uint32_t DeviceMemoryFlags;

/* memory.h:49 — never fails, returns the plain flag word. */
DeviceMemoryFlags GetFlags(void);

/* memory.h:51-55 — pinned (portable) host memory, zero-filled (cuda_malloc.cu:44-52). */
CGoCallResHandle HostAlloc(size_t bytes);
CGoCallResHandle HostFree(void *p);
CGoCallResHandle HostMemCpy(void *dst, const void *src, size_t bytes);

/* memory.h:57-61 — one HIP stream per call; `res` is the stream handle. */
CGoCallResHandle CreateCudaStream(int device);
CGoCallResHandle WaitForCudaStream(void *s, int device);
CGoCallResHandle DestroyCudaStream(void *s, int device);

/* memory.h:63-65 — device memory; DeviceAllocate returns ZEROED memory (cuda_malloc.cu:97-104). */
CGoCallResHandle DeviceAllocate(size_t bytes, int device);
CGoCallResHandle DeviceFree(void *p, int device);

/* memory.h:67-74 — asynchronous copies on `stream`. */
CGoCallResHandle AsyncCopyHostToDevice(void *dst, void *src, size_t bytes, void *stream, int device);
CGoCallResHandle AsyncCopyDeviceToDevice(void *dst, void *src, size_t bytes, void *stream, int device);
CGoCallResHandle AsyncCopyDeviceToHost(void *dst, void *src, size_t bytes, void *stream, int device);

/* memory.h:76-86 — device discovery, profiler hooks, pool statistics. */
CGoCallResHandle GetDeviceCount(void);
CGoCallResHandle GetDeviceGlobalMemoryInMB(int device);
CGoCallResHandle CudaProfilerStart(void);
CGoCallResHandle CudaProfilerStop(void);
CGoCallResHandle GetDeviceMemoryInfo(size_t *freeSize, size_t *totalSize, int device);

/* memory.h:90-98 — library-internal (called by libalgorithm.so with the device already set). */
CGoCallResHandle deviceMalloc(void **devPtr, size_t size);
CGoCallResHandle deviceFree(void *devPtr);
CGoCallResHandle deviceMemset(void *devPtr, int value, size_t count);
CGoCallResHandle asyncCopyHostToDevice(void *dst, const void *src, size_t count, void *stream);
CGoCallResHandle asyncCopyDeviceToHost(void *dst, const void *src, size_t count, void *stream);
CGoCallResHandle waitForCudaStream(void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ARES_MEMORY_H_ */
