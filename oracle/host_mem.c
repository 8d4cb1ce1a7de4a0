/* host_mem.c — host-memory stand-in for libmem.so, used ONLY together with aql_oracle.c.
 *
 * TEST INFRASTRUCTURE.  Restates the reference's QUERY_MODE=HOST allocator
 * (cgoutils/memory/malloc.c:21-162): "device" memory is zeroed malloc memory, streams are
 * NULL, copies are memcpy, one simulated device.  Exports the libmem ABI of
 * include/ares_memory.h so that the host-side driver can run the same call sequence against
 * the oracle as against the HIP library.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ares_memory.h"

#define OK(p) ((CGoCallResHandle){(void *)(p), NULL})

static char *fmt_error(const char *fn, const char *msg) { /* memory.h:36-42 */
  char *b = (char *)malloc(100);
  snprintf(b, 100, "ERROR when making C function %s: %s\n", fn, msg);
  return b;
}

DeviceMemoryFlags GetFlags(void) { return HASH_REDUCTION_SUPPORT; } /* malloc.c:21-23 */

CGoCallResHandle HostAlloc(size_t bytes) { /* malloc.c:25-30 */
  void *p = malloc(bytes ? bytes : 1);
  if (p) memset(p, 0, bytes);
  return OK(p);
}
CGoCallResHandle HostFree(void *p) { free(p); return OK(NULL); }
CGoCallResHandle HostMemCpy(void *dst, const void *src, size_t bytes) { memcpy(dst, src, bytes); return OK(NULL); }
CGoCallResHandle CreateCudaStream(int device) { (void)device; return OK(NULL); }
CGoCallResHandle WaitForCudaStream(void *s, int device) { (void)s; (void)device; return OK(NULL); }
CGoCallResHandle DestroyCudaStream(void *s, int device) { (void)s; (void)device; return OK(NULL); }
CGoCallResHandle DeviceAllocate(size_t bytes, int device) { (void)device; return HostAlloc(bytes); }
CGoCallResHandle DeviceFree(void *p, int device) { (void)device; return HostFree(p); }
CGoCallResHandle AsyncCopyHostToDevice(void *dst, void *src, size_t bytes, void *stream, int device) {
  (void)stream; (void)device; memcpy(dst, src, bytes); return OK(NULL); }
CGoCallResHandle AsyncCopyDeviceToDevice(void *dst, void *src, size_t bytes, void *stream, int device) {
  (void)stream; (void)device; memmove(dst, src, bytes); return OK(NULL); }
CGoCallResHandle AsyncCopyDeviceToHost(void *dst, void *src, size_t bytes, void *stream, int device) {
  (void)stream; (void)device; memcpy(dst, src, bytes); return OK(NULL); }
CGoCallResHandle GetDeviceCount(void) { return OK((void *)1); }                      /* malloc.c:97-100 */
CGoCallResHandle GetDeviceGlobalMemoryInMB(int device) { (void)device; return OK((void *)24392); } /* :102-106 */
CGoCallResHandle CudaProfilerStart(void) { return OK(NULL); }
CGoCallResHandle CudaProfilerStop(void) { return OK(NULL); }
CGoCallResHandle GetDeviceMemoryInfo(size_t *freeSize, size_t *totalSize, int device) {
  (void)freeSize; (void)totalSize; (void)device;
  return (CGoCallResHandle){NULL, fmt_error("GetDeviceMemoryInfo", "Not supported")};
}
CGoCallResHandle deviceMalloc(void **devPtr, size_t size) { *devPtr = malloc(size); return OK(NULL); }
CGoCallResHandle deviceFree(void *devPtr) { free(devPtr); return OK(NULL); }
CGoCallResHandle deviceMemset(void *devPtr, int value, size_t count) { memset(devPtr, value, count); return OK(NULL); }
CGoCallResHandle asyncCopyHostToDevice(void *dst, const void *src, size_t count, void *stream) {
  (void)stream; memcpy(dst, src, count); return OK(NULL); }
CGoCallResHandle asyncCopyDeviceToHost(void *dst, const void *src, size_t count, void *stream) {
  (void)stream; memcpy(dst, src, count); return OK(NULL); }
CGoCallResHandle waitForCudaStream(void *stream) { (void)stream; return OK(NULL); }
