// Host helpers shared by the entry points (see common.hpp).
#include "common.hpp"

namespace ares {

namespace {
struct PinnedSlot {
  uint64_t *ptr = nullptr;
  PinnedSlot() {
    if (hipHostMalloc(reinterpret_cast<void **>(&ptr), 8 * sizeof(uint64_t), hipHostMallocPortable) != hipSuccess) {
      (void)hipGetLastError();
      ptr = nullptr;
    }
  }
  ~PinnedSlot() {
    if (ptr) (void)hipHostFree(ptr);
  }
};
}  // namespace

uint64_t *pinned_words() {
  thread_local PinnedSlot slot;
  if (!slot.ptr) throw AlgorithmError("ERROR: cannot allocate pinned result words");
  return slot.ptr;
}

void read_back_u32(const uint32_t *dev, uint32_t *host, int count, hipStream_t stream) {
  uint32_t *pinned = reinterpret_cast<uint32_t *>(pinned_words());
  hip_check(hipMemcpyAsync(pinned, dev, sizeof(uint32_t) * count, hipMemcpyDeviceToHost, stream), "read back result");
  hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
  for (int i = 0; i < count; i++) host[i] = pinned[i];
}

}  // namespace ares
